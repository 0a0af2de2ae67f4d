"""Seeded random sweeps of the HIP nodes against the CPU oracle (-m gpu): every case draws its
sizes, tap counts, rates, phases, states and batch cuts from a fixed-seed generator, so a
failure reproduces by case number.  Complements test_gpu_parity.py's hand-picked cases (a sweep
of this kind found the post-FIR-mixer + FM demod bug of the overlap-save fusion)."""
import numpy as np
import pytest

import oracle
from test_gpu_parity import TOL, circ, fft_close, fir_close, lowpass_taps, rand_c  # noqa: F401

pytestmark = pytest.mark.gpu

# COMMS_TEST_SEED_OFFSET=k re-runs the sweeps on other draws (the committed default, 0, is what CI runs)
SEED_OFFSET = int(__import__("os").environ.get("COMMS_TEST_SEED_OFFSET", "0"))


@pytest.fixture(scope="module")
def c():
    import comms_rs_amd as c

    assert c.device_count() >= 1, "no MI355X visible: the HIP path cannot be tested (no CPU fallback)"
    return c


def cuts_of(rng, n, k=2, multiple=1):
    pts = {0, n, *(int(v) // multiple * multiple for v in rng.integers(0, n + 1, k))}
    return sorted(pts)


def test_fir_random_sweep(c):
    rng = np.random.default_rng(11 + 1000 * SEED_OFFSET)
    algos = [c.FIR_AUTO, c.FIR_DIRECT, c.FIR_OS1024, c.FIR_OS4096, c.FIR_OS16K]
    for case in range(40):
        algo = algos[case % len(algos)]
        hi = {c.FIR_AUTO: 700, c.FIR_DIRECT: 300, c.FIR_OS1024: 257, c.FIR_OS4096: 700, c.FIR_OS16K: 900}[algo]
        n_taps = int(rng.integers(1, hi + 1))
        taps = rand_c(rng, n_taps) if rng.integers(0, 2) else rand_c(rng, n_taps).real.astype(np.complex64)
        n = int(rng.integers(1, 12000))
        x = rand_c(rng, n)
        # optional user state, shorter / equal / longer than the taps (zip truncation, fir.rs:53)
        state = None
        if rng.integers(0, 3) == 0:
            state = rand_c(rng, int(rng.integers(1, n_taps + 20)))
        node = c.BatchFirNode(taps, state)
        if algo != c.FIR_AUTO:
            node.set_algo(algo)
        ost = oracle.default_state(taps) if state is None else state.copy()
        got, want = [], []
        cuts = cuts_of(rng, n)
        for a, b in zip(cuts[:-1], cuts[1:]):
            got.append(node.run(x[a:b]))
            want.append(oracle.batch_fir(x[a:b], taps, ost))  # literal form: any state length (zip, fir.rs:53)
        fir_close(np.concatenate(got), np.concatenate(want), taps[:min(n_taps, ost.size)], x)


def test_mixer_decimate_upsample_fm_random_sweep(c):
    rng = np.random.default_rng(12 + 1000 * SEED_OFFSET)
    for case in range(30):
        n = int(rng.integers(1, 50000))
        x = rand_c(rng, n)
        dphase, phase = float(rng.uniform(-20, 20)), float(rng.uniform(-10, 10))
        gm, om = c.MixerNode(dphase, phase), oracle.Mixer(phase, dphase)
        gfm, ofm = c.FMDemodNode(), oracle.FM()
        for a, b in zip(*(lambda cs: (cs[:-1], cs[1:]))(cuts_of(rng, n))):
            g, w = gm.run(x[a:b]), om.mix(x[a:b])
            assert np.all(np.abs(g - w) <= TOL * np.abs(x[a:b]) + 1e-30), case
            gf, wf = gfm.run(x[a:b]), ofm.demod(x[a:b])
            # the angle of x[i] * conj(x[i-1]) is ill-conditioned where that product is ~0
            prod = np.abs(x[a:b]) * np.abs(np.concatenate([x[max(a - 1, 0):a] if a else [0], x[a:b - 1]]))
            ok = prod > 1e-3
            assert np.max(circ(gf.astype(np.float64) - wf)[ok], initial=0.0) <= 1e-5 / 1e-3 * 1e-2, case
        rate = int(rng.integers(0, 40))
        for dt in (np.complex64, np.float32, np.int16, np.uint8, np.complex128):
            v = (rng.integers(0, 200, n)).astype(dt)
            assert np.array_equal(c.DecimateNode(rate).run(v), oracle.decimate(v, rate)), (case, rate, dt)
            if n * max(rate, 1) <= 400000:
                assert np.array_equal(c.UpsampleNode(rate).run(v), oracle.upsample(v, rate)), (case, rate, dt)


def test_fft_random_sweep(c):
    rng = np.random.default_rng(13 + 1000 * SEED_OFFSET)
    sizes = [2 ** k for k in range(1, 18)] + [3, 6, 10, 12, 15, 60, 100, 360, 1000, 1536, 3000, 4097, 6000, 10000]
    for case in range(40):
        n = int(sizes[int(rng.integers(0, len(sizes)))])
        inverse = bool(rng.integers(0, 2))
        batch = int(rng.integers(1, max(2, min(600, (1 << 19) // n))))
        x = rand_c(rng, n * batch)
        got = c.FFTBatchNode(n, inverse).run(x)
        f = (lambda v: np.fft.ifft(v, axis=1) * n) if inverse else (lambda v: np.fft.fft(v, axis=1))
        want = f(x.astype(np.complex128).reshape(batch, n)).reshape(-1)
        fft_close(got, want)
        if n <= 2048:  # and the oracle's own f64 DFT on the first transform
            fft_close(got[:n], oracle.fft(x[:n], inverse))


def test_pulse_random_sweep(c):
    rng = np.random.default_rng(14 + 1000 * SEED_OFFSET)
    for case in range(30):
        sps = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 20, 32]))
        n_taps = int(rng.integers(1, 400))
        taps = rand_c(rng, n_taps) if rng.integers(0, 2) else rand_c(rng, n_taps).real.astype(np.complex64)
        n = int(rng.integers(1, 4000))
        sym = rand_c(rng, n)
        node, st = c.PulseNode(taps, sps), oracle.default_state(taps)
        mixed = rng.integers(0, 2) == 1  # half the cases with the mixer fused in (transmit chain)
        if mixed:
            dphase, phase = float(rng.uniform(-20, 20)), float(rng.uniform(-10, 10))
            node.set_mixer(dphase, phase)
            omix = oracle.Mixer(phase, dphase)
        got, want = [], []
        for a, b in zip(*(lambda cs: (cs[:-1], cs[1:]))(cuts_of(rng, n))):
            got.append(node.run(sym[a:b]))
            w = oracle.pulse(sym[a:b], taps, sps, st)
            want.append(omix.mix(w) if mixed else w)
        fir_close(np.concatenate(got), np.concatenate(want), taps, sym)


def test_fir_ticketed_kernel_random_sweep(c):
    """Long batches (>= 4096 segments) run fir_os1024_dyn_kernel: random tap counts (both halo widths), batch lengths
    that end on and off a segment boundary, two calls with carried state -- bit-identical to the fixed-run kernel, and
    equal to the oracle on windows at the start, the call boundary and the end."""
    import torch

    rng = np.random.default_rng(15 + 1000 * SEED_OFFSET)
    s = torch.cuda.current_stream().cuda_stream
    for case in range(6):
        n_taps = int(rng.integers(9, 258))
        wv = 1024 - 64 * (1 if n_taps <= 65 else 2 if n_taps <= 129 else 3 if n_taps <= 193 else 4)
        n1 = int(rng.integers(4096, 5200)) * wv + (0 if case % 3 == 0 else int(rng.integers(1, wv)))
        n2 = int(rng.integers(4096, 4500)) * wv + int(rng.integers(0, 2)) * int(rng.integers(1, wv))
        taps = rand_c(rng, n_taps) if case % 2 else rand_c(rng, n_taps).real.astype(np.complex64)
        seed = int(rng.integers(1, 1 << 30))
        x = torch.empty(n1 + n2, dtype=torch.complex64, device="cuda:0")
        c.synth_iq_dev(x.data_ptr(), n1 + n2, 0, seed)
        ys = []
        for mode in (0, 1):
            node = c.BatchFirNode(taps).set_algo(c.FIR_OS1024 if mode else c.FIR_OS1024_FIXED)
            assert node.kernel_for(n1) == ("fir_os1024_dyn_kernel" if mode else "fir_os1024_kernel"), case
            y = torch.empty_like(x)
            node.run_dev(x.data_ptr(), n1, y.data_ptr(), s)
            node.run_dev(x.data_ptr() + 8 * n1, n2, y.data_ptr() + 8 * n1, s)
            torch.cuda.synchronize()
            ys.append(y)
        assert torch.equal(torch.view_as_real(ys[0]), torch.view_as_real(ys[1])), (case, n_taps, n1, n2)
        for a in (0, n1 - 2000, n1 + n2 - 3000):
            lo = max(0, a - (n_taps - 1))
            xs = c.synth_iq(a + 3000 - lo, lo, seed)
            want = oracle.batch_fir(xs, taps, oracle.default_state(taps), norotate=True)[a - lo:]
            fir_close(ys[1][a:a + 3000].cpu().numpy(), want, taps, xs)
        del x, ys


def test_poly8_chain_random_sweep(c):
    """fir_poly8_kernel on random draws: rate (8, 4 and the multiples of 4 up to 64), tap count (every halo width, real / complex
    taps), mixer order, FM demod (rates 8 and 4), oscillator, user FIR state, batch cuts on multiples of the rate -- against the
    oracle's nodes in series."""
    import os

    rng = np.random.default_rng(16 + 1000 * SEED_OFFSET)
    for case in range(int(os.environ.get("COMMS_TEST_POLY_CASES", "60"))):  # (more draws for a one-off soak)
        rate = int(rng.choice([8, 8, 8, 4, 4, 12, 16, 20, 24, 28, 32, 40, 48, 56, 60, 64]))
        fm = bool(rng.integers(0, 2)) and rate in (4, 8)
        n_taps = int(rng.integers(1, (505 if fm else 513) + 1))
        taps = lowpass_taps(n_taps, float(rng.uniform(0.02, 0.06)))
        if rng.integers(0, 2):
            taps = (taps * np.exp(1j * rng.uniform(-0.05, 0.05) * np.arange(n_taps))).astype(np.complex64)
        after = bool(rng.integers(0, 2)) and not fm
        dphase, phase = float(rng.uniform(-3, 3)), float(rng.uniform(-3, 3))
        n = rate * int(rng.integers(1, 24000 // rate))
        t = np.arange(n)
        x = (np.exp(1j * (0.3 * dphase * t + 2.0 * np.sin(t / 700.0))) * (1 + 0.05 * rng.standard_normal(n))).astype(np.complex64)
        node = c.ChainNode(dphase, phase, taps, rate, fm, mixer_after_fir=after, kernel="poly")
        assert node.kernel == "poly", case
        ost, om, ofm = oracle.default_state(taps), oracle.Mixer(phase, dphase), oracle.FM()
        if rng.integers(0, 3) == 0:  # a user state: the halo of a sharded stream (raw samples, newest first)
            st = rand_c(rng, n_taps)
            node.set_fir_state(st)
            ost = (om_state := st.copy())
            if not after:  # mixer first: the reference's FIR state holds MIXED samples, the oscillator running backwards from `phase`
                k = np.arange(1, n_taps + 1)
                ost = (st.astype(np.complex128) * np.exp(1j * (phase - k * dphase))).astype(np.complex64)
        scale = np.sum(np.abs(taps)) * max(np.max(np.abs(x)), 1.0)
        last = 0j
        for a, b in zip(*(lambda cs: (cs[:-1], cs[1:]))(cuts_of(rng, n, 3, rate))):
            if after:
                y = oracle.decimate(om.mix(oracle.batch_fir(x[a:b], taps, ost, norotate=True)), rate)
            else:
                y = oracle.decimate(oracle.batch_fir(om.mix(x[a:b]), taps, ost, norotate=True), rate)
            got = node.run(x[a:b])
            if fm:
                w = ofm.demod(y)
                mag = np.minimum(np.abs(y), np.abs(np.concatenate([[last], y[:-1]])))
                last = y[-1]
                assert np.max(circ(got.astype(np.float64) - w) * mag) <= 4 * TOL * scale, (case, rate, n_taps, a, b)
            else:
                assert np.max(np.abs(got - y)) <= 2 * TOL * scale, (case, rate, n_taps, after, a, b)


def test_long_chain_random_sweep(c):
    """The decimating 4096- / 16384-point overlap-save kernels (258 ... 1537 / 1538 ... 4097 taps, mixer and decimator in the store stage) on random draws:
    rate (2 ... 3000, the polyphase kernel's rates excluded below 514 taps by the chain itself), taps, mixer order, FM demod (a second
    launch), oscillator, user FIR state, batch cuts on multiples of the rate -- against the oracle's nodes in series.
    COMMS_TEST_LONG_CASES: more draws (the default keeps the suite short)."""
    import os

    rng = np.random.default_rng(23 + 1000 * SEED_OFFSET)
    for case in range(int(os.environ.get("COMMS_TEST_LONG_CASES", "40"))):
        rate = int(rng.choice([2, 3, 5, 6, 7, 9, 10, 11, 13, 14, 15, 17, 25, 50, 100, 255, 256, 257, 1000, int(rng.integers(2, 3000))]))
        n_taps = int(rng.integers(258, 514)) if case % 3 == 0 else int(rng.integers(514, 1538)) if case % 3 == 1 else int(rng.integers(1538, 4098))
        fm = bool(rng.integers(0, 3) == 0)
        taps = lowpass_taps(n_taps, float(rng.uniform(0.01, 0.05)))
        if rng.integers(0, 2):
            taps = (taps * np.exp(1j * rng.uniform(-0.05, 0.05) * np.arange(n_taps))).astype(np.complex64)
        after = bool(rng.integers(0, 2)) and not fm
        dphase, phase = float(rng.uniform(-3, 3)), float(rng.uniform(-3, 3))
        n = rate * int(rng.integers(1, max(2, (30000 if n_taps <= 1537 else 60000) // rate)))
        t = np.arange(n)
        x = (np.exp(1j * (0.3 * dphase * t + 2.0 * np.sin(t / 700.0))) * (1 + 0.05 * rng.standard_normal(n))).astype(np.complex64)
        node = c.ChainNode(dphase, phase, taps, rate, fm, mixer_after_fir=after, kernel="freq")
        assert node.kernel == "freq", (case, rate, n_taps)
        ost, om, ofm = oracle.default_state(taps), oracle.Mixer(phase, dphase), oracle.FM()
        if rng.integers(0, 3) == 0:  # a user state: the halo of a sharded stream (raw samples, newest first)
            st = rand_c(rng, n_taps)
            node.set_fir_state(st)
            ost = st.copy()
            if not after:  # mixer first: the reference's FIR state holds MIXED samples, the oscillator running backwards from `phase`
                k = np.arange(1, n_taps + 1)
                ost = (st.astype(np.complex128) * np.exp(1j * (phase - k * dphase))).astype(np.complex64)
        scale = np.sum(np.abs(taps)) * max(np.max(np.abs(x)), 1.0)
        last = 0j
        for a, b in zip(*(lambda cs: (cs[:-1], cs[1:]))(cuts_of(rng, n, 3, rate))):
            if after:
                y = oracle.decimate(om.mix(oracle.batch_fir(x[a:b], taps, ost, norotate=True)), rate)
            else:
                y = oracle.decimate(oracle.batch_fir(om.mix(x[a:b]), taps, ost, norotate=True), rate)
            got = node.run(x[a:b])
            if fm:
                w = ofm.demod(y)
                mag = np.minimum(np.abs(y), np.abs(np.concatenate([[last], y[:-1]])))
                last = y[-1]
                assert np.max(circ(got.astype(np.float64) - w) * mag, initial=0.0) <= 4 * TOL * scale, (case, rate, n_taps, a, b)
            else:
                assert np.max(np.abs(got - y), initial=0.0) <= 2 * TOL * scale, (case, rate, n_taps, after, a, b)


def test_poly8_ticketed_batches_random_sweep(c):
    """Long batches (thousands of segments drawn from the ticket counter, both chunk sizes of the dealing): random tap counts and
    lengths on and off a segment boundary, two calls with carried state -- against the overlap-save fusion (fir_os1024_kernel<MODE>,
    another algorithm: full-rate transforms, seven of eight outputs dropped) on every output."""
    import torch

    rng = np.random.default_rng(17 + 1000 * SEED_OFFSET)
    s = torch.cuda.current_stream().cuda_stream
    for case in range(6):
        fm = case % 3 == 2
        n_taps = int(rng.integers(9, (505 if fm else 513) + 1))
        rows = max(2, (n_taps + (8 if fm else 0) - 1 + 63) // 64)
        new = 1024 - 64 * (8 if rows == 7 else rows)
        n1 = int(rng.integers(4096, 5200)) * new + (0 if case % 2 == 0 else 8 * int(rng.integers(1, new // 8)))
        n2 = (int(rng.integers(40000, 45000)) if case == 5 else int(rng.integers(300, 4500))) * new + 8 * int(rng.integers(0, new // 8))
        taps = lowpass_taps(n_taps, 1 / 20.0)
        x = torch.empty(n1 + n2, dtype=torch.complex64, device="cuda:0")
        c.synth_iq_dev(x.data_ptr(), n1 + n2, 0, int(rng.integers(1, 1 << 30)))
        ys = []
        for kern in ("poly", "freq"):
            node = c.ChainNode(0.7, 0.2, taps, 8, fm, mixer_after_fir=not fm and case % 2 == 0, kernel=kern)
            y = torch.empty((n1 + n2) // 8, dtype=torch.float32 if fm else torch.complex64, device="cuda:0")
            node.run_dev(x.data_ptr(), n1, y.data_ptr(), s)
            node.run_dev(x.data_ptr() + 8 * n1, n2, y.data_ptr() + (4 if fm else 8) * (n1 // 8), s)
            torch.cuda.synchronize()
            ys.append(y)
        d = (ys[0] - ys[1]).abs()
        bound = 2 * TOL * float(np.sum(np.abs(taps))) * x[: 1 << 20].abs().max().item()
        if fm:  # angles: on the circle, and only a statistic -- the filtered synthetic stream passes close to zero now and then
            d = torch.minimum(d, 2 * np.pi - d)
            assert d.median().item() <= 1e-5 and (d > 1e-2).float().mean().item() <= 1e-4, (case, n_taps, d.max().item())
        else:
            assert d.max().item() <= 2 * bound, (case, n_taps, d.max().item(), bound)
        del x, ys
