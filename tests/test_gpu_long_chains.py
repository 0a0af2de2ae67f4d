"""GPU parity tests of the decimating chains of long filters on the 4096- and 16384-point overlap-save kernels (csrc/fir.hip,
fir_os4096_kernel<.., DEC> / fir_os16k_kernel<.., DEC>: mixer and decimator in the store stage; 258 ... 1537 / 1538 ... 4097 taps at
the rates the polyphase kernel does not run): BatchFirNode (src/filter/fir.rs:87-102), MixerNode (src/mixer.rs:73-85), DecimateNode (src/util/resample_node.rs:53-65)
[, FMDemodNode (src/modulation/analog.rs:22-35)] in either mixer order against the oracle's nodes in series.

Tolerances as for every other chain kernel: a chain's decimated output max|d| <= 2 * 1e-5 * sum|taps| * max|x| (FIR error plus the
mixer's rounding of it); a demodulated angle weighted by the smaller of the two samples it is the argument of."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

TOL = 1e-5


@pytest.fixture(scope="module")
def c():
    import comms_rs_amd as c

    assert c.device_count() >= 1, "no MI355X visible: the HIP path cannot be tested (no CPU fallback)"
    return c


def rand_c(rng, n):
    return (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)


def lpf(n_taps, cutoff):
    k = np.arange(n_taps) - (n_taps - 1) / 2.0
    return (2 * cutoff * np.sinc(2 * cutoff * k) * np.hamming(n_taps)).astype(np.complex64)


def circ(d):
    return np.abs((d + np.pi) % (2 * np.pi) - np.pi)


def chain_close(got, want, taps, x):
    d = np.abs(got.astype(np.complex128) - want.astype(np.complex128))
    bound = 2 * TOL * np.sum(np.abs(taps)) * max(np.max(np.abs(x)), 1e-30)
    assert got.shape == want.shape
    assert d.max(initial=0.0) <= bound, (d.max(), bound, int(np.argmax(d)))


def oracle_chain(x, taps, ost, om, rate, after):
    if after:
        return oracle.decimate(om.mix(oracle.batch_fir(x, taps, ost, norotate=True)), rate)
    return oracle.decimate(oracle.batch_fir(om.mix(x), taps, ost, norotate=True), rate)


# segments of 4096 - 256 ceil((taps - 1) / 256) outputs; the kept outputs of a row of 256 are every rate-th lane's (rates below
# 256), or one lane's of some rows (rates above)
@pytest.mark.parametrize("after", [False, True])
@pytest.mark.parametrize("n_taps,rate,cplx", [(258, 2, False), (300, 3, True), (383, 5, False), (511, 7, True), (514, 8, False), (600, 10, True),
                                              (769, 16, False), (1025, 4, True), (1281, 9, False), (1537, 6, True), (700, 100, False),
                                              (900, 255, True), (600, 256, False), (1000, 257, False), (520, 1000, True), (600, 5000, False),
                                              (1538, 2, False), (2049, 5, True), (2050, 8, False), (3073, 3, True), (3074, 10, False),
                                              (4097, 7, True), (2500, 1024, False), (3000, 1025, True), (4000, 100, False)])
def test_long_chain_against_oracle_ragged_calls(c, n_taps, rate, cplx, after):
    rng = np.random.default_rng(n_taps + rate)
    taps = lpf(n_taps, 0.4 / max(rate, 2) if rate < 50 else 0.01)
    if cplx:
        taps = (taps * np.exp(1j * 0.01 * np.arange(n_taps))).astype(np.complex64)
    dphase, phase = 2 * np.pi * 0.07, 0.3
    node = c.ChainNode(dphase, phase, taps, rate, False, mixer_after_fir=after)
    assert node.fused and node.kernel == "freq"
    unit = rate
    n = unit * (((3840 * 5 if n_taps <= 1537 else 12288 * 3) + 777) // unit + 3)
    x = rand_c(rng, n)
    ost, om = oracle.default_state(taps), oracle.Mixer(phase, dphase)
    cuts = [0, unit, unit * 2, unit * (3000 // unit + 1), unit * (9000 // unit + 1), n]
    cuts = sorted(set(min(v, n) for v in cuts))
    for a, b in zip(cuts[:-1], cuts[1:]):
        chain_close(node.run(x[a:b]), oracle_chain(x[a:b], taps, ost, om, rate, after), taps, x)
    # the FIR history is the reference's `state` (newest first), raw samples in either mixer order; the oscillator has advanced n steps
    np.testing.assert_array_equal(node.fir_state(n_taps), x[::-1][:n_taps])
    want_phase = (phase + n * dphase) % (2 * np.pi)
    assert abs(((node.phase - want_phase) + np.pi) % (2 * np.pi) - np.pi) < 1e-6


@pytest.mark.parametrize("n_taps,rate", [(300, 5), (600, 8), (1025, 10), (1537, 3), (700, 40), (2049, 5), (4097, 16)])
def test_long_fm_chain_against_oracle(c, n_taps, rate):
    """mixer -> FIR -> /R -> FM demod: the demodulator is FMDemodNode's kernel over the kept samples, FM.prev carried."""
    rng = np.random.default_rng(n_taps)
    taps = lpf(n_taps, 1 / (2.5 * rate))
    n = rate * (30000 // rate + 5)
    t = np.arange(n)
    x = (np.exp(1j * (0.02 * t + 3.0 * np.sin(2 * np.pi * t / 5000.0))) * (1 + 0.1 * rng.standard_normal(n))).astype(np.complex64)
    node = c.ChainNode(0.3, 0.1, taps, rate, True)
    assert node.kernel == "freq"
    ost, om, ofm = oracle.default_state(taps), oracle.Mixer(0.1, 0.3), oracle.FM()
    cuts = [0, rate, rate * 3, rate * (9000 // rate), n]
    last = 0j
    for a, b in zip(cuts[:-1], cuts[1:]):
        y = oracle.decimate(oracle.batch_fir(om.mix(x[a:b]), taps, ost, norotate=True), rate)
        w = ofm.demod(y)
        got = node.run(x[a:b])
        assert got.dtype == np.float32 and got.shape == w.shape
        mag = np.minimum(np.abs(y), np.abs(np.concatenate([[last], y[:-1]])))
        last = y[-1]
        assert np.max(circ(got.astype(np.float64) - w) * mag) <= 4 * TOL * np.sum(np.abs(taps)) * np.max(np.abs(x))
    assert abs(complex(node.fm_prev) - complex(y[-1])) <= 2 * TOL * np.sum(np.abs(taps)) * np.max(np.abs(x))
    # checkpoint: a second node restored from the first one's state continues identically
    node2 = c.ChainNode(0.3, 0.1, taps, rate, True)
    node2.set_fir_state(node.fir_state(n_taps))
    node2.phase = node.phase
    node2.fm_prev = node.fm_prev
    tail = x[: rate * 300]
    g1, g2 = node.run(tail), node2.run(tail)
    assert np.max(circ(g1.astype(np.float64) - g2.astype(np.float64))) <= 1e-4


@pytest.mark.parametrize("fmt", ["i16", "u8"])
@pytest.mark.parametrize("n_taps,rate", [(400, 5), (1000, 8), (2000, 6)])
def test_long_chain_reads_raw_iq(c, fmt, n_taps, rate):
    """Raw i16 / u8 IQ converted in the kernel's load stage (the 16384-point kernel: in a pass in front of it): the same bits as
    the chain over the converted samples."""
    idx = np.arange(rate * 5000, dtype=np.float64)
    z = np.exp(1j * (-2 * np.pi * 0.05 * idx + 8.0 * np.cos(2 * np.pi * idx / 4096)))
    if fmt == "u8":
        raw = np.stack([np.clip(np.rint(z.real * 100 + 127.5), 0, 255), np.clip(np.rint(z.imag * 100 + 127.5), 0, 255)], 1).astype(np.uint8)
        x = oracle.iq_u8_to_c32(raw)
    else:
        raw = np.stack([np.rint(z.real * 8192), np.rint(z.imag * 8192)], 1).astype(np.int16)
        raw[:4] = [[-32768, 32767], [0, -1], [1, 0], [12345, -12345]]
        x = oracle.iq_i16_to_c32(raw, 1.0 / 8192)
    n = x.size
    taps = lpf(n_taps, 1 / (2.5 * rate))
    a = c.ChainNode(0.05, 0.3, taps, rate, False).set_input_format(fmt, 1.0 / 8192)
    b = c.ChainNode(0.05, 0.3, taps, rate, False)
    assert a.kernel == "freq" and b.kernel == "freq"
    cuts = [0, rate, rate * 700, rate * 701, n]
    got = np.concatenate([a.run(raw[p:q]) for p, q in zip(cuts[:-1], cuts[1:])])
    ref = np.concatenate([b.run(x[p:q]) for p, q in zip(cuts[:-1], cuts[1:])])
    assert np.array_equal(got, ref)
    y = oracle.decimate(oracle.batch_fir(oracle.Mixer(0.3, 0.05).mix(x), taps, oracle.default_state(taps), norotate=True), rate)
    chain_close(got, y, taps, x)


@pytest.mark.parametrize("n_taps", [769, 2500])
def test_long_chain_at_2p24_matches_the_series_of_launches(c, n_taps):
    """2^24 samples (every workgroup of the persistent grid walks several segments: the carried output index), 769 / 2500 taps, rate 5:
    the one launch against the chain as a series of launches (FIR, mixer-decimator), which the tests above and
    tests/test_gpu_parity.py hold to the oracle."""
    import torch

    n = 5 * ((1 << 24) // 5)
    taps = lpf(n_taps, 0.08)
    xd = torch.empty(n, dtype=torch.complex64, device="cuda:0")
    c.synth_iq_dev(xd.data_ptr(), n, 0, 77)
    s = torch.cuda.current_stream().cuda_stream
    ys = []
    for kw in (dict(), dict(unfused=True)):
        node = c.ChainNode(2 * np.pi * 0.05, 0.2, taps, 5, False, mixer_after_fir=True, **kw)
        assert node.kernel == ("unfused" if kw else "freq")
        y = torch.empty(n // 5, dtype=torch.complex64, device="cuda:0")
        for _ in range(2):  # (the second call continues the stream)
            node.run_dev(xd.data_ptr(), n, y.data_ptr(), s)
        torch.cuda.synchronize()
        ys.append(y)
    bound = 2 * TOL * float(np.sum(np.abs(taps))) * xd[: 1 << 20].abs().max().item()
    assert (ys[0] - ys[1]).abs().max().item() <= 2 * bound
    assert ys[0].abs().max().item() > 0.01


@pytest.mark.parametrize("fm", [False, True])
@pytest.mark.parametrize("rate,n_taps,cplx", [(2, 31, False), (2, 64, True), (2, 129, False), (4, 63, False), (4, 128, True)])
def test_wave_private_kernel_at_rates_2_and_4(c, rate, n_taps, cplx, fm):
    """fir_decim_wave_kernel at rates 2 and 4 (round 5; rate 8 since its first form): a batch whose 8000 tiles of 128 outputs spread
    evenly over the chip's waves (two per wave, the last run ragged) reaches it in the product build.  Against the overlap-save
    fusion (another algorithm) on every output, two calls with the state carried; the first 2^15 outputs against the oracle."""
    import torch

    n_out = 128 * 8000 - 37
    n = rate * n_out
    taps = lpf(n_taps, 0.4 / rate)
    if cplx:
        taps = (taps * np.exp(1j * 0.01 * np.arange(n_taps))).astype(np.complex64)
    xd = torch.empty(n, dtype=torch.complex64, device="cuda:0")
    c.synth_iq_dev(xd.data_ptr(), n, 0, 1234 + rate)
    s = torch.cuda.current_stream().cuda_stream
    dphase, phase = 2 * np.pi * 0.013, 0.2
    ys = []
    for kern in ("time", "freq"):
        node = c.ChainNode(dphase, phase, taps, rate, fm, mixer_after_fir=not fm, kernel=kern)
        assert node.kernel == kern
        outs = []
        for _ in range(2):  # (the second call continues the stream)
            y = torch.empty(n_out, dtype=torch.float32 if fm else torch.complex64, device="cuda:0")
            node.run_dev(xd.data_ptr(), n, y.data_ptr(), s)
            outs.append(y)
        torch.cuda.synchronize()
        ys.append(outs)
    scale = float(np.sum(np.abs(taps))) * xd[: 1 << 20].abs().max().item()
    x_head = xd[: rate * (1 << 15)].cpu().numpy()
    om = oracle.Mixer(phase, dphase)
    if fm:
        yo = oracle.decimate(oracle.batch_fir(om.mix(x_head), taps, oracle.default_state(taps), norotate=True), rate)
        w = oracle.FM().demod(yo)
        mag = np.minimum(np.abs(yo), np.abs(np.concatenate([[0.0], yo[:-1]])))
        got = ys[0][0][: 1 << 15].cpu().numpy()
        assert np.max(circ(got.astype(np.float64) - w) * mag) <= 4 * TOL * scale
        for a, b in zip(*ys):  # the two kernels on every output, on the circle; the angle is ill-conditioned where the signal is ~0
            d = torch.remainder(a.double() - b.double() + np.pi, 2 * np.pi) - np.pi
            assert torch.quantile(d.abs()[:: 64], 0.999).item() <= 1e-3
    else:
        yo = oracle.decimate(om.mix(oracle.batch_fir(x_head, taps, oracle.default_state(taps), norotate=True)), rate)
        chain_close(ys[0][0][: 1 << 15].cpu().numpy(), yo, taps, x_head)
        for a, b in zip(*ys):
            assert (a - b).abs().max().item() <= 4 * TOL * scale
