"""GPU parity tests of fir_poly8_kernel (csrc/fir_poly8.hip): the rate-8 chains -- BatchFirNode (src/filter/fir.rs:87-102),
MixerNode (src/mixer.rs:73-85), DecimateNode (src/util/resample_node.rs:53-65) [, FMDemodNode (src/modulation/analog.rs:22-35)]
in either mixer order -- computed as eight polyphase branches in the frequency domain, against the oracle's nodes in series.

Tolerances as for every other chain kernel (tests/test_gpu_parity.py): a chain's decimated output max|d| <= 2 * 1e-5 * sum|taps| *
max|x| (FIR error plus the mixer's rounding of it); a demodulated angle <= 1e-4 rad on the circle where both samples it is the
argument of exceed 0.05 (it is ill-conditioned where the filtered signal is ~0)."""
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


# halo rows by tap count: <= 129 taps two, <= 193 three, <= 257 four, <= 321 five, <= 385 six, <= 513 eight; ragged calls cross
# segment sizes 896 / 832 / 768 / 704 / 640 / 512
@pytest.mark.parametrize("after", [False, True])
@pytest.mark.parametrize("n_taps,cplx", [(255, False), (255, True), (257, True), (256, False), (194, False), (193, True), (131, False),
                                         (130, True), (129, False), (127, False), (65, True), (33, False), (8, True), (1, False),
                                         (258, True), (300, False), (321, True), (322, False), (385, True), (386, False), (449, False),
                                         (511, True), (513, False)])
def test_poly8_chain_against_oracle_ragged_calls(c, n_taps, cplx, after):
    rng = np.random.default_rng(n_taps * 2 + cplx)
    taps = oracle.rrc_taps(n_taps, 8.0, 0.35) if n_taps > 8 else np.ones(n_taps, np.complex64)
    if cplx:
        taps = (taps * np.exp(1j * 0.01 * np.arange(n_taps))).astype(np.complex64)
    dphase, phase = 2 * np.pi * 0.1, 0.3
    node = c.ChainNode(dphase, phase, taps, 8, False, mixer_after_fir=after, kernel="poly")
    assert node.fused and node.kernel == "poly"
    n = 896 * 37 + 8 * 11
    x = rand_c(rng, n)
    ost, om = oracle.default_state(taps), oracle.Mixer(phase, dphase)
    cuts = [0, 8, 776, 768 * 3, 768 * 3 + 16, 704 * 7 + 8, 640 * 12, 512 * 19 + 24, 832 * 20 + 8 * 50, n]
    for a, b in zip(cuts[:-1], cuts[1:]):
        if after:
            w = oracle.decimate(om.mix(oracle.batch_fir(x[a:b], taps, ost, norotate=True)), 8)
        else:
            w = oracle.decimate(oracle.batch_fir(om.mix(x[a:b]), taps, ost, norotate=True), 8)
        chain_close(node.run(x[a:b]), w, taps, x)
    # the FIR history is the reference's `state` (newest first), raw samples in either mixer order
    np.testing.assert_array_equal(node.fir_state(n_taps), x[::-1][:n_taps])


@pytest.mark.parametrize("rate", [8, 4])
@pytest.mark.parametrize("n_taps", [127, 63, 9, 121, 122, 185, 186, 249, 250, 255, 313, 314, 377, 378, 441, 505])
def test_poly8_fm_chain_against_oracle(c, n_taps, rate):
    """mixer -> FIR -> /8 -> FM demod (BASELINE config 3's order): the demodulator takes y[j-1] from the lane below, the
    segment's first output from the halo position in front of it (121 / 185 / 249 taps: the last counts with that spare
    position), the call's first from FM.prev.  At rate 4 the kept stream interleaves the two output phases: y[j-1] of the first
    phase is the second phase of the lane below."""
    rng = np.random.default_rng(n_taps)
    taps = lpf(n_taps, 1 / 16.0)
    n = 896 * 30 + 8 * 7 + (4 if rate == 4 else 0)
    t = np.arange(n)
    x = (np.exp(1j * (0.02 * t + 3.0 * np.sin(2 * np.pi * t / 5000.0))) * (1 + 0.1 * rng.standard_normal(n))).astype(np.complex64)
    node = c.ChainNode(0.3, 0.1, taps, rate, True, kernel="poly")
    assert node.kernel == "poly"
    ost, om, ofm = oracle.default_state(taps), oracle.Mixer(0.1, 0.3), oracle.FM()
    cuts = [0, 8, 16, 896, 896 * 3 + 24, 512 * 13, 832 * 11, n] if rate == 8 else [0, 4, 8, 20, 896, 896 * 3 + 20, 512 * 13 + 4, 832 * 11 + 4, n]
    for a, b in zip(cuts[:-1], cuts[1:]):
        y = oracle.decimate(oracle.batch_fir(om.mix(x[a:b]), taps, ost, norotate=True), rate)
        yp = np.concatenate([[complex(ofm.prev[0]) if a else 1.0], y[:-1]])
        w = ofm.demod(y)
        got = node.run(x[a:b])
        assert got.dtype == np.float32 and got.shape == w.shape
        ok = np.minimum(np.abs(y), np.abs(yp)) > 0.05
        assert ok.mean() > 0.9 or b - a < 8 * 200  # (the zero-state start of the stream: the filter has not filled yet)
        assert np.max(circ(got.astype(np.float64) - w)[ok], initial=0.0) <= 1e-4
    p = node.fm_prev
    assert abs(complex(p) - complex(y[-1])) <= 2 * TOL * np.sum(np.abs(taps)) * np.max(np.abs(x))


def test_poly8_where_it_runs(c):
    t255, t127, t31 = lpf(255, 0.05), lpf(127, 0.05), lpf(31, 0.05)
    assert c.ChainNode(0.3, 0.1, t255, 8, False, kernel="poly").kernel == "poly"
    assert c.ChainNode(0.3, 0.1, lpf(249, 0.05), 8, True, kernel="poly").kernel == "poly"
    assert c.ChainNode(0.3, 0.1, lpf(250, 0.05), 8, True, kernel="poly").kernel == "poly"  # (a fifth halo row)
    assert c.ChainNode(0.3, 0.1, lpf(513, 0.05), 8, False, kernel="poly").kernel == "poly"
    assert c.ChainNode(0.3, 0.1, lpf(505, 0.05), 4, True, kernel="poly").kernel == "poly"
    assert c.ChainNode(0.3, 0.1, lpf(506, 0.05), 8, True, kernel="poly").kernel == "poly"  # (no spare halo position: demodulator behind it)
    # more than 513 taps / another rate: the flag asks, the chain falls back
    assert c.ChainNode(0.3, 0.1, lpf(514, 0.05), 8, False, kernel="poly").kernel != "poly"
    assert c.ChainNode(0.3, 0.1, lpf(514, 0.05), 8, True, kernel="poly").kernel != "poly"
    # 258 ... 513 taps: the chain picks it by itself (the alternative is the overlap-save FIR + a mixer-decimator), with the
    # demodulator as its own launch where the kernel has none (other rates; past 505 taps)
    for rate, fm, nt in [(8, False, 300), (8, True, 400), (4, False, 511), (16, False, 385), (32, True, 300), (8, True, 510), (12, False, 300)]:
        assert c.ChainNode(0.3, 0.1, lpf(nt, 0.05), rate, fm).kernel == "poly", (rate, fm, nt)
    assert c.ChainNode(0.3, 0.1, lpf(300, 0.05), 8, False, kernel="time").kernel != "poly"
    assert c.ChainNode(0.3, 0.1, lpf(300, 0.05), 28, False).kernel == "poly"  # (rates 4 m', m' odd: up to 36 with such filters)
    assert c.ChainNode(0.3, 0.1, lpf(300, 0.05), 44, False).kernel == "time_any"
    assert c.ChainNode(0.3, 0.1, t127, 5, False, kernel="poly").kernel != "poly"
    assert c.ChainNode(0.3, 0.1, t127, 72, False, kernel="poly").kernel != "poly"
    assert c.ChainNode(0.3, 0.1, t127, 4, True, kernel="poly").kernel == "poly"  # FM demod in the kernel: rates 8 and 4
    assert c.ChainNode(0.3, 0.1, t127, 16, True, kernel="poly").kernel != "poly"
    # rates 4 (two output phases) and 8 m up to 64 (every m-th output of the rate-8 form)
    for rate in (4, 16, 24, 32, 40, 48, 56, 64, 12, 20, 60):
        assert c.ChainNode(0.3, 0.1, t127, rate, False, kernel="poly").kernel == "poly"
    rng = np.random.default_rng(3)
    x = rand_c(rng, 8 * 4096)
    # a chain of kind "time" at rate 8: from 64 taps every call runs on the polyphase kernel, and says so afterwards
    for taps, fm, want in [(t255, False, "poly"), (t127, True, "poly"), (t31, False, "time")]:
        node = c.ChainNode(0.3, 0.1, taps, 8, fm)
        assert node.kernel == "time"
        node.run(x)
        assert node.kernel == want
    node = c.ChainNode(0.3, 0.1, t255, 8, False, kernel="time")  # forced: the time-domain kernel on every call
    node.run(x)
    assert node.kernel == "time"
    node = c.ChainNode(0.3, 0.1, t255, 8, False).set_input_format("i16", 1.0 / 32768)  # raw formats: converted in its load stage
    node.run((rng.integers(-2000, 2000, (8 * 512, 2))).astype(np.int16))
    assert node.kernel == "poly"
    with pytest.raises(c.CommsError):
        c.ChainNode(0.3, 0.1, t255, 8, False, kernel="poly").run(x[:12])  # n not a multiple of the rate


@pytest.mark.parametrize("fm", [False, True])
def test_poly8_short_calls_and_single_outputs(c, fm):
    """Calls of one output (8 samples), of less than a segment, of exactly one segment: the guarded first / last segments."""
    rng = np.random.default_rng(11)
    taps = lpf(201, 0.05)
    x = rand_c(rng, 8 * 700)
    node = c.ChainNode(0.2, 0.0, taps, 8, fm, kernel="poly")
    ref = c.ChainNode(0.2, 0.0, taps, 8, fm, kernel="freq")
    cref = c.ChainNode(0.2, 0.0, taps, 8, False, kernel="freq")  # the decimated filter output the angles are arguments of
    pos, last = 0, 0j
    for m in (8, 8, 16, 768, 776, 760, 8, 1536, 8 * 100):
        a, b, y = node.run(x[pos:pos + m]), ref.run(x[pos:pos + m]), cref.run(x[pos:pos + m])
        pos += m
        if fm:
            mag = np.minimum(np.abs(y), np.abs(np.concatenate([[last], y[:-1]])))
            last = y[-1]
            bound = 4 * TOL * np.sum(np.abs(taps)) * np.max(np.abs(x))  # conditioned: error of the angle x the smaller magnitude
            assert np.max(circ(a.astype(np.float64) - b) * mag) <= bound
        else:
            chain_close(a, b, taps, x)
    assert pos <= x.size


def test_poly8_is_deterministic_and_matches_the_time_kernel_at_2p24(c):
    """Segments are drawn from a ticket counter in whatever order the waves arrive: the outputs do not depend on it."""
    import torch

    n = 1 << 24
    x = torch.empty(n, dtype=torch.complex64, device="cuda:0")
    c.synth_iq_dev(x.data_ptr(), n, 0)
    taps = oracle.rrc_taps(255, 8.0, 0.35)
    s = torch.cuda.current_stream().cuda_stream
    outs = []
    for kern in ("poly", "poly", "time"):
        node = c.ChainNode(2 * np.pi * 0.05, 0.1, taps, 8, False, mixer_after_fir=True, kernel=kern)
        o = torch.empty(n // 8, dtype=torch.complex64, device="cuda:0")
        node.run_dev(x.data_ptr(), n, o.data_ptr(), s)
        torch.cuda.synchronize()
        outs.append(o)
    assert torch.equal(outs[0], outs[1])
    d = (outs[0] - outs[2]).abs().max().item()
    assert d <= 2 * TOL * np.sum(np.abs(taps)) * x[: 1 << 20].abs().max().item()


def test_poly8_shard_continues_its_left_neighbour(c):
    """A stream cut in two: the right half's node starts from the left half's FIR history and oscillator phase (and FM.prev)
    and reproduces the single node's outputs (another segmentation: within the tolerance, not bit for bit)."""
    rng = np.random.default_rng(5)
    taps = lpf(255, 0.05)
    n = 8 * 6000
    x = rand_c(rng, n)
    whole = c.ChainNode(0.25, 0.4, taps, 8, False, mixer_after_fir=True, kernel="poly").run(x)
    cut = 8 * 2500
    left = c.ChainNode(0.25, 0.4, taps, 8, False, mixer_after_fir=True, kernel="poly")
    yl = left.run(x[:cut])
    right = c.ChainNode(0.25, 0.0, taps, 8, False, mixer_after_fir=True, kernel="poly")
    right.set_fir_state(x[:cut][::-1][:255])
    right.phase = left.phase
    yr = right.run(x[cut:])
    chain_close(np.concatenate([yl, yr]), whole, taps, x)


@pytest.mark.parametrize("fm,n_taps", [(False, 255), (True, 127), (False, 100), (False, 400), (True, 500)])
@pytest.mark.parametrize("fmt", ["i16", "u8"])
def test_poly8_reads_raw_iq(c, fmt, fm, n_taps):
    """Raw i16 / u8 IQ (src/io/raw_iq.rs:16,50-51; the RTL-SDR bytes of examples/fm_radio.rs:82-90) into the polyphase kernel:
    converted in its load stage with iqformat.hip's arithmetic, so bit for bit what the same chain makes of the converted
    samples, in ragged calls (guarded first / last segments included); and the oracle's convert -> nodes in series."""
    idx = np.arange(8 * 9000, dtype=np.float64)
    z = np.exp(1j * (-2 * np.pi * 0.05 * idx + 8.0 * np.cos(2 * np.pi * idx / 4096)))
    if fmt == "u8":
        raw = np.stack([np.clip(np.rint(z.real * 100 + 127.5), 0, 255), np.clip(np.rint(z.imag * 100 + 127.5), 0, 255)], 1).astype(np.uint8)
        raw[:256, 0], raw[:256, 1] = np.arange(256), np.arange(256)[::-1]
        x = oracle.iq_u8_to_c32(raw)
    else:
        raw = np.stack([np.rint(z.real * 8192), np.rint(z.imag * 8192)], 1).astype(np.int16)
        raw[:4] = [[-32768, 32767], [0, -1], [1, 0], [12345, -12345]]
        x = oracle.iq_i16_to_c32(raw, 1.0 / 8192)
    n = x.size
    taps = lpf(n_taps, 1 / 16.0)
    a = c.ChainNode(0.05, 0.3, taps, 8, fm, kernel="poly").set_input_format(fmt, 1.0 / 8192)
    b = c.ChainNode(0.05, 0.3, taps, 8, fm, kernel="poly")
    assert a.kernel == "poly" and b.kernel == "poly"
    cuts = [0, 8, 8 * 700, 8 * 701, n]
    got = np.concatenate([a.run(raw[p:q]) for p, q in zip(cuts[:-1], cuts[1:])])
    ref = np.concatenate([b.run(x[p:q]) for p, q in zip(cuts[:-1], cuts[1:])])
    assert np.array_equal(got, ref)
    y = oracle.decimate(oracle.batch_fir(oracle.Mixer(0.3, 0.05).mix(x), taps, oracle.default_state(taps), norotate=True), 8)
    if fm:
        want = oracle.FM().demod(y)
        mag = np.minimum(np.abs(y), np.abs(np.concatenate([[0.0], y[:-1]])))
        assert np.max((circ(got.astype(np.float64) - want) * mag)[n_taps // 8 + 2:]) <= 4 * TOL * np.sum(np.abs(taps)) * np.max(np.abs(x))
    else:
        chain_close(got, y, taps, x)


@pytest.mark.parametrize("rate", [4, 16, 24, 32, 40, 48, 56, 64, 12, 20, 28, 36, 44, 52, 60])
@pytest.mark.parametrize("n_taps,cplx,after", [(255, False, True), (257, True, False), (131, False, False), (100, True, True), (33, False, True),
                                               (300, True, False), (513, False, True)])
def test_poly8_other_rates_against_oracle(c, rate, n_taps, cplx, after):
    """Rate 4: the outputs y[8j + 4] too, from the same forward transforms through a second set of branch spectra (taps h[8m - c + 4])
    and a second inverse half, interleaved with y[8j].  Rates 8 m: every m-th output of the rate-8 form; rates 4 m (m odd): every
    m-th output of the rate-4 form.  Ragged calls (a single
    output; lengths that are multiples of the rate but not of 8 at rate 4), state carried, both mixer orders."""
    rng = np.random.default_rng(rate * 1000 + n_taps)
    taps = oracle.rrc_taps(n_taps, 8.0, 0.35)
    if cplx:
        taps = (taps * np.exp(1j * 0.01 * np.arange(n_taps))).astype(np.complex64)
    dphase, phase = 2 * np.pi * 0.1, 0.3
    node = c.ChainNode(dphase, phase, taps, rate, False, mixer_after_fir=after, kernel="poly")
    assert node.kernel == "poly"
    n = rate * (34000 // rate + 11)
    x = rand_c(rng, n)
    ost, om = oracle.default_state(taps), oracle.Mixer(phase, dphase)
    cuts = [0, rate, rate * 9, rate * (2500 // rate), rate * (9000 // rate + 1), n]
    for a, b in zip(cuts[:-1], cuts[1:]):
        if after:
            w = oracle.decimate(om.mix(oracle.batch_fir(x[a:b], taps, ost, norotate=True)), rate)
        else:
            w = oracle.decimate(oracle.batch_fir(om.mix(x[a:b]), taps, ost, norotate=True), rate)
        chain_close(node.run(x[a:b]), w, taps, x)
    np.testing.assert_array_equal(node.fir_state(n_taps), x[::-1][:n_taps])


def test_poly8_other_rates_where_the_chain_uses_it(c):
    """An unforced chain reaches the kernel at rate 4 from 128 taps (64 on long batches), at rates 16 ... 56 from 32 taps, at rate 64
    from 128 taps (the any-rate kernel keeps short filters on short batches: it reads only its windows), and says so."""
    rng = np.random.default_rng(4)
    x = rand_c(rng, 64 * 600)
    for rate, n_taps, want in [(4, 255, "poly"), (4, 100, "time"), (16, 63, "poly"), (16, 17, "time"), (24, 200, "poly"), (40, 40, "poly"),
                               (64, 255, "poly"), (64, 100, "time_any"), (72, 255, "time_any"), (12, 255, "poly"), (12, 127, "time"),
                               (20, 255, "poly"), (28, 255, "time_any"), (10, 255, "time")]:
        node = c.ChainNode(0.3, 0.1, lpf(n_taps, 1 / (2.5 * rate)), rate, False, mixer_after_fir=True)
        node.run(x[: x.size - x.size % rate])
        assert node.kernel == want, (rate, n_taps, node.kernel)
    # at rate 4 with the FM demodulator and a short filter the chain stays where it was
    assert c.ChainNode(0.3, 0.1, lpf(63, 0.1), 4, True).kernel == "time"


@pytest.mark.parametrize("rate", [4, 32])
def test_poly8_other_rates_raw_iq_and_long_batches(c, rate):
    """Raw i16 input at the other rates (bit for bit the converted stream's outputs), and a batch of thousands of ticketed segments
    against the overlap-save fusion on every output."""
    import torch

    idx = np.arange(rate * 4000, dtype=np.float64)
    z = np.exp(1j * (-2 * np.pi * 0.05 * idx + 8.0 * np.cos(2 * np.pi * idx / 4096)))
    raw = np.stack([np.rint(z.real * 8192), np.rint(z.imag * 8192)], 1).astype(np.int16)
    x = oracle.iq_i16_to_c32(raw, 1.0 / 8192)
    taps = lpf(201, 1 / (2.5 * rate))
    a = c.ChainNode(0.05, 0.3, taps, rate, False, kernel="poly").set_input_format("i16", 1.0 / 8192)
    b = c.ChainNode(0.05, 0.3, taps, rate, False, kernel="poly")
    cuts = [0, rate, rate * 1700, x.size]
    got = np.concatenate([a.run(raw[p:q]) for p, q in zip(cuts[:-1], cuts[1:])])
    ref = np.concatenate([b.run(x[p:q]) for p, q in zip(cuts[:-1], cuts[1:])])
    assert np.array_equal(got, ref)
    n = 768 * 9000 + rate * 5
    n -= n % rate
    xd = torch.empty(n, dtype=torch.complex64, device="cuda:0")
    c.synth_iq_dev(xd.data_ptr(), n, 0, 77)
    s = torch.cuda.current_stream().cuda_stream
    ys = []
    for kern in ("poly", "freq"):
        node = c.ChainNode(0.7, 0.2, taps, rate, False, mixer_after_fir=True, kernel=kern)
        y = torch.empty(n // rate, dtype=torch.complex64, device="cuda:0")
        node.run_dev(xd.data_ptr(), n, y.data_ptr(), s)
        torch.cuda.synchronize()
        ys.append(y)
    bound = 2 * TOL * float(np.sum(np.abs(taps))) * xd[: 1 << 20].abs().max().item()
    assert (ys[0] - ys[1]).abs().max().item() <= 2 * bound


@pytest.mark.parametrize("rate,n_taps", [(4, 255), (4, 129), (16, 200), (24, 65), (32, 255), (64, 249), (12, 255), (16, 400), (64, 513),
                                         (8, 510), (4, 511), (20, 300)])
def test_poly8_other_rates_with_a_separate_demodulator(c, rate, n_taps):
    """FM chains at the rates the polyphase kernel runs without its own demodulator: mixer / FIR / decimate on it, FMDemodNode's
    kernel behind it over the kept samples -- the oracle's four nodes in series, ragged calls, FM.prev carried."""
    rng = np.random.default_rng(rate + n_taps)
    taps = lpf(n_taps, 1 / (2.5 * rate))
    n = rate * (40000 // rate + 7)
    t = np.arange(n)
    x = (np.exp(1j * (0.02 * t + 3.0 * np.sin(2 * np.pi * t / 5000.0))) * (1 + 0.1 * rng.standard_normal(n))).astype(np.complex64)
    node = c.ChainNode(0.3, 0.1, taps, rate, True)
    ost, om, ofm = oracle.default_state(taps), oracle.Mixer(0.1, 0.3), oracle.FM()
    cuts = [0, rate, rate * 3, rate * (9000 // rate), n]
    last = 0j
    for a, b in zip(cuts[:-1], cuts[1:]):
        y = oracle.decimate(oracle.batch_fir(om.mix(x[a:b]), taps, ost, norotate=True), rate)
        w = ofm.demod(y)
        got = node.run(x[a:b])
        assert node.kernel == "poly"
        mag = np.minimum(np.abs(y), np.abs(np.concatenate([[last], y[:-1]])))
        last = y[-1]
        assert np.max(circ(got.astype(np.float64) - w) * mag) <= 4 * TOL * np.sum(np.abs(taps)) * np.max(np.abs(x))
    assert abs(complex(node.fm_prev) - complex(y[-1])) <= 2 * TOL * np.sum(np.abs(taps)) * np.max(np.abs(x))


def test_non_finite_sample_reach(c):
    """A NaN input sample: the reference's direct sum carries it into the N outputs that follow.  The GPU kernels carry it at least
    there and not much further: the time-domain chain kernel through its zero-padded tap window (0 x NaN = NaN: a few outputs either
    side), the frequency-domain one through the segment the sample falls in and, as halo, the next.  Everything else matches."""
    rng = np.random.default_rng(21)
    taps = lpf(101, 0.05)
    n = 8 * 2000
    x = rand_c(rng, n)
    p = 8 * 700 + 3
    x[p] = np.nan + 0j
    want = oracle.decimate(oracle.Mixer(0.0, 0.2).mix(oracle.batch_fir(x, taps, oracle.default_state(taps), norotate=True)), 8)
    bad = np.isnan(want.real) | np.isnan(want.imag)
    j = np.arange(n // 8)
    assert np.array_equal(bad, (8 * j >= p) & (8 * j <= p + 100))  # the outputs whose sum holds sample p
    t = c.ChainNode(0.2, 0.0, taps, 8, False, mixer_after_fir=True, kernel="time").run(x)
    tb = np.isnan(t.real) | np.isnan(t.imag)
    assert tb[bad].all() and not tb[8 * j < p - 64].any() and not tb[8 * j > p + 100 + 64].any()
    chain_close(t[~tb], want[~tb], taps, np.nan_to_num(x))
    f = c.ChainNode(0.2, 0.0, taps, 8, False, mixer_after_fir=True, kernel="poly").run(x)
    fb = np.isnan(f.real) | np.isnan(f.imag)
    assert fb[bad].all()
    seg = 896 // 8  # (101 taps: two halo rows, 896 new samples = 112 outputs per segment)
    lo, hi = (p // 896) * seg, (p // 896 + 2) * seg
    assert not fb[:lo].any() and not fb[hi:].any()
    chain_close(f[~fb], want[~fb], taps, np.nan_to_num(x))
