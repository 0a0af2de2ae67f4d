"""Block estimators: oracle vs the reference's own test
properties (CPU), and the device reductions vs the oracle (GPU).

The reference tests (frequency_estimator.rs:57-95, phase_estimator.rs:77-126) draw symbols
from SmallRng::seed_from_u64(0), which cannot be reproduced without the rand crate; they
assert |truth - estimate| below 0.01 / 1e-6 / 0.01.  The same constructions with numpy's
generator must meet the same bounds."""
import numpy as np
import pytest

import oracle


def psk_stream(rng, m, n, truth):
    return np.exp(1j * (2 * np.pi * rng.integers(0, m, n) / m + truth))


def qam16_stream(rng, n, truth):
    x = rng.integers(0, 16, n)
    return 2.0 * ((x % 4) - 1.5 + 1j * (np.trunc(x / 4.0) - 1.5)) * np.exp(1j * truth)


def freq_stream(rng, truth):
    # 4-PSK, x4 zero-stuffed, 16-tap RRC (beta 0.75), then a frequency shift of `truth` rad/sample
    sym = np.exp(2j * np.pi * rng.integers(0, 4, 4096) / 4)
    up = np.zeros(4096 * 4, np.complex128)
    up[::4] = sym
    taps = oracle.rrc_taps(16, 4.0, 0.75, dtype=np.complex128)
    data = oracle.batch_fir(up, taps, oracle.default_state(taps))
    return data * np.exp(1j * truth * np.arange(data.size))


def test_oracle_meets_the_reference_test_bounds():
    rng = np.random.default_rng(0)
    assert abs(0.123456789 - oracle.frequency_offset_estimate(freq_stream(rng, 0.123456789))) < 0.01
    assert abs(0.123456 - oracle.psk_phase_estimate(psk_stream(rng, 8, 1000, 0.123456), 8)) < 1e-6
    assert abs(0.123456 - oracle.qam_phase_estimate(qam16_stream(rng, 1000, 0.123456))) < 0.01
    # independent definitions
    x = rng.standard_normal(500) + 1j * rng.standard_normal(500)
    assert abs(oracle.frequency_offset_estimate(x) - np.angle(np.sum(x[1:] * np.conj(x[:-1])))) < 1e-12
    for m in (1, 2, 3, 4, 8):
        assert abs(oracle.psk_phase_estimate(x, m) - np.angle(np.sum(x ** m)) / m) < 1e-12
    assert abs(oracle.qam_phase_estimate(x) - np.angle(np.sum(-(x ** 4))) / 4) < 1e-12


@pytest.mark.gpu
def test_device_estimators_vs_oracle():
    import comms_rs_amd as c

    rng = np.random.default_rng(1)
    cases = [freq_stream(rng, 0.123456789), psk_stream(rng, 8, 1000, 0.123456), qam16_stream(rng, 1000, 0.123456),
             rng.standard_normal(1 << 20) + 1j * rng.standard_normal(1 << 20), np.array([1 + 1j]), np.array([2 - 1j, 0.5j])]
    for x in cases:
        # parallel tree vs sequential fold: rounding-level differences only
        assert abs(c.frequency_offset_estimate(x) - oracle.frequency_offset_estimate(x)) < 1e-9
        for m in (1, 2, 4, 8):
            assert abs(c.psk_phase_estimate(x, m) - oracle.psk_phase_estimate(x, m)) < 1e-9
        assert abs(c.qam_phase_estimate(x) - oracle.qam_phase_estimate(x)) < 1e-9
    assert abs(0.123456 - c.psk_phase_estimate(cases[1], 8)) < 1e-6  # the reference's own bound
    with pytest.raises(c.CommsError):
        c.psk_phase_estimate(cases[1], 0)


# ------------------------------------------------------------------ timing estimator + NCO
def timing_stream(rng, alpha=0.5, sps=10, nsym=1000):
    # timing_estimator.rs:151-176: QPSK, x10 zero-stuffed, 101-tap RRC
    sym = np.exp(1j * (2 * np.pi * rng.integers(0, 4, nsym) / 4 + np.pi / 4))
    up = np.zeros(nsym * sps, np.complex128)
    up[::sps] = sym
    taps = oracle.rrc_taps(sps * 10 + 1, float(sps), alpha, dtype=np.complex128)
    return oracle.batch_fir(up, taps, oracle.default_state(taps))


def test_qfilt_golden_oracle_and_host():
    import json
    import pathlib

    g = json.loads((pathlib.Path(__file__).parent / "golden" / "reference_kats.json").read_text())["qfilt_taps"]
    want = np.array(g["expected"])
    got = oracle.qfilt_taps(g["n_taps"], g["alpha"], g["sam_per_sym"])
    assert got.size == 21 and np.all(np.abs(got[:20] - want) < g["tol_abs"])
    assert oracle.qfilt_taps(20, 0.25, 2).size == 21  # even counts are incremented (math.rs:317-320)
    with pytest.raises(ValueError):
        oracle.qfilt_taps(21, 1.5, 2)
    # l'Hospital branch: 2*alpha*tt == +-1 exactly (alpha 0.5, tt = +-1)
    t = oracle.qfilt_taps(9, 0.5, 2)
    assert np.isfinite(t).all() and abs(t[2] - np.sin(np.pi * 0.5 * -1.0) / -8.0) < 1e-16
    import comms_rs_amd as c  # host code of the product library: no GPU needed

    assert np.array_equal(c.qfilt_taps(21, 0.25, 2), got)
    assert np.array_equal(c.qfilt_taps(9, 0.5, 2), t)
    with pytest.raises(c.CommsError):
        c.qfilt_taps(21, -0.1, 2)


def test_oracle_timing_meets_the_reference_test_bound():
    # timing_estimator.rs:178-192: estimate of samples[truth..] within 0.01 samples of -truth
    rng = np.random.default_rng(0)
    x = timing_stream(rng)
    for truth in (2, 0, 4):
        assert abs(truth + oracle.timing_push(x[truth:], 10, 5, 0.5)) < 0.01


def test_oracle_nco_against_closed_form():
    # no test in the reference (nco.rs has none): parity unpinned; check the restatement
    # against the closed form exp(i*(phase0 + cumsum(dphase + perr)))
    rng = np.random.default_rng(2)
    perr = 0.01 * rng.standard_normal(5000)
    nco = oracle.Nco(0.1, np.pi / 4)
    got = np.concatenate([nco.push(perr[:1234]), nco.push(perr[1234:])])
    want = np.exp(1j * (np.pi / 4 + np.cumsum(0.1 + perr)))
    assert np.max(np.abs(got - want)) < 1e-11
    assert np.max(np.abs(oracle.Nco(0.1 + 4 * np.pi).push(perr) - oracle.Nco(0.1).push(perr))) < 1e-9


@pytest.mark.gpu
def test_device_timing_estimator_vs_oracle():
    import comms_rs_amd as c

    rng = np.random.default_rng(3)
    x = timing_stream(rng)
    for n, d, alpha, off in ((10, 5, 0.5, 2), (10, 5, 0.5, 0), (10, 30, 0.25, 3), (2, 5, 0.25, 0), (4, 1, 1.0, 1),
                             (10, 60, 0.3, 1)):  # 1201 taps: two passes of the one-pass kernel
        est = c.TimingEstimatorNode(n, d, alpha)
        got = est.run(x[off:])
        want = oracle.timing_push(x[off:], n, d, alpha)
        assert abs(got - want) < 1e-9, (n, d, alpha, off, got, want)
        assert abs(est.run(x[off:]) - got) == 0.0  # fresh filter state on every push, reproducible
    assert abs(2 + c.TimingEstimatorNode(10, 5, 0.5).run(x[2:])) < 0.01  # the reference's own bound
    # ragged / tiny blocks, shorter than the filter
    for ln in (1, 7, 100, 257, 513):
        y = rng.standard_normal(ln) + 1j * rng.standard_normal(ln)
        assert abs(c.TimingEstimatorNode(10, 5, 0.5).run(y) - oracle.timing_push(y, 10, 5, 0.5)) < 1e-9
    # a long block: the rotor angle -pi*i/n reaches 6.6e5 rad, sample indices cross many tiles
    long = np.tile(x, 200)[: (1 << 21) + 333]
    assert abs(c.TimingEstimatorNode(10, 5, 0.5).run(long) - oracle.timing_push(long, 10, 5, 0.5)) < 1e-9
    with pytest.raises(c.CommsError):
        c.TimingEstimatorNode(10, 5, 1.5)
    with pytest.raises(c.CommsError):
        c.TimingEstimatorNode(0, 5, 0.5)


@pytest.mark.gpu
def test_device_nco_vs_oracle():
    import comms_rs_amd as c

    rng = np.random.default_rng(4)
    for dphase, phase, n in ((0.1, np.pi / 4, 5000), (0.1, None, 1), (6.0, 1.0, 2049), (-0.3, 0.0, 1 << 20),
                             (2 * np.pi * 0.123 + 4 * np.pi, 5.0, 300000)):
        perr = 0.01 * rng.standard_normal(n)
        ref, dev = oracle.Nco(dphase, phase), c.NcoNode(dphase, phase)
        cuts = sorted({0, n // 3, n // 3 + 1, n})
        for a, b in zip(cuts[:-1], cuts[1:]):  # the phase persists across calls
            got, want = dev.run(perr[a:b]), ref.push(perr[a:b])
            assert np.max(np.abs(got - want)) < 1e-9, (dphase, n, a, b)
        assert abs(np.exp(1j * dev.phase) - np.exp(1j * ref.phase.value)) < 1e-9
    assert c.NcoNode(0.1).run(np.zeros(0)).size == 0
    with pytest.raises(c.CommsError):
        c.NcoNode(float("nan"))


@pytest.mark.gpu
def test_device_nco_long_block():
    """2^24 + 777 phase errors, device-resident, twice in a row on one node: 8193 tiles (sum / scan / apply over a
    persistent grid, the rotor from the 1024-entry table + Taylor remainder).  With a constant error the phase has the closed
    form phase0 + (i + 1) * (dphase + e); checked in extended precision at the tile seams and at random places."""
    import torch

    import comms_rs_amd as c

    n = (1 << 24) + 777
    dphase, phase0, e = 0.1234567, 0.5, 1e-3
    perr = torch.full((n,), e, dtype=torch.float64, device="cuda:0")
    out = torch.empty(n, dtype=torch.complex128, device="cuda:0")
    node = c.NcoNode(dphase, phase0)
    s = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(1)
    idx = np.unique(np.concatenate([[0, 1, 2047, 2048, 2049, 4095, 4096, n - 2, n - 1], rng.integers(0, n, 4000),
                                    2048 * rng.integers(1, n // 2048, 500), 2048 * rng.integers(1, n // 2048, 500) - 1]))
    two_pi = 2 * np.longdouble(np.pi)
    inc = np.longdouble(dphase) + np.longdouble(e)
    for call in range(2):
        node.run_dev(perr.data_ptr(), n, out.data_ptr(), s)
        torch.cuda.synchronize()
        got = out[torch.from_numpy(idx).cuda()].cpu().numpy()
        ph = np.fmod(np.longdouble(phase0) + (np.longdouble(call) * n + idx.astype(np.longdouble) + 1) * inc, two_pi)
        want = np.cos(ph).astype(np.float64) + 1j * np.sin(ph).astype(np.float64)
        assert np.max(np.abs(got - want)) < 1e-9, call
    end = np.fmod(np.longdouble(phase0) + 2 * np.longdouble(n) * inc, two_pi)
    assert abs(np.exp(1j * node.phase) - np.exp(1j * float(end))) < 1e-9


@pytest.mark.gpu
def test_block_rate_carrier_loop_device_vs_oracle():
    """the NCO inside a feedback loop, run at block rate.  Per block: the NCO (device)
    produces exp(i phase) from the loop's per-sample phase error, the block is de-rotated by it, the frequency
    estimator (device) measures the residual offset, a first-order loop filter feeds it back as the next block's
    phase error (Nco::push's perr input, nco.rs:71-77; the feedback edge is what connect_nodes_feedback! wires,
    node/mod.rs:213-219).  The same loop runs on the oracle's Nco + frequency_offset_estimate: the two
    trajectories must agree block by block, and the loop must pull in the carrier offset."""
    import comms_rs_amd as c

    rng = np.random.default_rng(11)
    block, n_blocks = 4096, 40
    w_true, w_nco, mu = 0.1234, 0.1, 0.5   # rad / sample: the carrier, the NCO's nominal rate, the loop gain
    n = block * n_blocks
    x = np.exp(1j * (w_true * np.arange(n) + 0.7)) + 0.05 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    dev, ref = c.NcoNode(w_nco, 0.3), oracle.Nco(w_nco, 0.3)
    e_dev = e_ref = 0.0
    res_dev, res_ref = [], []
    for b in range(n_blocks):
        xb = x[b * block:(b + 1) * block]
        y_dev = xb * np.conj(dev.run(np.full(block, e_dev)))
        y_ref = xb * np.conj(ref.push(np.full(block, e_ref)))
        r_dev = c.frequency_offset_estimate(y_dev)
        r_ref = oracle.frequency_offset_estimate(y_ref)
        res_dev.append(r_dev)
        res_ref.append(r_ref)
        e_dev += mu * r_dev
        e_ref += mu * r_ref
    res_dev, res_ref = np.array(res_dev), np.array(res_ref)
    assert np.max(np.abs(res_dev - res_ref)) < 1e-9          # the closed loops follow the same trajectory
    assert abs(e_dev - e_ref) < 1e-9
    assert abs(res_dev[0] - (w_true - w_nco)) < 1e-3           # open-loop residual = the offset
    assert np.max(np.abs(res_dev[-10:])) < 1e-3                # pulled in
    assert abs(w_nco + e_dev - w_true) < 1e-3                  # the loop's frequency = the carrier's
