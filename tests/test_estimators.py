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
