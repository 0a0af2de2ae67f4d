"""The Complex<i16> instantiations of FirNode / BatchFirNode / PulseNode (the reference's fir(), batch_fir() and
PulseNode are generic over T: src/filter/fir.rs:43-54, :87-102, src/pulse.rs:38-93) -- on the type the
reference's own golden tests use (src/filter/fir_node.rs:259-313, src/pulse.rs:129-183), bit-exact, including
the wrap-around of i16 arithmetic.  Run with -m gpu."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c():
    import comms_rs_amd as c

    assert c.device_count() >= 1, "no MI355X visible: the HIP path cannot be tested (no CPU fallback)"
    return c


def i16(a):
    return np.asarray(a, dtype=np.int16).reshape(-1, 2)


def test_fir_i16_reference_golden_sample_by_sample_and_in_batches(c, kats):
    k = kats["fir_i16"]
    node = c.FirNodeI16(i16(k["taps"]))
    got = np.array([node.run(np.array(v, np.int16)) for v in k["input"]])
    assert np.array_equal(got[:9], i16(k["expected"]))
    # the batch node, two samples per message as the reference's batch test feeds it (fir_node.rs:342-449)
    node = c.BatchFirNodeI16(i16(k["taps"]))
    x = i16(k["input"])
    got = np.concatenate([node.run(x[i:i + 2]) for i in range(0, x.shape[0], 2)])
    assert np.array_equal(got[:9], i16(k["expected"]))


def test_pulse_i16_reference_golden(c, kats):
    k = kats["pulse_rect_i16"]
    taps = np.stack([np.ones(k["n_taps"]), np.zeros(k["n_taps"])], 1).astype(np.int16)   # rect_taps::<i16>(4)
    node = c.PulseNodeI16(taps, k["sam_per_sym"])
    got = np.concatenate([node.run(np.array([s], np.int16)) for s in k["symbols"]])
    assert np.array_equal(got, i16(k["expected"]))


@pytest.mark.parametrize("n_taps,n", [(1, 5), (5, 1), (63, 10000), (255, 70001), (5000, 9000)])
def test_batch_fir_i16_vs_oracle_with_wraparound(c, n_taps, n):
    """Full-range random taps and samples: nearly every product overflows i16; the device accumulates modulo 2^32
    and truncates once, the oracle wraps at every step as the reference's release build does -- the same ring."""
    rng = np.random.default_rng(n_taps)
    taps = rng.integers(-32768, 32768, (n_taps, 2), dtype=np.int16)
    x = rng.integers(-32768, 32768, (n, 2), dtype=np.int16)
    node = c.BatchFirNodeI16(taps)
    st = np.zeros_like(taps)
    cuts = sorted({0, n // 3, n // 3 + 1, n})
    for a, b in zip(cuts[:-1], cuts[1:]):   # state carries across calls
        assert np.array_equal(node.run(x[a:b]), oracle.batch_fir(x[a:b], taps, st)), (n_taps, a, b)
    assert np.array_equal(node.state(min(n_taps, 7)), st[:min(n_taps, 7)])


def test_fir_i16_user_state_shorter_and_longer_than_the_taps(c):
    rng = np.random.default_rng(2)
    taps = rng.integers(-100, 100, (9, 2), dtype=np.int16)
    x = rng.integers(-100, 100, (50, 2), dtype=np.int16)
    for n_state in (4, 9, 15):   # zip(taps, state): only min(len) taps take part (fir.rs:53)
        st = rng.integers(-100, 100, (n_state, 2), dtype=np.int16)
        want = oracle.batch_fir(x, taps, st.copy())
        assert np.array_equal(c.BatchFirNodeI16(taps, st).run(x), want), n_state
    with pytest.raises(c.CommsError):
        c.BatchFirNodeI16(np.zeros((0, 2), np.int16))


@pytest.mark.parametrize("n_taps,sps", [(4, 4), (63, 4), (17, 3), (8, 16), (100, 7), (5, 1)])
def test_pulse_i16_vs_oracle(c, n_taps, sps):
    rng = np.random.default_rng(sps)
    taps = rng.integers(-3000, 3000, (n_taps, 2), dtype=np.int16)
    sym = rng.integers(-3000, 3000, (4001, 2), dtype=np.int16)
    node, got, want = c.PulseNodeI16(taps, sps), [], []
    st = np.zeros_like(taps)
    for a, b in ((0, 1), (1, 1500), (1500, 4001)):
        got.append(node.run(sym[a:b]))
        want.append(oracle.pulse(sym[a:b], taps, sps, st))
    assert np.array_equal(np.concatenate(got), np.concatenate(want))
    with pytest.raises(c.CommsError):
        c.PulseNodeI16(taps, 0)
