"""FirNode<f64> / BatchFirNode<f64> / PulseNode<f64> (comms_fir_f64_*, comms_pulse_f64_*; round 5).  The reference's FIR
is generic over the sample type (src/filter/fir.rs:43-54, :87-102, src/pulse.rs:82-92); its batch_fir doc example
(fir.rs:68-86) and its timing estimator (timing_estimator.rs:102-103) run it on Complex<f64>.  The kernel does the
reference's arithmetic operation for operation -- num::Complex's four-multiplication product, the sum folded from zero
with the taps ascending, no FMA -- so every output is compared for EQUALITY, bit for bit, with the oracle's restatement of
the same lines; and the reference's own integer goldens (fir_node.rs:259-313, pulse.rs:129-183: exact small integers, which
f64 represents exactly) pin both."""
import json
import os

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu
KATS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_kats.json")


@pytest.fixture(scope="module")
def c():
    import comms_rs_amd as c

    assert c.device_count() >= 1, "no MI355X visible: the HIP path cannot be tested (no CPU fallback)"
    return c


def bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


def rand_c128(rng, n, scale=1.0):
    return (scale * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex128)


def test_fir_f64_reference_integer_golden(c):
    """The reference's FIR golden (Complex<i16> vectors, fir_node.rs:259-313) through the f64 node: the same numbers, exactly."""
    k = json.load(open(KATS))["fir_i16"]
    taps = np.array([complex(a, b) for a, b in k["taps"]], np.complex128)
    x = np.array([complex(a, b) for a, b in k["input"]], np.complex128)
    want = np.array([complex(a, b) for a, b in k["expected"]], np.complex128)
    got = c.BatchFirNodeF64(taps).run(x)
    assert np.array_equal(got[:want.size], want)
    node = c.FirNodeF64(taps)                       # FirNode<f64>: one sample per call
    assert np.array_equal(np.array([node.run(v) for v in x])[:want.size], want)


def test_pulse_f64_reference_integer_golden(c):
    """The reference's pulse golden (pulse.rs:129-183: five QPSK symbols through rect_taps(4) at four samples per symbol)."""
    k = json.load(open(KATS))["pulse_rect_i16"]
    taps = np.ones(k["n_taps"], np.complex128)     # rect_taps(4): ones (util/math.rs:48-55)
    sym = np.array([complex(a, b) for a, b in k["symbols"]], np.complex128)
    want = np.array([complex(a, b) for a, b in k["expected"]], np.complex128)
    node = c.PulseNodeF64(taps, k["sam_per_sym"])
    got = np.concatenate([node.run(sym[i:i + 1]) for i in range(sym.size)])   # one symbol per run() call, as PulseNode does
    assert np.array_equal(got, want)
    assert np.array_equal(c.PulseNodeF64(taps, k["sam_per_sym"]).run(sym), want)


def test_batch_fir_f64_doc_example_input(c):
    """The inputs of the reference's batch_fir doc example (fir.rs:68-86; it asserts nothing): cos(0..99), four real taps, a
    user state -- bit for bit the oracle's restatement, and the state it leaves."""
    x = np.cos(np.arange(100, dtype=np.float64)).astype(np.complex128)
    taps = np.array([0.2, 0.6, 0.6, 0.2], np.complex128)
    st0 = np.array([1.0, 0.5, 0.25, 0.125], np.complex128)
    ost = st0.copy()
    want = oracle.batch_fir(x, taps, ost)
    node = c.BatchFirNodeF64(taps, st0)
    got = node.run(x)
    assert np.array_equal(bits(got), bits(want))
    assert got[0] == 0.2 * 1.0 + 0.6 * 1.0 + 0.6 * 0.5 + 0.2 * 0.25
    assert np.array_equal(bits(node.state(4)), bits(ost))


@pytest.mark.parametrize("n_taps,n", [(1, 1), (3, 7), (31, 1000), (127, 4097), (255, 20000), (2500, 3000)])
def test_batch_fir_f64_bit_exact_with_state_over_calls(c, n_taps, n):
    rng = np.random.default_rng(70 + n_taps)
    taps = rand_c128(rng, n_taps, 0.3)
    x = rand_c128(rng, n)
    ost = oracle.default_state(taps)
    node = c.BatchFirNodeF64(taps)
    cuts = sorted({0, n // 3, n // 3 + 1, n})
    for a, b in zip(cuts[:-1], cuts[1:]):
        want = oracle.batch_fir(x[a:b], taps, ost)
        got = node.run(x[a:b])
        assert np.array_equal(bits(got), bits(want)), (a, b)
    assert np.array_equal(bits(node.state(n_taps)), bits(ost))
    # zip(taps, state) truncation (fir.rs:53): a shorter user state cuts the filter
    if n_taps >= 3:
        st = rand_c128(rng, n_taps - 2)
        ost2 = st.copy()
        want = oracle.batch_fir(x[:50], taps, ost2)
        assert np.array_equal(bits(c.BatchFirNodeF64(taps, st).run(x[:50])), bits(want))
    # set_state: the hand-over of a shard / a checkpoint
    node2 = c.BatchFirNodeF64(taps).set_state(node.state(n_taps))
    tail = rand_c128(rng, 64)
    assert np.array_equal(bits(node2.run(tail)), bits(node.run(tail)))


@pytest.mark.parametrize("sps,n_taps", [(1, 5), (4, 63), (3, 40), (8, 255), (20, 7)])
def test_pulse_f64_bit_exact(c, sps, n_taps):
    rng = np.random.default_rng(90 + sps)
    taps = rand_c128(rng, n_taps, 0.2)
    sym = rand_c128(rng, 777)
    ost = oracle.default_state(taps)
    node = c.PulseNodeF64(taps, sps)
    for a, b in ((0, 1), (1, 300), (300, 777)):
        want = oracle.pulse(sym[a:b], taps, sps, ost)
        got = node.run(sym[a:b])
        assert got.shape == want.shape
        # (the kernel skips the products by the stuffed zeros: they add +-0.0 to a sum folded from +0.0, which changes no bit
        # of it -- equality of VALUES and of every non-zero bit pattern; an exactly-zero sum is +0.0 both ways)
        assert np.array_equal(bits(got), bits(want)), (a, b)


def test_fir_f64_device_entry_and_errors(c):
    import ctypes as C

    import torch
    from comms_rs_amd._lib import lib

    rng = np.random.default_rng(5)
    taps = rand_c128(rng, 33, 0.3)
    x = rand_c128(rng, 5000)
    xd = torch.from_numpy(x).to("cuda:0")
    yd = torch.empty_like(xd)
    node = c.BatchFirNodeF64(taps)
    node.run_dev(xd.data_ptr(), x.size, yd.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(bits(yd.cpu().numpy()), bits(oracle.batch_fir(x, taps, oracle.default_state(taps))))
    with pytest.raises(c.CommsError) as e:   # in place is refused, like the other FIR nodes
        node.run_dev(xd.data_ptr(), x.size, xd.data_ptr(), 0)
    assert e.value.code == c.COMMS_ERR_ARG
    h = C.c_void_p()
    assert lib().comms_fir_f64_create(None, 0, None, 0, 0, C.byref(h)) == c.COMMS_ERR_ARG   # the reference panics on an empty state
