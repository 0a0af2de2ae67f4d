"""The host-pointer (`Vec<Complex<T>>` in, `Vec` out) path of long batches is cut into chunks that are copied in, run and
copied out in a pipeline (Handle::run_host_units, csrc/common.hpp): the reference's own node signature,
run(&[Complex<T>]) -> Vec<Complex<T>> (src/filter/fir_node.rs:215-220), is what an unchanged graph calls.

Chunking must not be visible in the samples.  A chunk is a batch like any other: the node streams across chunks as it
streams across calls.  So (a) for every node the pipelined call equals, BIT FOR BIT, the same node fed the same chunks
one run() at a time with the pipeline switched off (same launches in the same order), with ragged totals and state
carried in; (b) where the arithmetic of an output does not depend on the batch length -- mixer, decimate, upsample, FM
demod, the direct-form FIR, FFT batches, integer FIR -- it also equals the single-shot call bit for bit; (c) the
frequency-domain FIR and the fused chains pick their kernel by batch length, so against the single shot they are held to
the parity tolerance instead.  (A decimating chain of rate 4 and more is not pipelined at all: its output is a small
fraction of its input, and one big copy in is faster than chunks.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHUNK = 16 << 20  # Handle::kHostChunkBytes: bytes of the larger side per chunk


def chunks(n_units, in_u, out_u):
    per = max(1, CHUNK // max(in_u, out_u))
    return [(a, min(a + per, n_units)) for a in range(0, n_units, per)]


_SINGLE = r'''
import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
from host_pipeline_cases import CASES
x, make, units = CASES[sys.argv[1]]()
np.save(sys.argv[2], make().run(x))
'''


def single_shot(case, tmp_path):
    """The same call in a process whose pipeline is off (COMMS_HOST_PIPE_BYTES=0 is read once per process)."""
    out = str(tmp_path / (case + ".npy"))
    env = dict(os.environ, COMMS_HOST_PIPE_BYTES="0")
    code = _SINGLE % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code, case, out], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return np.load(out)


from host_pipeline_cases import CASES, _lpf  # noqa: E402  (tests/ is on sys.path under pytest's rootdir conftest)

EXACT_VS_SINGLE = {"fir_direct", "mixer", "fmdemod", "decimate", "upsample", "fft"}


@pytest.mark.parametrize("case", sorted(CASES))
def test_pipelined_host_call_equals_the_node_fed_chunk_by_chunk(case, tmp_path):
    x, make, (in_u, out_u) = CASES[case]()
    in_elem = 8
    n_units = -(-x.size * in_elem // in_u)
    cuts = chunks(n_units, in_u, out_u)
    assert len(cuts) >= 3, "the case must span several chunks"
    got = make().run(x)                       # one call: pipelined (in + out >= 64 MiB)
    node = make()                             # the same chunks, one short call each -- every call below the pipeline's limit
    parts = []
    for a, b in cuts:
        s0, s1 = a * in_u // in_elem, min(b * in_u // in_elem, x.size)
        assert (s1 - s0) * (in_elem + out_u * in_elem // in_u) < (64 << 20)
        parts.append(node.run(x[s0:s1]))
    want = np.concatenate(parts)
    assert got.shape == want.shape and got.dtype == want.dtype
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), "chunked call differs from chunk-by-chunk calls"
    one = single_shot(case, tmp_path)
    assert one.shape == got.shape
    if case in EXACT_VS_SINGLE:
        assert np.array_equal(got.view(np.uint8), one.view(np.uint8)), "chunked call differs from the single-shot call"
    else:   # fir_auto (overlap-save: an output's rounding depends on where its segment starts), chain_r2 (tile-relative rotors)
        import comms_rs_amd as c
        taps = c.rrc_taps(255, 8.0, 0.35) if case == "fir_auto" else _lpf(63, 1 / 5.0)
        bound = 2e-5 * float(np.sum(np.abs(taps))) * float(np.max(np.abs(x)))
        assert np.max(np.abs(got.astype(np.complex128) - one.astype(np.complex128))) <= bound


def test_pipelined_host_call_carries_state_in_and_out():
    """State handed in before a pipelined call and the state it leaves: as after the same samples in short calls."""
    import comms_rs_amd as c

    taps = _lpf(63, 0.05)
    x = c.synth_iq((5 << 20) + 101, 0, 41)
    st0 = c.synth_iq(63, 0, 42)
    a = c.BatchFirNode(taps, st0).set_algo(c.FIR_DIRECT)
    b = c.BatchFirNode(taps, st0).set_algo(c.FIR_DIRECT)
    ya = a.run(x)
    yb = np.concatenate([b.run(x[i:i + (1 << 19)]) for i in range(0, x.size, 1 << 19)])
    assert np.array_equal(ya.view(np.uint8), yb.view(np.uint8))
    assert np.array_equal(a.state(63).view(np.uint8), b.state(63).view(np.uint8))
    tail = x[:4096]
    assert np.array_equal(a.run(tail).view(np.uint8), b.run(tail).view(np.uint8))


def test_pipelined_host_calls_from_several_node_threads_at_once():
    """comms-rs runs one OS thread per node (src/node/mod.rs:276-284): four threads, each with its own FIR node, make
    pipelined host calls at the same time (every call brings its own helper thread and second stream); each gets exactly
    the samples a lone call gets."""
    from concurrent.futures import ThreadPoolExecutor

    import comms_rs_amd as c

    taps = [_lpf(31 + 8 * i, 0.1) for i in range(4)]
    xs = [c.synth_iq((5 << 20) + 1000 * i, 0, 50 + i) for i in range(4)]
    want = [c.BatchFirNode(t).set_algo(c.FIR_DIRECT).run(x) for t, x in zip(taps, xs)]
    nodes = [c.BatchFirNode(t).set_algo(c.FIR_DIRECT) for t in taps]
    with ThreadPoolExecutor(4) as ex:
        got = list(ex.map(lambda i: nodes[i].run(xs[i]), range(4)))
    for g, w in zip(got, want):
        assert np.array_equal(g.view(np.uint8), w.view(np.uint8))
