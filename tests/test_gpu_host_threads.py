"""Node threads come and go (comms-rs starts one thread per node, src/node/mod.rs:276-284; a graph may be torn down and
built again).  The handle-less host-pointer entries (decimate, upsample, IQ formats, estimators) borrow a per-thread
handle -- a pooled stream, device scratch, pinned staging -- which must END with its thread: 64 threads that each call
comms_decimate_run once, one after the other, have to share the ONE stream the first of them created."""
import ctypes as C
import os
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_per_thread_handles_end_with_their_thread():
    import comms_rs_amd as c

    path = os.path.join(os.path.dirname(c.LIB_PATH), "libcomms_hip_diag.so")
    assert os.path.exists(path), "the diagnostic build is part of build(): %s" % path
    d = C.CDLL(path)
    vp, sz = C.c_void_p, C.c_size_t
    d.comms_decimate_run.argtypes = [vp, sz, sz, sz, vp, C.POINTER(sz), C.c_int32]
    d.comms_debug_streams_created.argtypes = [C.c_int32]
    d.comms_debug_streams_created.restype = C.c_long
    x = (np.arange(4096) + 1j * np.arange(4096)).astype(np.complex64)
    errs = []

    def node_thread(i):
        out = np.empty(1024, np.complex64)
        n_out = sz()
        st = d.comms_decimate_run(x.ctypes.data, x.size, 8, 4, out.ctypes.data, C.byref(n_out), 0)
        if st != 0 or n_out.value != 1024 or not np.array_equal(out, x[::4]):
            errs.append((i, st, n_out.value))

    node_thread(-1)  # this (the test's) thread keeps its handle: one stream
    before = d.comms_debug_streams_created(0)
    for i in range(64):
        t = threading.Thread(target=node_thread, args=(i,))
        t.start()
        t.join()
    assert not errs, errs
    after = d.comms_debug_streams_created(0)
    # the first short-lived thread creates a stream, every later one takes it from the pool (<= 2: headroom for one stream)
    assert after - before <= 2, "per-thread handles leak their streams: %d created by 64 threads" % (after - before)
    # eight threads alive at once need at most eight streams, and waves of eight that follow reuse them: never more than
    # the threads that were alive together (without the fix: one per thread ever started, 64 + 48 here)
    def wave():
        ts = [threading.Thread(target=node_thread, args=(100 + i,)) for i in range(8)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        time.sleep(0.2)  # (join() returns when the Python thread body ends; the OS thread runs its thread_local destructors after that)
    for _ in range(6):
        wave()
    assert not errs, errs
    assert d.comms_debug_streams_created(0) - before <= 8 + 2
