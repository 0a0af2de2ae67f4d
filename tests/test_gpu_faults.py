"""Device-side failures must surface as COMMS_ERR_DEVICE, not as wrong samples with COMMS_OK (HIP errors map to
NodeError::PermanentError, src/node/mod.rs:68-73).  The 16384-point FIR kernel orders its LDS exchanges with polled
counters instead of barriers; the polls are bounded, and a wave whose wait runs out raises the handle's sticky error
word.  The diagnostic build (libcomms_hip_diag.so, same sources + -DCOMMS_DIAG) can withhold one signal to force that."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_os16k_failed_wait_is_reported_not_silent():
    import torch

    import comms_rs_amd as c

    path = os.path.join(os.path.dirname(c.LIB_PATH), "libcomms_hip_diag.so")
    assert os.path.exists(path), "the diagnostic build is part of build(): %s" % path
    d = C.CDLL(path)
    vp, sz = C.c_void_p, C.c_size_t
    d.comms_fir_create.argtypes = [vp, sz, vp, sz, C.c_int32, C.POINTER(vp)]
    d.comms_fir_set_algo.argtypes = [vp, C.c_int32]
    d.comms_fir_run_dev.argtypes = [vp, vp, sz, vp, vp]
    d.comms_fir_get_state.argtypes = [vp, vp, sz]
    d.comms_fir_destroy.argtypes = [vp]
    d.comms_last_error.restype = C.c_char_p
    k = np.arange(2100) - 1049.5
    taps = np.ascontiguousarray((0.02 * np.sinc(0.02 * k) * np.hamming(2100)).astype(np.float32).astype(np.complex64))
    n = 13312 * 8  # eight 16384-point segments (three halo rows each): one per workgroup
    x = torch.empty(n, dtype=torch.complex64, device="cuda:0")
    y = torch.empty_like(x)
    c.synth_iq_dev(x.data_ptr(), n, 0, 5)
    s = torch.cuda.current_stream().cuda_stream
    h = vp()
    assert d.comms_fir_create(taps.ctypes.data, taps.size, None, 0, 0, C.byref(h)) == 0
    assert d.comms_fir_set_algo(h, c.FIR_OS16K) == 0
    try:
        assert d.comms_fir_run_dev(h, x.data_ptr(), n, y.data_ptr(), s) == 0
        torch.cuda.synchronize()
        good = y.clone()
        ref = c.BatchFirNode(taps).set_algo(c.FIR_OS16K)  # the product build agrees with the diagnostic one
        ref.run_dev(x.data_ptr(), n, y.data_ptr(), s)
        torch.cuda.synchronize()
        assert torch.equal(good, y)
        state = np.zeros(2100, np.complex64)
        assert d.comms_fir_get_state(h, state.ctypes.data, state.size) == 0
        d.comms_debug_os16k_fault(1)
        assert d.comms_fir_run_dev(h, x.data_ptr(), n, y.data_ptr(), s) == 0  # the launch itself is fine ...
        torch.cuda.synchronize()                                              # ... its waits run out (a few tenths of a second)
        d.comms_debug_os16k_fault(0)
        st = d.comms_fir_run_dev(h, x.data_ptr(), n, y.data_ptr(), s)
        assert st == c.COMMS_ERR_DEVICE, st
        assert b"wait ran out" in d.comms_last_error()
        assert d.comms_fir_get_state(h, state.ctypes.data, state.size) == c.COMMS_ERR_DEVICE
        # another handle is untouched
        h2 = vp()
        assert d.comms_fir_create(taps.ctypes.data, taps.size, None, 0, 0, C.byref(h2)) == 0
        assert d.comms_fir_set_algo(h2, c.FIR_OS16K) == 0
        assert d.comms_fir_run_dev(h2, x.data_ptr(), n, y.data_ptr(), s) == 0
        torch.cuda.synchronize()
        assert torch.equal(good, y)
        assert d.comms_fir_destroy(h2) == 0
    finally:
        d.comms_debug_os16k_fault(0)
        assert d.comms_fir_destroy(h) == 0


def test_os16k_failed_wait_fails_the_host_call_that_hit_it_and_ends_in_bounded_time():
    """The host-pointer entry (what BatchFirNode::run of the Rust / C++ / Python front ends calls) synchronises: the call
    whose launch ran out of its waits must itself return COMMS_ERR_DEVICE, not the invalid samples with COMMS_OK.  And a
    launch with MANY segments per workgroup must not pay a timeout per segment: a wave whose wait ran out leaves its
    segment loop (its signals stop, the other waves follow), so the launch ends after a few timeouts."""
    import time

    import comms_rs_amd as c

    path = os.path.join(os.path.dirname(c.LIB_PATH), "libcomms_hip_diag.so")
    d = C.CDLL(path)
    vp, sz = C.c_void_p, C.c_size_t
    d.comms_fir_create.argtypes = [vp, sz, vp, sz, C.c_int32, C.POINTER(vp)]
    d.comms_fir_set_algo.argtypes = [vp, C.c_int32]
    d.comms_fir_run.argtypes = [vp, vp, sz, vp]
    d.comms_fir_set_state.argtypes = [vp, vp, sz]
    d.comms_fir_destroy.argtypes = [vp]
    d.comms_last_error.restype = C.c_char_p
    k = np.arange(2100) - 1049.5
    taps = np.ascontiguousarray((0.02 * np.sinc(0.02 * k) * np.hamming(2100)).astype(np.float32).astype(np.complex64))
    n = 13312 * 256 * 6  # six segments for every workgroup of the chip
    x = c.synth_iq(n, 0, 7)
    y = np.empty_like(x)
    h = vp()
    assert d.comms_fir_create(taps.ctypes.data, taps.size, None, 0, 0, C.byref(h)) == 0
    assert d.comms_fir_set_algo(h, c.FIR_OS16K) == 0
    try:
        assert d.comms_fir_run(h, x.ctypes.data, n, y.ctypes.data) == 0
        good = y.copy()
        d.comms_debug_os16k_fault(1)
        t0 = time.perf_counter()
        st = d.comms_fir_run(h, x.ctypes.data, n, y.ctypes.data)
        dt = time.perf_counter() - t0
        d.comms_debug_os16k_fault(0)
        assert st == c.COMMS_ERR_DEVICE, "the failing call itself must fail (got %d)" % st
        assert b"wait ran out" in d.comms_last_error()
        assert dt < 6.0, "a failed launch took %.1f s: a timeout per segment?" % dt
        state = np.zeros(2100, np.complex64)
        assert d.comms_fir_set_state(h, state.ctypes.data, state.size) == c.COMMS_ERR_DEVICE  # the handle stays unusable
        # a fresh handle works
        h2 = vp()
        assert d.comms_fir_create(taps.ctypes.data, taps.size, None, 0, 0, C.byref(h2)) == 0
        assert d.comms_fir_set_algo(h2, c.FIR_OS16K) == 0
        assert d.comms_fir_run(h2, x.ctypes.data, n, y.ctypes.data) == 0
        assert np.array_equal(good, y)
        assert d.comms_fir_destroy(h2) == 0
    finally:
        d.comms_debug_os16k_fault(0)
        assert d.comms_fir_destroy(h) == 0
