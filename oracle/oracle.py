"""ctypes/numpy bindings of liboracle.so (built from oracle.cpp by `make`).

TEST INFRASTRUCTURE ONLY -- see oracle.cpp.  Complex arrays are numpy
complex64 / complex128 (interleaved re,im == num::Complex<T> #[repr(C)]),
Complex<i16> is an int16 array of shape (n, 2).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_SRC = os.path.join(_HERE, "oracle.cpp")


def build(force=False):
    """Compile liboracle.so with the committed Makefile (gcc only)."""
    stale = (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(_SRC)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_mixer_wrap_dphase.restype = C.c_double
        _lib.orc_mixer_wrap_dphase.argtypes = [C.c_double]
        _lib.orc_sinc.restype = C.c_double
        _lib.orc_sinc.argtypes = [C.c_double]
        _lib.orc_decimate.restype = C.c_size_t
        _lib.orc_upsample.restype = C.c_size_t
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _sz(n):
    return C.c_size_t(int(n))


_CX = {np.dtype(np.complex64): "f32", np.dtype(np.complex128): "f64"}


def _as_c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def default_state(taps):
    """fir_node.rs:97-105 / :201-211 -- `None` state = zeros of taps.len()."""
    return np.zeros_like(np.asarray(taps))


# ------------------------------------------------------------------ FIR
def fir(x, taps, state):
    """fir.rs:43-54.  `state` is updated in place; returns one sample."""
    taps = np.ascontiguousarray(taps)
    if taps.dtype == np.int16:
        xx = _as_c(x, np.int16).reshape(2)
        out = np.zeros(2, np.int16)
        lib().orc_fir_i16(_p(xx), _p(taps), _sz(len(taps)), _p(state), _sz(len(state)), _p(out))
        return out
    assert taps.dtype == np.complex64 and state.dtype == np.complex64
    xx = np.array([x], np.complex64)
    out = np.zeros(1, np.complex64)
    lib().orc_fir_f32(_p(xx), _p(taps), _sz(len(taps)), _p(state), _sz(len(state)), _p(out))
    return out[0]


def batch_fir(x, taps, state, norotate=False):
    """fir.rs:87-102.  `state` (len = taps.len() by default) is updated in place.

    norotate=True selects the bit-identical variant without the per-sample
    memmove (f32 only)."""
    taps = np.ascontiguousarray(taps)
    x = np.ascontiguousarray(x, dtype=taps.dtype)
    assert state.dtype == taps.dtype and state.flags.c_contiguous
    out = np.zeros_like(x)
    n = x.shape[0]
    if taps.dtype == np.int16:
        f = lib().orc_batch_fir_i16
    elif taps.dtype == np.complex64:
        f = lib().orc_batch_fir_norotate_f32 if norotate else lib().orc_batch_fir_f32
    elif taps.dtype == np.complex128:
        f = lib().orc_batch_fir_f64
    else:
        raise TypeError(taps.dtype)
    f(_p(x), _sz(n), _p(taps), _sz(len(taps)), _p(state), _sz(len(state)), _p(out))
    return out


def pulse(sym, taps, sam_per_sym, state):
    """pulse.rs:82-92.  Returns n_sym * sam_per_sym samples; state in place."""
    assert sam_per_sym >= 1, "reference underflows (panics) at sam_per_sym == 0"
    taps = np.ascontiguousarray(taps)
    sym = np.ascontiguousarray(sym, dtype=taps.dtype)
    n = sym.shape[0]
    if taps.dtype == np.int16:
        out = np.zeros((n * sam_per_sym, 2), np.int16)
        f = lib().orc_pulse_i16
    elif taps.dtype == np.complex128:
        out = np.zeros(n * sam_per_sym, np.complex128)
        f = lib().orc_pulse_f64
    else:
        assert taps.dtype == np.complex64
        out = np.zeros(n * sam_per_sym, np.complex64)
        f = lib().orc_pulse_f32
    f(_p(sym), _sz(n), _p(taps), _sz(len(taps)), _sz(sam_per_sym), _p(state), _p(out))
    return out


# ------------------------------------------------------------------ mixer
class Mixer:
    """mixer.rs:17-85.  NB argument order (phase, dphase) as in Mixer::new."""

    def __init__(self, phase, dphase):
        self.phase = C.c_double(float(phase))
        self.dphase = lib().orc_mixer_wrap_dphase(float(dphase))

    def mix(self, x):
        x = np.ascontiguousarray(x)
        out = np.zeros_like(x)
        f = {"f32": lib().orc_mixer_f32, "f64": lib().orc_mixer_f64}[_CX[x.dtype]]
        f(_p(x), _sz(x.shape[0]), C.byref(self.phase), C.c_double(self.dphase), _p(out))
        return out


# ------------------------------------------------------------------ FFT
def fft(x, inverse=False):
    """fft/mod.rs:73-96 (f64 inside, unnormalised, cast back to x.dtype)."""
    x = np.ascontiguousarray(x)
    out = np.zeros_like(x)
    f = {"f32": lib().orc_fft_f32, "f64": lib().orc_fft_f64}[_CX[x.dtype]]
    f(_p(x), _sz(x.shape[0]), C.c_int(1 if inverse else 0), _p(out))
    return out


# ------------------------------------------------------------------ resampling
def decimate(x, rate):
    """resample_node.rs:53-65.  Works on the first axis of any dtype."""
    x = np.ascontiguousarray(x)
    n = x.shape[0]
    elem = x.dtype.itemsize * int(np.prod(x.shape[1:], dtype=np.int64))
    n_out = n if rate in (0, 1) else (n + rate - 1) // rate
    out = np.zeros((n_out,) + x.shape[1:], x.dtype)
    got = lib().orc_decimate(_p(x), _sz(n), _sz(elem), _sz(rate), _p(out))
    assert got == n_out
    return out


def upsample(x, rate):
    """resample_node.rs:120-131."""
    x = np.ascontiguousarray(x)
    n = x.shape[0]
    elem = x.dtype.itemsize * int(np.prod(x.shape[1:], dtype=np.int64))
    n_out = n if rate in (0, 1) else n * rate
    out = np.zeros((n_out,) + x.shape[1:], x.dtype)
    got = lib().orc_upsample(_p(x), _sz(n), _sz(elem), _sz(rate), _p(out))
    assert got == n_out
    return out


# ------------------------------------------------------------------ FM demod
class FM:
    """modulation/analog.rs:8-48.  prev starts at 0+0i and persists."""

    def __init__(self, dtype=np.complex64):
        self.prev = np.zeros(1, dtype)

    def demod(self, x):
        x = np.ascontiguousarray(x, dtype=self.prev.dtype)
        real = np.float32 if x.dtype == np.complex64 else np.float64
        out = np.zeros(x.shape[0], real)
        f = {"f32": lib().orc_fm_demod_f32, "f64": lib().orc_fm_demod_f64}[_CX[x.dtype]]
        f(_p(x), _sz(x.shape[0]), _p(self.prev), _p(out))
        return out


# ------------------------------------------------------------------ tap design
class InvalidRolloffError(ValueError):
    """util/mod.rs:8-11 MathError::InvalidRolloffError."""


def _taps(fn, n_taps, args, dtype):
    re = np.zeros(int(n_taps), np.float64)
    rc = fn(*([C.c_uint32(int(n_taps))] + [C.c_double(a) for a in args] + [_p(re)]))
    if rc != 0:
        raise InvalidRolloffError()
    # `T::from(f64)` then Complex::new(re, 0)
    return re.astype(np.float32 if dtype == np.complex64 else np.float64).astype(dtype)


def rrc_taps(n_taps, sam_per_sym, beta, dtype=np.complex64):
    """util/math.rs:221-280."""
    return _taps(lib().orc_rrc_taps, n_taps, (sam_per_sym, beta), dtype)


def rc_taps(n_taps, sam_per_sym, beta, dtype=np.complex64):
    """util/math.rs:151-196."""
    return _taps(lib().orc_rc_taps, n_taps, (sam_per_sym, beta), dtype)


def gaussian_taps(n_taps, sam_per_sym, alpha, dtype=np.complex64):
    """util/math.rs:79-102."""
    return _taps(lib().orc_gaussian_taps, n_taps, (sam_per_sym, alpha), dtype)


def rect_taps(n_taps, dtype=np.complex64):
    """util/math.rs:48-55."""
    if dtype == np.int16:
        t = np.zeros((int(n_taps), 2), np.int16)
        t[:, 0] = 1
        return t
    return np.ones(int(n_taps), dtype)


def sinc(x):
    """util/math.rs:120-126."""
    return lib().orc_sinc(float(x))


# ------------------------------------------------------------------ raw IQ wire formats
def iq_i16_to_c32(x, scale=1.0):
    """raw_iq.rs:16,50-51 + math.rs:20-28 (cast_complex), times scale."""
    x = np.ascontiguousarray(x, dtype=np.int16).reshape(-1, 2)
    out = np.zeros(x.shape[0], np.complex64)
    lib().orc_iq_i16_to_f32(_p(x), _sz(x.shape[0]), C.c_float(scale), _p(out))
    return out


def iq_c32_to_i16(x, scale=1.0):
    """examples/single_thread_bpsk.rs:40-44: `(scale * x) as i16` per component."""
    x = np.ascontiguousarray(x, dtype=np.complex64)
    out = np.zeros((x.size, 2), np.int16)
    lib().orc_iq_f32_to_i16(_p(x), _sz(x.size), C.c_float(scale), _p(out))
    return out


def iq_u8_to_c32(x):
    """examples/fm_radio.rs:82-90."""
    x = np.ascontiguousarray(x, dtype=np.uint8).reshape(-1, 2)
    out = np.zeros(x.shape[0], np.complex64)
    lib().orc_iq_u8_to_f32(_p(x), _sz(x.shape[0]), _p(out))
    return out


# ------------------------------------------------------------------ block estimators (f64)
def frequency_offset_estimate(samples):
    """frequency_estimator.rs:27-42."""
    x = np.ascontiguousarray(samples, dtype=np.complex128)
    lib().orc_frequency_offset_estimate.restype = C.c_double
    return lib().orc_frequency_offset_estimate(_p(x), _sz(x.size))


def psk_phase_estimate(symbols, m):
    """phase_estimator.rs:26-33."""
    x = np.ascontiguousarray(symbols, dtype=np.complex128)
    lib().orc_psk_phase_estimate.restype = C.c_double
    return lib().orc_psk_phase_estimate(_p(x), _sz(x.size), C.c_uint32(int(m)))


def qam_phase_estimate(symbols):
    """phase_estimator.rs:58-65."""
    x = np.ascontiguousarray(symbols, dtype=np.complex128)
    lib().orc_qam_phase_estimate.restype = C.c_double
    return lib().orc_qam_phase_estimate(_p(x), _sz(x.size))


def qfilt_taps(n_taps, alpha, sam_per_sym):
    """util/math.rs:307-342; raises ValueError for alpha outside [0, 1]."""
    out = np.zeros(int(n_taps) | 1, np.float64)
    if lib().orc_qfilt_taps(C.c_uint32(int(n_taps)), C.c_double(alpha), C.c_uint32(int(sam_per_sym)), _p(out)):
        raise ValueError("InvalidRolloffError")
    return out


def timing_push(samples, n, d, alpha):
    """timing_estimator.rs:85-112 (one push of a fresh-state estimator)."""
    x = np.ascontiguousarray(samples, dtype=np.complex128)
    lib().orc_timing_push.restype = C.c_double
    return lib().orc_timing_push(_p(x), _sz(x.size), C.c_uint32(int(n)), C.c_uint32(int(d)), C.c_double(alpha))


class Nco:
    """nco.rs:41-50, :71-77; NcoNode::new(dphase, phase) argument order."""

    def __init__(self, dphase, phase=None):
        self.dphase = float(dphase)
        self.phase = C.c_double(0.0 if phase is None else float(phase))

    def push(self, perr):
        e = np.ascontiguousarray(perr, dtype=np.float64)
        out = np.zeros(e.size, np.complex128)
        lib().orc_nco_push(C.c_double(self.dphase), C.byref(self.phase), _p(e), _sz(e.size), _p(out))
        return out


# ------------------------------------------------------------------ PRBS source
def prns_u8(poly_mask, state, n):
    """prns.rs:64-71 on an 8-bit register.  Returns (bits, new_state)."""
    st = C.c_uint8(state)
    out = np.zeros(int(n), np.uint8)
    lib().orc_prns_u8(C.c_uint8(poly_mask), C.byref(st), _sz(n), _p(out))
    return out, st.value
