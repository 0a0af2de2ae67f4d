"""Node classes over the C ABI, named and parameterised like the reference's.

Every class has
  run(x)                  host numpy in -> host numpy out (H2D + kernel + D2H)
  run_dev(in_ptr, n, out_ptr, stream=0)
                          raw device pointers (ints), asynchronous on `stream`
                          (a hipStream_t as int; 0 = HIP's legacy default stream,
                          _lib.STREAM_HANDLE = the handle's own stream).
torch tensors are not part of this API: callers pass tensor.data_ptr().
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib

_c64 = np.dtype(np.complex64)


def _as_c64(a):
    a = np.ascontiguousarray(a, dtype=np.complex64)
    return a


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


_IQ = {"c32": (0, np.complex64, 1), "i16": (1, np.int16, 2), "u8": (2, np.uint8, 2)}


def _as_input(a, fmt):
    """Contiguous input array of a node whose input format is `fmt`, and its sample count."""
    code, dtype, per = _IQ[fmt]
    a = np.ascontiguousarray(a, dtype=dtype)
    return a, a.size // per


def device_count():
    n = C.c_int32(0)
    st = lib().comms_device_count(C.byref(n))
    return n.value if st == 0 else 0


class _Handle:
    _destroy = None

    def __init__(self):
        self._h = C.c_void_p()

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            getattr(lib(), self._destroy)(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------ device buffers
class DeviceBuf:
    """Ref-counted device allocation (comms_buf_*): the device-resident message type."""

    def __init__(self, nbytes, device=0, _h=None):
        if _h is not None:
            self._h = _h
        else:
            self._h = C.c_void_p()
            check(lib().comms_buf_alloc(nbytes, device, C.byref(self._h)))

    @property
    def ptr(self):
        return lib().comms_buf_ptr(self._h) or 0

    @property
    def nbytes(self):
        return lib().comms_buf_size(self._h)

    def clone(self):
        """Rust `Clone`: bump the refcount, share the allocation."""
        check(lib().comms_buf_retain(self._h))
        return DeviceBuf(0, _h=self._h)

    def upload(self, arr, offset=0):
        arr = np.ascontiguousarray(arr)
        check(lib().comms_buf_upload(self._h, offset, _ptr(arr), arr.nbytes))
        return self

    def download(self, dtype, count, offset=0):
        out = np.empty(count, dtype)
        check(lib().comms_buf_download(self._h, offset, _ptr(out), out.nbytes))
        return out

    def release(self):
        if self._h:
            check(lib().comms_buf_release(self._h))
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


# ------------------------------------------------------------------ FIR
class BatchFirNode(_Handle):
    """BatchFirNode::new(taps, state) / run (fir_node.rs:193-220)."""
    _destroy = "comms_fir_destroy"

    def __init__(self, taps, state=None, device=0):
        super().__init__()
        taps = _as_c64(taps)
        if state is None:
            check(lib().comms_fir_create(_ptr(taps), taps.size, None, 0, device, C.byref(self._h)))
        else:
            state = _as_c64(state)
            check(lib().comms_fir_create(_ptr(taps), taps.size, _ptr(state), state.size, device, C.byref(self._h)))

    def set_algo(self, algo):
        check(lib().comms_fir_set_algo(self._h, algo))
        return self

    def algo_for(self, n):
        a = C.c_int32()
        check(lib().comms_fir_get_algo(self._h, n, C.byref(a)))
        return a.value

    def kernel_for(self, n):
        """Name of the kernel a batch of n samples is run by (diagnostics)."""
        buf = C.create_string_buffer(64)
        check(lib().comms_fir_get_kernel(self._h, n, buf, 64))
        return buf.value.decode()

    _fmt = "c32"

    def set_input_format(self, fmt, scale=1.0):
        """Raw-IQ input: "c32" (default), "i16" (interleaved int16 re/im, times `scale`) or "u8"
        (RTL-SDR bytes, (x - 127.5) / 127.5).  run() then takes an integer array of 2 n values."""
        check(lib().comms_fir_set_input_format(self._h, _IQ[fmt][0], float(scale)))
        self._fmt = fmt
        return self

    def run(self, x):
        x, n = _as_input(x, self._fmt)
        out = np.empty(n, np.complex64)
        check(lib().comms_fir_run(self._h, _ptr(x), n, _ptr(out)))
        return out

    def run_dev(self, in_ptr, n, out_ptr, stream=0):
        check(lib().comms_fir_run_dev(self._h, in_ptr, n, out_ptr, stream))

    def state(self, n_state):
        st = np.empty(n_state, np.complex64)
        check(lib().comms_fir_get_state(self._h, _ptr(st), n_state))
        return st

    def set_state(self, state):
        state = _as_c64(state)
        check(lib().comms_fir_set_state(self._h, _ptr(state), state.size))


class FirNode(BatchFirNode):
    """FirNode::new(taps, state) / run(&Complex<T>) (fir_node.rs:89-113): one sample per call."""

    def run(self, x):
        return BatchFirNode.run(self, np.array([x], np.complex64))[0]


class PulseNode(_Handle):
    """PulseNode::new(taps, sam_per_sym) / run (pulse.rs:71-92)."""
    _destroy = "comms_pulse_destroy"

    def __init__(self, taps, sam_per_sym, device=0):
        super().__init__()
        taps = _as_c64(taps)
        self.sam_per_sym = int(sam_per_sym)
        check(lib().comms_pulse_create(_ptr(taps), taps.size, self.sam_per_sym, device, C.byref(self._h)))

    def set_mixer(self, dphase, phase=None):
        """Fuse the MixerNode::new(dphase, phase) that follows this node into its launch (transmit chain)."""
        check(lib().comms_pulse_set_mixer(self._h, float(dphase), 0.0 if phase is None else float(phase)))
        return self

    @property
    def phase(self):
        """Oscillator phase of the fused mixer for the next output sample."""
        p = C.c_double()
        check(lib().comms_pulse_get_phase(self._h, C.byref(p)))
        return p.value

    _out_i16 = False

    def set_output_format(self, fmt, scale=1.0):
        """"i16": run() returns int16 (n, 2) = `(scale * y) as i16`, the IQOutput wire format, written by the
        kernel's store stage; "c32" restores the default."""
        check(lib().comms_pulse_set_output_format(self._h, _IQ[fmt][0], float(scale)))
        self._out_i16 = fmt == "i16"
        return self

    def run(self, sym):
        """One symbol (scalar) -> sam_per_sym samples, or a batch of symbols."""
        s = _as_c64(np.atleast_1d(sym))
        if self._out_i16:
            out = np.empty((s.size * self.sam_per_sym, 2), np.int16)
        else:
            out = np.empty(s.size * self.sam_per_sym, np.complex64)
        check(lib().comms_pulse_run(self._h, _ptr(s), s.size, _ptr(out)))
        return out

    def run_dev(self, sym_ptr, n_sym, out_ptr, stream=0):
        check(lib().comms_pulse_run_dev(self._h, sym_ptr, n_sym, out_ptr, stream))


# ------------------------------------------------------------------ Complex<i16> instantiations
def _as_c16(a):
    a = np.ascontiguousarray(a, dtype=np.int16)
    return a.reshape(-1, 2)


class BatchFirNodeI16(_Handle):
    """BatchFirNode<i16>::new(taps, state) / run (fir_node.rs:193-220) on Complex<i16> = int16 (n, 2) arrays,
    wrapping arithmetic.  A single sample (shape (2,)) in gives a single sample out: FirNode<i16>."""
    _destroy = "comms_fir_i16_destroy"

    def __init__(self, taps, state=None, device=0):
        super().__init__()
        taps = _as_c16(taps)
        if state is None:
            check(lib().comms_fir_i16_create(_ptr(taps), taps.shape[0], None, 0, device, C.byref(self._h)))
        else:
            state = _as_c16(state)
            check(lib().comms_fir_i16_create(_ptr(taps), taps.shape[0], _ptr(state), state.shape[0], device, C.byref(self._h)))

    def run(self, x):
        single = np.ndim(x) == 1 and np.size(x) == 2
        x = _as_c16(x)
        out = np.empty_like(x)
        check(lib().comms_fir_i16_run(self._h, _ptr(x), x.shape[0], _ptr(out)))
        return out[0] if single else out

    def run_dev(self, in_ptr, n, out_ptr, stream=0):
        check(lib().comms_fir_i16_run_dev(self._h, in_ptr, n, out_ptr, stream))

    def state(self, n_state):
        st = np.empty((int(n_state), 2), np.int16)
        check(lib().comms_fir_i16_get_state(self._h, _ptr(st), int(n_state)))
        return st


FirNodeI16 = BatchFirNodeI16


class BatchFirNodeF64(_Handle):
    """BatchFirNode<f64>::new(taps, state) / run (fir_node.rs:193-220) on Complex<f64> = numpy complex128: the reference's
    arithmetic operation for operation, bit-identical outputs (comms_fir_f64_*).  A scalar in gives a scalar out: FirNode<f64>."""
    _destroy = "comms_fir_f64_destroy"

    def __init__(self, taps, state=None, device=0):
        super().__init__()
        taps = np.ascontiguousarray(taps, dtype=np.complex128).ravel()
        if state is None:
            check(lib().comms_fir_f64_create(_ptr(taps), taps.size, None, 0, device, C.byref(self._h)))
        else:
            state = np.ascontiguousarray(state, dtype=np.complex128).ravel()
            check(lib().comms_fir_f64_create(_ptr(taps), taps.size, _ptr(state), state.size, device, C.byref(self._h)))

    def run(self, x):
        scalar = np.ndim(x) == 0
        x = np.ascontiguousarray(np.atleast_1d(x), dtype=np.complex128)
        out = np.empty_like(x)
        check(lib().comms_fir_f64_run(self._h, _ptr(x), x.size, _ptr(out)))
        return out[0] if scalar else out

    def run_dev(self, in_ptr, n, out_ptr, stream=0):
        check(lib().comms_fir_f64_run_dev(self._h, in_ptr, n, out_ptr, stream))

    def state(self, n_state):
        st = np.empty(int(n_state), np.complex128)
        check(lib().comms_fir_f64_get_state(self._h, _ptr(st), int(n_state)))
        return st

    def set_state(self, state):
        st = np.ascontiguousarray(state, dtype=np.complex128).ravel()
        check(lib().comms_fir_f64_set_state(self._h, _ptr(st), st.size))
        return self


FirNodeF64 = BatchFirNodeF64


class PulseNodeF64(_Handle):
    """PulseNode<f64>::new(taps, sam_per_sym) / run (pulse.rs:71-92) on Complex<f64>, bit-identical to the reference."""
    _destroy = "comms_pulse_f64_destroy"

    def __init__(self, taps, sam_per_sym, device=0):
        super().__init__()
        taps = np.ascontiguousarray(taps, dtype=np.complex128).ravel()
        self.sam_per_sym = int(sam_per_sym)
        check(lib().comms_pulse_f64_create(_ptr(taps), taps.size, self.sam_per_sym, device, C.byref(self._h)))

    def run(self, sym):
        s = np.ascontiguousarray(np.atleast_1d(sym), dtype=np.complex128)
        out = np.empty(s.size * self.sam_per_sym, np.complex128)
        check(lib().comms_pulse_f64_run(self._h, _ptr(s), s.size, _ptr(out)))
        return out

    def run_dev(self, sym_ptr, n_sym, out_ptr, stream=0):
        check(lib().comms_pulse_f64_run_dev(self._h, sym_ptr, n_sym, out_ptr, stream))


class PulseNodeI16(_Handle):
    """PulseNode<i16>::new(taps, sam_per_sym) / run (pulse.rs:71-92) on Complex<i16>."""
    _destroy = "comms_pulse_i16_destroy"

    def __init__(self, taps, sam_per_sym, device=0):
        super().__init__()
        taps = _as_c16(taps)
        self.sam_per_sym = int(sam_per_sym)
        check(lib().comms_pulse_i16_create(_ptr(taps), taps.shape[0], self.sam_per_sym, device, C.byref(self._h)))

    def run(self, sym):
        s = _as_c16(sym)
        out = np.empty((s.shape[0] * self.sam_per_sym, 2), np.int16)
        check(lib().comms_pulse_i16_run(self._h, _ptr(s), s.shape[0], _ptr(out)))
        return out

    def run_dev(self, sym_ptr, n_sym, out_ptr, stream=0):
        check(lib().comms_pulse_i16_run_dev(self._h, sym_ptr, n_sym, out_ptr, stream))


# ------------------------------------------------------------------ mixer
class MixerNode(_Handle):
    """MixerNode::new(dphase, phase) (mixer.rs:128-141); run() mixes a slice."""
    _destroy = "comms_mixer_destroy"

    def __init__(self, dphase, phase=None, device=0):
        super().__init__()
        check(lib().comms_mixer_create(float(dphase), 0.0 if phase is None else float(phase), device, C.byref(self._h)))

    def run(self, x):
        """Complex<f32> samples -> MixerNode<f32>; a numpy complex128 array (or scalar) -> MixerNode<f64>, Complex<f64>
        out: the sample type follows the input's dtype, as the reference's generic `MixerNode<T>` (mixer.rs:93) -- note that
        numpy's DEFAULT complex dtype is complex128.  Lists, Python scalars and every other dtype are converted to
        complex64 (tests/test_gpu_parity.py::test_mixer_run_dtype_follows_the_input_dtype pins this)."""
        scalar = np.ndim(x) == 0
        if isinstance(x, (np.ndarray, np.generic)) and x.dtype == np.complex128:
            a = np.ascontiguousarray(np.atleast_1d(x), dtype=np.complex128)
            out = np.empty_like(a)
            check(lib().comms_mixer_run_f64(self._h, _ptr(a), a.size, _ptr(out)))
            return out[0] if scalar else out
        a = _as_c64(np.atleast_1d(x))
        out = np.empty_like(a)
        check(lib().comms_mixer_run(self._h, _ptr(a), a.size, _ptr(out)))
        return out[0] if scalar else out

    def run_dev(self, in_ptr, n, out_ptr, stream=0):
        check(lib().comms_mixer_run_dev(self._h, in_ptr, n, out_ptr, stream))

    def run_f64_dev(self, in_ptr, n, out_ptr, stream=0):
        check(lib().comms_mixer_run_f64_dev(self._h, in_ptr, n, out_ptr, stream))

    @property
    def phase(self):
        p = C.c_double()
        check(lib().comms_mixer_get_phase(self._h, C.byref(p)))
        return p.value

    @phase.setter
    def phase(self, value):
        """Phase of the next sample (checkpoint restore / start phase of a stream shard)."""
        check(lib().comms_mixer_set_phase(self._h, float(value)))


# ------------------------------------------------------------------ resampling
class DecimateNode:
    """DecimateNode::new(dec_rate) / decimate (resample_node.rs:23, :53-65)."""

    def __init__(self, dec_rate, device=0):
        self.dec_rate, self.device = int(dec_rate), device

    def out_len(self, n):
        m = C.c_size_t()
        check(lib().comms_decimate_out_len(n, self.dec_rate, C.byref(m)))
        return m.value

    def run(self, x):
        x = np.ascontiguousarray(x)
        elem = x.dtype.itemsize * int(np.prod(x.shape[1:], dtype=np.int64))
        out = np.empty((self.out_len(x.shape[0]),) + x.shape[1:], x.dtype)
        m = C.c_size_t()
        check(lib().comms_decimate_run(_ptr(x), x.shape[0], elem, self.dec_rate, _ptr(out), C.byref(m), self.device))
        assert m.value == out.shape[0]
        return out

    decimate = run

    def run_dev(self, in_ptr, n, elem, out_ptr, stream=0):
        m = C.c_size_t()
        check(lib().comms_decimate_run_dev(in_ptr, n, elem, self.dec_rate, out_ptr, C.byref(m), self.device, stream))
        return m.value


class UpsampleNode:
    """UpsampleNode::new(ups_rate) / upsample (resample_node.rs:87, :120-131)."""

    def __init__(self, ups_rate, device=0):
        self.ups_rate, self.device = int(ups_rate), device

    def out_len(self, n):
        m = C.c_size_t()
        check(lib().comms_upsample_out_len(n, self.ups_rate, C.byref(m)))
        return m.value

    def run(self, x):
        x = np.ascontiguousarray(x)
        elem = x.dtype.itemsize * int(np.prod(x.shape[1:], dtype=np.int64))
        out = np.empty((self.out_len(x.shape[0]),) + x.shape[1:], x.dtype)
        m = C.c_size_t()
        check(lib().comms_upsample_run(_ptr(x), x.shape[0], elem, self.ups_rate, _ptr(out), C.byref(m), self.device))
        assert m.value == out.shape[0]
        return out

    upsample = run

    def run_dev(self, in_ptr, n, elem, out_ptr, stream=0):
        m = C.c_size_t()
        check(lib().comms_upsample_run_dev(in_ptr, n, elem, self.ups_rate, out_ptr, C.byref(m), self.device, stream))
        return m.value


# ------------------------------------------------------------------ FM demod
class FMDemodNode(_Handle):
    """FMDemodNode::new() / run (analog_node.rs:43-51)."""
    _destroy = "comms_fmdemod_destroy"

    def __init__(self, device=0):
        super().__init__()
        check(lib().comms_fmdemod_create(device, C.byref(self._h)))

    def run(self, x):
        x = _as_c64(x)
        out = np.empty(x.size, np.float32)
        check(lib().comms_fmdemod_run(self._h, _ptr(x), x.size, _ptr(out)))
        return out

    def run_dev(self, in_ptr, n, out_ptr, stream=0):
        check(lib().comms_fmdemod_run_dev(self._h, in_ptr, n, out_ptr, stream))

    @property
    def prev(self):
        """FM.prev (analog.rs:9,31): the last input sample of the previous batch."""
        p = np.zeros(1, np.complex64)
        check(lib().comms_fmdemod_get_prev(self._h, _ptr(p)))
        return p[0]

    @prev.setter
    def prev(self, value):
        p = np.array([value], np.complex64)
        check(lib().comms_fmdemod_set_prev(self._h, _ptr(p)))


# ------------------------------------------------------------------ FFT
class FFTBatchNode(_Handle):
    """FFTBatchNode::new(fft_size, ifft) / run (fft_node.rs:65-83)."""
    _destroy = "comms_fft_destroy"

    def __init__(self, fft_size, ifft, device=0):
        super().__init__()
        self.fft_size = int(fft_size)
        check(lib().comms_fft_create(self.fft_size, 1 if ifft else 0, device, C.byref(self._h)))

    def run(self, x):
        x = _as_c64(x)
        out = np.empty_like(x)
        check(lib().comms_fft_run(self._h, _ptr(x), x.size, _ptr(out)))
        return out

    def run_dev(self, in_ptr, n, out_ptr, stream=0):
        check(lib().comms_fft_run_dev(self._h, in_ptr, n, out_ptr, stream))


class FFTBatchNodeF64(_Handle):
    """FFTBatchNode<f64>::new(fft_size, ifft) / run (fft_node.rs:65-83) on Complex<f64> = numpy complex128 (comms_fft_f64_*):
    the instantiation of the reference's doc examples; correct to f64 rounding."""
    _destroy = "comms_fft_f64_destroy"

    def __init__(self, fft_size, ifft, device=0):
        super().__init__()
        self.fft_size = int(fft_size)
        check(lib().comms_fft_f64_create(self.fft_size, 1 if ifft else 0, device, C.byref(self._h)))

    def run(self, x):
        x = np.ascontiguousarray(x, dtype=np.complex128).ravel()
        out = np.empty_like(x)
        check(lib().comms_fft_f64_run(self._h, _ptr(x), x.size, _ptr(out)))
        return out

    def run_dev(self, in_ptr, n, out_ptr, stream=0):
        check(lib().comms_fft_f64_run_dev(self._h, in_ptr, n, out_ptr, stream))


class FFTSampleNodeF64(FFTBatchNodeF64):
    """FFTSampleNode<f64> (fft_node.rs:142-167), #[aggregate]: a sample per call, the spectrum every fft_size samples."""

    def __init__(self, fft_size, ifft, device=0):
        super().__init__(fft_size, ifft, device)
        self._samples = []

    def run(self, sample):
        self._samples.append(np.complex128(sample))
        if len(self._samples) == self.fft_size:
            res = FFTBatchNodeF64.run(self, np.array(self._samples, np.complex128))
            self._samples = []
            return res
        return None


class FMDemodNodeF64(_Handle):
    """FMDemodNode<f64>::new() / run (analog_node.rs:20-52; FM::demod analog.rs:22-35) on Complex<f64>."""
    _destroy = "comms_fmdemod_f64_destroy"

    def __init__(self, device=0):
        super().__init__()
        check(lib().comms_fmdemod_f64_create(device, C.byref(self._h)))

    def run(self, x):
        x = np.ascontiguousarray(x, dtype=np.complex128).ravel()
        out = np.empty(x.size, np.float64)
        check(lib().comms_fmdemod_f64_run(self._h, _ptr(x), x.size, _ptr(out)))
        return out

    def run_dev(self, in_ptr, n, out_ptr, stream=0):
        check(lib().comms_fmdemod_f64_run_dev(self._h, in_ptr, n, out_ptr, stream))

    @property
    def prev(self):
        p = np.zeros(1, np.complex128)
        check(lib().comms_fmdemod_f64_get_prev(self._h, _ptr(p)))
        return p[0]

    @prev.setter
    def prev(self, value):
        p = np.array([value], np.complex128)
        check(lib().comms_fmdemod_f64_set_prev(self._h, _ptr(p)))


class FFTSampleNode(FFTBatchNode):
    """FFTSampleNode::new(fft_size, ifft) / run (fft_node.rs:142-167), #[aggregate]:
    push one sample; returns None until fft_size samples arrived, then the FFT."""

    def __init__(self, fft_size, ifft, device=0):
        super().__init__(fft_size, ifft, device)
        self._samples = []

    def run(self, sample):
        self._samples.append(np.complex64(sample))
        if len(self._samples) == self.fft_size:
            res = FFTBatchNode.run(self, np.array(self._samples, np.complex64))
            self._samples = []
            return res
        return None


# ------------------------------------------------------------------ fused chain
class ChainNode(_Handle):
    """mixer / FIR / decimate [/ FM demod] as one node (comms_chain_*), an additional node.
    mixer_after_fir=False: mixer -> FIR -> decimate [-> FM]; True: FIR -> mixer -> decimate."""
    _destroy = "comms_chain_destroy"

    def __init__(self, dphase, phase, taps, rate, fm_demod, device=0, mixer_after_fir=False, unfused=False,
                 kernel="auto"):
        """kernel: "auto", "freq" (always an overlap-save kernel: the 1024-point one up to 257 taps, the 4096- / 16384-point
        ones with mixer and decimator in their store stage up to 1537 / 4097), "time" (the decimating time-domain kernels wherever
        they apply) or "poly" (the polyphase frequency-domain kernel: rates 4, 8, 12 ... 64, <= 513 taps, 505 with FM demod)."""
        super().__init__()
        taps = _as_c64(taps)
        self.rate, self.fm_demod = int(rate), bool(fm_demod)
        flags = (1 if fm_demod else 0) | (2 if mixer_after_fir else 0) | (4 if unfused else 0)
        flags |= {"auto": 0, "freq": 8, "time": 16, "poly": 32}[kernel]
        check(lib().comms_chain_create_ex(float(dphase), float(phase), _ptr(taps), taps.size, self.rate,
                                          flags, device, C.byref(self._h)))

    @property
    def fused(self):
        f = C.c_int32()
        check(lib().comms_chain_is_fused(self._h, C.byref(f)))
        return bool(f.value)

    @property
    def kernel(self):
        """"unfused", "freq" (fir_os1024_kernel; fir_os4096_kernel / fir_os16k_kernel in their decimating form for 258 ... 4097
        taps), "time" (fir_decim_kernel / fir_decim_wave_kernel), "time_any" (fir_decim_any_kernel) or "poly" (fir_poly8_kernel:
        forced, picked at creation for 258 ... 513 taps, or what the chain's last call ran on)."""
        f = C.c_int32()
        check(lib().comms_chain_is_fused(self._h, C.byref(f)))
        return ("unfused", "freq", "time", "time_any", "poly")[f.value]

    _fmt = "c32"

    def set_input_format(self, fmt, scale=1.0):
        """Raw-IQ input ("c32" / "i16" / "u8", see BatchFirNode.set_input_format): converted in the load
        stage of the time-domain chain kernel, by one extra pass in front of the other chain forms."""
        check(lib().comms_chain_set_input_format(self._h, _IQ[fmt][0], float(scale)))
        self._fmt = fmt
        return self

    def run(self, x):
        x, n = _as_input(x, self._fmt)
        out = np.empty(n // self.rate, np.float32 if self.fm_demod else np.complex64)
        check(lib().comms_chain_run(self._h, _ptr(x), n, _ptr(out)))
        return out

    def run_dev(self, in_ptr, n, out_ptr, stream=0):
        check(lib().comms_chain_run_dev(self._h, in_ptr, n, out_ptr, stream))

    def set_fir_state(self, state):
        """FIR history, reference layout (newest first) -- e.g. the halo of a sharded stream."""
        st = _as_c64(state)
        check(lib().comms_chain_set_fir_state(self._h, _ptr(st), st.size))
        return self

    def fir_state(self, n_state):
        st = np.empty(int(n_state), np.complex64)
        check(lib().comms_chain_get_fir_state(self._h, _ptr(st), st.size))
        return st

    @property
    def phase(self):
        """Oscillator phase of the next input sample."""
        p = C.c_double()
        check(lib().comms_chain_get_phase(self._h, C.byref(p)))
        return p.value

    @phase.setter
    def phase(self, value):
        check(lib().comms_chain_set_phase(self._h, float(value)))

    @property
    def fm_prev(self):
        """FM.prev of the chain's demodulator: the last decimated filter output of the previous batch."""
        p = np.zeros(1, np.complex64)
        check(lib().comms_chain_get_fm_prev(self._h, _ptr(p)))
        return p[0]

    @fm_prev.setter
    def fm_prev(self, value):
        p = np.array([value], np.complex64)
        check(lib().comms_chain_set_fm_prev(self._h, _ptr(p)))


# ------------------------------------------------------------------ tap design
def _taps(fn, n_taps, *args):
    out = np.empty(int(n_taps), np.complex64)
    check(fn(int(n_taps), *args, _ptr(out)))
    return out


def rrc_taps(n_taps, sam_per_sym, beta):
    """util/math.rs:221-280; raises CommsError(code 1) for beta outside [0,1]."""
    return _taps(lib().comms_rrc_taps, n_taps, float(sam_per_sym), float(beta))


def rc_taps(n_taps, sam_per_sym, beta):
    return _taps(lib().comms_rc_taps, n_taps, float(sam_per_sym), float(beta))


def gaussian_taps(n_taps, sam_per_sym, alpha):
    return _taps(lib().comms_gaussian_taps, n_taps, float(sam_per_sym), float(alpha))


def rect_taps(n_taps):
    return _taps(lib().comms_rect_taps, n_taps)


# ------------------------------------------------------------------ raw IQ wire formats
def iq_i16_to_c32(x, scale=1.0, device=0):
    """int16 array of shape (n, 2) (re, im) -> complex64 (cast_complex, times scale)."""
    x = np.ascontiguousarray(x, dtype=np.int16).reshape(-1, 2)
    out = np.empty(x.shape[0], np.complex64)
    check(lib().comms_iq_i16_to_c32(_ptr(x), x.shape[0], float(scale), _ptr(out), device))
    return out


def iq_c32_to_i16(x, scale=1.0, device=0):
    """complex64 -> int16 (n, 2): `(scale * x) as i16` per component (Rust `as` semantics)."""
    x = _as_c64(x)
    out = np.empty((x.size, 2), np.int16)
    check(lib().comms_iq_c32_to_i16(_ptr(x), x.size, float(scale), _ptr(out), device))
    return out


def real_to_c32_dev(in_ptr, n, out_ptr, device=0, stream=0):
    """x -> Complex(x, 0) on the device (examples/fm_radio.rs Convert2Node)."""
    check(lib().comms_iq_real_to_c32_dev(in_ptr, n, out_ptr, device, stream))


def c32_re_dev(in_ptr, n, out_ptr, device=0, stream=0):
    """x -> x.re on the device (examples/fm_radio.rs Convert3Node)."""
    check(lib().comms_iq_c32_re_dev(in_ptr, n, out_ptr, device, stream))


def iq_u8_to_c32(x, device=0):
    """uint8 (n, 2) RTL-SDR bytes -> complex64: (x - 127.5) / 127.5."""
    x = np.ascontiguousarray(x, dtype=np.uint8).reshape(-1, 2)
    out = np.empty(x.shape[0], np.complex64)
    check(lib().comms_iq_u8_to_c32(_ptr(x), x.shape[0], _ptr(out), device))
    return out


# ------------------------------------------------------------------ block estimators
def frequency_offset_estimate(samples, device=0):
    """frequency_estimator.rs:27-42 on Complex<f64> samples."""
    x = np.ascontiguousarray(samples, dtype=np.complex128)
    out = C.c_double()
    check(lib().comms_frequency_offset_estimate(_ptr(x), x.size, C.byref(out), device))
    return out.value


def psk_phase_estimate(symbols, m, device=0):
    """phase_estimator.rs:26-33."""
    x = np.ascontiguousarray(symbols, dtype=np.complex128)
    out = C.c_double()
    check(lib().comms_psk_phase_estimate(_ptr(x), x.size, int(m), C.byref(out), device))
    return out.value


def qam_phase_estimate(symbols, device=0):
    """phase_estimator.rs:58-65."""
    x = np.ascontiguousarray(symbols, dtype=np.complex128)
    out = C.c_double()
    check(lib().comms_qam_phase_estimate(_ptr(x), x.size, C.byref(out), device))
    return out.value


class TimingEstimatorNode(_Handle):
    """TimingEstimatorNode::new(n, d, alpha) / run (timing_estimator.rs:116-136): Complex<f64>
    block in, timing offset in samples out."""
    _destroy = "comms_timing_destroy"

    def __init__(self, n, d, alpha, device=0):
        super().__init__()
        check(lib().comms_timing_create(int(n), int(d), float(alpha), device, C.byref(self._h)))

    def run(self, samples):
        x = np.ascontiguousarray(samples, dtype=np.complex128)
        out = C.c_double()
        check(lib().comms_timing_push(self._h, _ptr(x), x.size, C.byref(out)))
        return out.value

    def run_dev(self, in_ptr, n, stream=0):
        out = C.c_double()
        check(lib().comms_timing_push_dev(self._h, in_ptr, n, C.byref(out), stream))
        return out.value


def qfilt_taps(n_taps, alpha, sam_per_sym):
    """util/math.rs:307-342 (f64, real); even n_taps is incremented."""
    out = np.empty(lib().comms_qfilt_len(int(n_taps)), np.float64)
    check(lib().comms_qfilt_taps(int(n_taps), float(alpha), int(sam_per_sym), _ptr(out)))
    return out


class NcoNode(_Handle):
    """NcoNode::new(dphase, phase) (nco.rs:118-133) in block form: run() takes a vector of
    phase errors and returns exp(i*phase) per sample (Complex<f64>)."""
    _destroy = "comms_nco_destroy"

    def __init__(self, dphase, phase=None, device=0):
        super().__init__()
        check(lib().comms_nco_create(float(dphase), 0.0 if phase is None else float(phase), device,
                                     C.byref(self._h)))

    def run(self, perr):
        e = np.ascontiguousarray(perr, dtype=np.float64)
        out = np.empty(e.size, np.complex128)
        check(lib().comms_nco_run(self._h, _ptr(e), e.size, _ptr(out)))
        return out

    def run_dev(self, in_ptr, n, out_ptr, stream=0):
        check(lib().comms_nco_run_dev(self._h, in_ptr, n, out_ptr, stream))

    @property
    def phase(self):
        out = C.c_double()
        check(lib().comms_nco_get_phase(self._h, C.byref(out)))
        return out.value


# ------------------------------------------------------------------ synthetic IQ
def synth_iq(n, first_index=0, seed=0xC0FFEE):
    out = np.empty(int(n), np.complex64)
    lib().comms_synth_iq_host(_ptr(out), out.size, first_index, seed)
    return out


def synth_iq_dev(out_ptr, n, first_index=0, seed=0xC0FFEE, device=0, stream=0):
    check(lib().comms_synth_iq_dev(out_ptr, n, first_index, seed, device, stream))


# ------------------------------------------------------------------ kernel timer
class KernelTimer:
    """comms_timer_*: hipEvent pairs recorded around a node's dominant kernel."""

    def __init__(self, n_pairs, device=0, stamps=False, stride=1):
        """stamps=False: hipEvent pairs (every `stride`-th launch of the attached node is bracketed);
        stamps=True: comms_timer_create_stamps -- the kernels stamp their own begin / end, no events, every launch an
        ordinary one (the kernel's duration in the stream); stamps="both": events every stride-th launch AND stamps
        on every launch (read_ms / read_stamps_ms)."""
        self._h = C.c_void_p()
        self.n = int(n_pairs)
        if stamps is True:
            check(lib().comms_timer_create_stamps(self.n, device, C.byref(self._h)))
        else:
            check(lib().comms_timer_create(self.n, device, C.byref(self._h)))
            if stamps == "both":
                check(lib().comms_timer_add_stamps(self._h, self.n))
            if stride != 1:
                check(lib().comms_timer_set_stride(self._h, int(stride)))

    def attach(self, node):
        name = {"comms_fir_destroy": "comms_fir_set_timer", "comms_mixer_destroy": "comms_mixer_set_timer",
                "comms_fmdemod_destroy": "comms_fmdemod_set_timer", "comms_fft_destroy": "comms_fft_set_timer",
                "comms_chain_destroy": "comms_chain_set_timer", "comms_pulse_destroy": "comms_pulse_set_timer"}[node._destroy]
        check(getattr(lib(), name)(node._h, self._h))
        self._node, self._setter = node, name
        return self

    def enable(self, on):
        """Detach from / re-attach to the node without touching what was recorded (to bracket only some launches)."""
        on = bool(on)
        if on != getattr(self, "_on", True):
            check(getattr(lib(), self._setter)(self._node._h, self._h if on else None))
            self._on = on

    def reset(self):
        check(lib().comms_timer_reset(self._h))

    def read_ms(self):
        out = np.zeros(self.n, np.float32)
        m = C.c_size_t()
        check(lib().comms_timer_read(self._h, _ptr(out), self.n, C.byref(m)))
        return out[:m.value].copy()

    def read_stamps_ms(self):
        """In-stream kernel durations of the stamped launches (0.0 where the launched kernel does not stamp)."""
        out = np.zeros(self.n, np.float32)
        m = C.c_size_t()
        check(lib().comms_timer_read_stamps(self._h, _ptr(out), self.n, C.byref(m)))
        return out[:m.value].copy()

    def close(self):
        if self._h:
            node = getattr(self, "_node", None)
            if node is not None and node._h:
                getattr(lib(), self._setter)(node._h, None)
            lib().comms_timer_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
