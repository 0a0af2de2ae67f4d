"""ctypes loader of libcomms_hip.so -- the C ABI declared in include/comms_hip.h.

The library is built in-tree by comms_rs_amd/csrc/Makefile (hipcc, gfx950).
There is no CPU fallback: if the library is missing this module raises, and
every create call fails with COMMS_ERR_DEVICE when no MI355X is visible.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# COMMS_HIP_LIB selects another build of the same ABI (e.g. lib/libcomms_hip_diag.so, `make diag`)
LIB_PATH = os.environ.get("COMMS_HIP_LIB") or os.path.join(_HERE, "lib", "libcomms_hip.so")
CSRC = os.path.join(_HERE, "csrc")

COMMS_OK, COMMS_ERR_ARG, COMMS_ERR_DEVICE = 0, 1, 2
FIR_AUTO, FIR_DIRECT, FIR_OVERLAP_SAVE, FIR_OS1024, FIR_OS4096, FIR_OS16K, FIR_OS1024_FIXED = 0, 1, 2, 3, 4, 5, 6
IQ_C32, IQ_I16, IQ_U8 = 0, 1, 2
STREAM_HANDLE = C.c_void_p(-1).value  # COMMS_STREAM_HANDLE: the handle's own stream


def build(force=False):
    """Compile every HIP source for gfx950 into lib/libcomms_hip.so (needs hipcc, no GPU)."""
    os.makedirs(os.path.join(_HERE, "lib"), exist_ok=True)
    if force:
        subprocess.check_call(["make", "-C", CSRC, "-s", "clean"])
    subprocess.check_call(["make", "-C", CSRC, "-s", "-j8"])
    return LIB_PATH


class CommsError(RuntimeError):
    """Non-zero comms_status_t.  code 1 ~ NodeError::DataError, 2 ~ NodeError::PermanentError."""

    def __init__(self, code, msg):
        super().__init__("comms_hip status %d: %s" % (code, msg))
        self.code = code


_lib = None

_sz, _i32, _u32, _u64, _f64, _vp = C.c_size_t, C.c_int32, C.c_uint32, C.c_uint64, C.c_double, C.c_void_p
_pp = C.POINTER(C.c_void_p)
_psz = C.POINTER(C.c_size_t)

# name -> argtypes; every one returns comms_status_t unless listed in _OTHER
_PROTOS = {
    "comms_device_count": [C.POINTER(_i32)],
    "comms_device_info": [_i32, C.c_char_p, _sz, C.POINTER(_i32), C.POINTER(_u64)],
    "comms_buf_alloc": [_sz, _i32, _pp],
    "comms_buf_retain": [_vp],
    "comms_buf_release": [_vp],
    "comms_buf_upload": [_vp, _sz, _vp, _sz],
    "comms_buf_download": [_vp, _sz, _vp, _sz],
    "comms_buf_pool_trim": [_i32],
    "comms_buf_record_ready": [_vp, _vp],
    "comms_buf_wait_ready": [_vp, _vp],
    "comms_buf_record_use": [_vp, _vp],
    "comms_buf_sync": [_vp],
    "comms_stream_create": [_i32, _pp],
    "comms_stream_synchronize": [_i32, _vp],
    "comms_stream_destroy": [_i32, _vp],
    "comms_stream_pool_trim": [_i32],
    "comms_timer_create": [_sz, _i32, _pp],
    "comms_timer_create_stamps": [_sz, _i32, _pp],
    "comms_timer_add_stamps": [_vp, _sz],
    "comms_timer_set_stride": [_vp, _sz],
    "comms_timer_read_stamps": [_vp, _vp, _sz, _psz],
    "comms_timer_reset": [_vp],
    "comms_timer_read": [_vp, _vp, _sz, _psz],
    "comms_timer_destroy": [_vp],
    "comms_fir_set_timer": [_vp, _vp],
    "comms_fir_set_input_format": [_vp, _i32, C.c_float],
    "comms_chain_set_input_format": [_vp, _i32, C.c_float],
    "comms_mixer_set_timer": [_vp, _vp],
    "comms_pulse_set_timer": [_vp, _vp],
    "comms_fmdemod_set_timer": [_vp, _vp],
    "comms_fft_set_timer": [_vp, _vp],
    "comms_chain_set_timer": [_vp, _vp],
    "comms_fir_create": [_vp, _sz, _vp, _sz, _i32, _pp],
    "comms_fir_set_algo": [_vp, _i32],
    "comms_fir_get_algo": [_vp, _sz, C.POINTER(_i32)],
    "comms_fir_get_kernel": [_vp, _sz, C.c_char_p, _sz],
    "comms_fir_run": [_vp, _vp, _sz, _vp],
    "comms_fir_run_dev": [_vp, _vp, _sz, _vp, _vp],
    "comms_fir_get_state": [_vp, _vp, _sz],
    "comms_fir_set_state": [_vp, _vp, _sz],
    "comms_fir_destroy": [_vp],
    "comms_fir_i16_create": [_vp, _sz, _vp, _sz, _i32, _pp],
    "comms_fir_i16_run": [_vp, _vp, _sz, _vp],
    "comms_fir_i16_run_dev": [_vp, _vp, _sz, _vp, _vp],
    "comms_fir_i16_get_state": [_vp, _vp, _sz],
    "comms_fir_i16_destroy": [_vp],
    "comms_fir_f64_create": [_vp, _sz, _vp, _sz, _i32, _pp],
    "comms_fir_f64_run": [_vp, _vp, _sz, _vp],
    "comms_fir_f64_run_dev": [_vp, _vp, _sz, _vp, _vp],
    "comms_fir_f64_get_state": [_vp, _vp, _sz],
    "comms_fir_f64_set_state": [_vp, _vp, _sz],
    "comms_fir_f64_destroy": [_vp],
    "comms_pulse_f64_create": [_vp, _sz, _sz, _i32, _pp],
    "comms_pulse_f64_run": [_vp, _vp, _sz, _vp],
    "comms_pulse_f64_run_dev": [_vp, _vp, _sz, _vp, _vp],
    "comms_pulse_f64_destroy": [_vp],
    "comms_pulse_i16_create": [_vp, _sz, _sz, _i32, _pp],
    "comms_pulse_i16_run": [_vp, _vp, _sz, _vp],
    "comms_pulse_i16_run_dev": [_vp, _vp, _sz, _vp, _vp],
    "comms_pulse_i16_destroy": [_vp],
    "comms_pulse_create": [_vp, _sz, _sz, _i32, _pp],
    "comms_pulse_run": [_vp, _vp, _sz, _vp],
    "comms_pulse_run_dev": [_vp, _vp, _sz, _vp, _vp],
    "comms_pulse_set_mixer": [_vp, _f64, _f64],
    "comms_pulse_get_phase": [_vp, C.POINTER(_f64)],
    "comms_pulse_set_output_format": [_vp, _i32, C.c_float],
    "comms_pulse_destroy": [_vp],
    "comms_mixer_create": [_f64, _f64, _i32, _pp],
    "comms_mixer_run": [_vp, _vp, _sz, _vp],
    "comms_mixer_run_dev": [_vp, _vp, _sz, _vp, _vp],
    "comms_mixer_run_f64": [_vp, _vp, _sz, _vp],
    "comms_mixer_run_f64_dev": [_vp, _vp, _sz, _vp, _vp],
    "comms_mixer_get_phase": [_vp, C.POINTER(_f64)],
    "comms_mixer_set_phase": [_vp, _f64],
    "comms_mixer_destroy": [_vp],
    "comms_decimate_out_len": [_sz, _sz, _psz],
    "comms_upsample_out_len": [_sz, _sz, _psz],
    "comms_decimate_run": [_vp, _sz, _sz, _sz, _vp, _psz, _i32],
    "comms_decimate_run_dev": [_vp, _sz, _sz, _sz, _vp, _psz, _i32, _vp],
    "comms_upsample_run": [_vp, _sz, _sz, _sz, _vp, _psz, _i32],
    "comms_upsample_run_dev": [_vp, _sz, _sz, _sz, _vp, _psz, _i32, _vp],
    "comms_fmdemod_create": [_i32, _pp],
    "comms_fmdemod_run": [_vp, _vp, _sz, _vp],
    "comms_fmdemod_run_dev": [_vp, _vp, _sz, _vp, _vp],
    "comms_fmdemod_get_prev": [_vp, _vp],
    "comms_fmdemod_set_prev": [_vp, _vp],
    "comms_fmdemod_destroy": [_vp],
    "comms_fft_create": [_sz, _i32, _i32, _pp],
    "comms_fft_run": [_vp, _vp, _sz, _vp],
    "comms_fft_run_dev": [_vp, _vp, _sz, _vp, _vp],
    "comms_fft_destroy": [_vp],
    "comms_fft_f64_create": [_sz, _i32, _i32, _pp],
    "comms_fft_f64_run": [_vp, _vp, _sz, _vp],
    "comms_fft_f64_run_dev": [_vp, _vp, _sz, _vp, _vp],
    "comms_fft_f64_destroy": [_vp],
    "comms_fmdemod_f64_create": [_i32, _pp],
    "comms_fmdemod_f64_run": [_vp, _vp, _sz, _vp],
    "comms_fmdemod_f64_run_dev": [_vp, _vp, _sz, _vp, _vp],
    "comms_fmdemod_f64_get_prev": [_vp, _vp],
    "comms_fmdemod_f64_set_prev": [_vp, _vp],
    "comms_fmdemod_f64_destroy": [_vp],
    "comms_rrc_taps": [_u32, _f64, _f64, _vp],
    "comms_rc_taps": [_u32, _f64, _f64, _vp],
    "comms_gaussian_taps": [_u32, _f64, _f64, _vp],
    "comms_rect_taps": [_sz, _vp],
    "comms_chain_create": [_f64, _f64, _vp, _sz, _sz, _i32, _i32, _pp],
    "comms_chain_create_ex": [_f64, _f64, _vp, _sz, _sz, _i32, _i32, _pp],
    "comms_chain_is_fused": [_vp, C.POINTER(_i32)],
    "comms_chain_run_dev": [_vp, _vp, _sz, _vp, _vp],
    "comms_chain_run": [_vp, _vp, _sz, _vp],
    "comms_chain_set_fir_state": [_vp, _vp, _sz],
    "comms_chain_get_fir_state": [_vp, _vp, _sz],
    "comms_chain_get_phase": [_vp, C.POINTER(_f64)],
    "comms_chain_set_phase": [_vp, _f64],
    "comms_chain_get_fm_prev": [_vp, _vp],
    "comms_chain_set_fm_prev": [_vp, _vp],
    "comms_chain_destroy": [_vp],
    "comms_iq_i16_to_c32": [_vp, _sz, C.c_float, _vp, _i32],
    "comms_iq_c32_to_i16": [_vp, _sz, C.c_float, _vp, _i32],
    "comms_iq_u8_to_c32": [_vp, _sz, _vp, _i32],
    "comms_iq_i16_to_c32_dev": [_vp, _sz, C.c_float, _vp, _i32, _vp],
    "comms_iq_c32_to_i16_dev": [_vp, _sz, C.c_float, _vp, _i32, _vp],
    "comms_iq_u8_to_c32_dev": [_vp, _sz, _vp, _i32, _vp],
    "comms_iq_real_to_c32_dev": [_vp, _sz, _vp, _i32, _vp],
    "comms_iq_c32_re_dev": [_vp, _sz, _vp, _i32, _vp],
    "comms_frequency_offset_estimate": [_vp, _sz, C.POINTER(_f64), _i32],
    "comms_psk_phase_estimate": [_vp, _sz, _u32, C.POINTER(_f64), _i32],
    "comms_qam_phase_estimate": [_vp, _sz, C.POINTER(_f64), _i32],
    "comms_frequency_offset_estimate_dev": [_vp, _sz, C.POINTER(_f64), _i32, _vp],
    "comms_psk_phase_estimate_dev": [_vp, _sz, _u32, C.POINTER(_f64), _i32, _vp],
    "comms_qam_phase_estimate_dev": [_vp, _sz, C.POINTER(_f64), _i32, _vp],
    "comms_timing_create": [_u32, _u32, _f64, _i32, _pp],
    "comms_timing_push": [_vp, _vp, _sz, C.POINTER(_f64)],
    "comms_timing_push_dev": [_vp, _vp, _sz, C.POINTER(_f64), _vp],
    "comms_timing_destroy": [_vp],
    "comms_qfilt_taps": [_u32, _f64, _u32, _vp],
    "comms_nco_create": [_f64, _f64, _i32, _pp],
    "comms_nco_run": [_vp, _vp, _sz, _vp],
    "comms_nco_run_dev": [_vp, _vp, _sz, _vp, _vp],
    "comms_nco_get_phase": [_vp, C.POINTER(_f64)],
    "comms_nco_destroy": [_vp],
    "comms_synth_iq_dev": [_vp, _sz, _u64, _u64, _i32, _vp],
    "comms_shard_range": [_sz, _u32, _u32, _psz, _psz],
    "comms_state_from_halo": [_vp, _sz, _vp],
    "comms_shard_mixer_phase": [_f64, _f64, C.c_int64, C.POINTER(_f64)],
    "comms_chain_prefix_len": [_sz, _sz, _i32, _psz],
}
_OTHER = {
    "comms_version": (C.c_char_p, []),
    "comms_last_error": (C.c_char_p, []),
    "comms_buf_ptr": (_vp, [_vp]),
    "comms_buf_size": (_sz, [_vp]),
    "comms_qfilt_len": (_sz, [_u32]),
    "comms_buf_device": (_i32, [_vp]),
    "comms_synth_iq_host": (None, [_vp, _sz, _u64, _u64]),
}


def all_symbols():
    return sorted(list(_PROTOS) + list(_OTHER))


def lib():
    """The loaded library (raises if it has not been built -- no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc, gfx950).  comms_rs_amd has no CPU fallback." % LIB_PATH)
        # PyTorch bundles its own HIP runtime.  When both live in one process the
        # runtime torch ships must be the one that is loaded first, or torch
        # cannot see the GPU afterwards -- so let torch load it if torch is here.
        if os.environ.get("COMMS_NO_TORCH_PRELOAD") != "1":
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        l = C.CDLL(LIB_PATH)
        missing = [n for n in all_symbols() if not hasattr(l, n)]
        if missing:
            raise ImportError("%s lacks symbols declared in include/comms_hip.h: %s"
                              % (LIB_PATH, ", ".join(missing)))
        for name, args in _PROTOS.items():
            f = getattr(l, name)
            f.restype = _i32
            f.argtypes = args
        for name, (res, args) in _OTHER.items():
            f = getattr(l, name)
            f.restype = res
            f.argtypes = args
        _lib = l
    return _lib


def check(status):
    if status != COMMS_OK:
        raise CommsError(status, lib().comms_last_error().decode("utf-8", "replace"))
