//! `extern "C"` declarations of include/comms_hip.h (the hot-path subset).
//! UNTESTED SOURCE: written against the header, never compiled (no rustc in the
//! build environment).
#![allow(non_camel_case_types)]
use num::Complex;
use std::os::raw::{c_char, c_void};

pub type comms_status_t = i32;
pub const COMMS_OK: comms_status_t = 0;
pub const COMMS_ERR_ARG: comms_status_t = 1;
pub const COMMS_ERR_DEVICE: comms_status_t = 2;

/// `comms_c32` == `num::Complex<f32>` (`#[repr(C)]`, interleaved re, im).
pub type comms_c32 = Complex<f32>;

#[repr(C)] pub struct comms_fir_t { _p: [u8; 0] }
#[repr(C)] pub struct comms_pulse_t { _p: [u8; 0] }
#[repr(C)] pub struct comms_mixer_t { _p: [u8; 0] }
#[repr(C)] pub struct comms_fmdemod_t { _p: [u8; 0] }
#[repr(C)] pub struct comms_fft_t { _p: [u8; 0] }
#[repr(C)] pub struct comms_buf_t { _p: [u8; 0] }

extern "C" {
    pub fn comms_last_error() -> *const c_char;

    pub fn comms_fir_create(taps: *const comms_c32, n_taps: usize, state: *const comms_c32, n_state: usize,
                            device: i32, out: *mut *mut comms_fir_t) -> comms_status_t;
    pub fn comms_fir_run(h: *mut comms_fir_t, input: *const comms_c32, n: usize, out: *mut comms_c32) -> comms_status_t;
    pub fn comms_fir_run_dev(h: *mut comms_fir_t, d_in: *const comms_c32, n: usize, d_out: *mut comms_c32,
                             stream: *mut c_void) -> comms_status_t;
    pub fn comms_fir_destroy(h: *mut comms_fir_t) -> comms_status_t;

    pub fn comms_pulse_create(taps: *const comms_c32, n_taps: usize, sam_per_sym: usize, device: i32,
                              out: *mut *mut comms_pulse_t) -> comms_status_t;
    pub fn comms_pulse_run(h: *mut comms_pulse_t, sym: *const comms_c32, n_sym: usize, out: *mut comms_c32) -> comms_status_t;
    pub fn comms_pulse_destroy(h: *mut comms_pulse_t) -> comms_status_t;

    pub fn comms_mixer_create(dphase: f64, phase: f64, device: i32, out: *mut *mut comms_mixer_t) -> comms_status_t;
    pub fn comms_mixer_run(h: *mut comms_mixer_t, input: *const comms_c32, n: usize, out: *mut comms_c32) -> comms_status_t;
    pub fn comms_mixer_destroy(h: *mut comms_mixer_t) -> comms_status_t;

    pub fn comms_decimate_out_len(n: usize, rate: usize, out_n: *mut usize) -> comms_status_t;
    pub fn comms_upsample_out_len(n: usize, rate: usize, out_n: *mut usize) -> comms_status_t;
    pub fn comms_decimate_run(input: *const c_void, n: usize, elem: usize, rate: usize, out: *mut c_void,
                              out_n: *mut usize, device: i32) -> comms_status_t;
    pub fn comms_upsample_run(input: *const c_void, n: usize, elem: usize, rate: usize, out: *mut c_void,
                              out_n: *mut usize, device: i32) -> comms_status_t;

    pub fn comms_fmdemod_create(device: i32, out: *mut *mut comms_fmdemod_t) -> comms_status_t;
    pub fn comms_fmdemod_run(h: *mut comms_fmdemod_t, input: *const comms_c32, n: usize, out: *mut f32) -> comms_status_t;
    pub fn comms_fmdemod_destroy(h: *mut comms_fmdemod_t) -> comms_status_t;

    pub fn comms_fft_create(fft_size: usize, inverse: i32, device: i32, out: *mut *mut comms_fft_t) -> comms_status_t;
    pub fn comms_fft_run(h: *mut comms_fft_t, input: *const comms_c32, n: usize, out: *mut comms_c32) -> comms_status_t;
    pub fn comms_fft_destroy(h: *mut comms_fft_t) -> comms_status_t;

    pub fn comms_rrc_taps(n_taps: u32, sam_per_sym: f64, beta: f64, out: *mut comms_c32) -> comms_status_t;

    pub fn comms_buf_alloc(bytes: usize, device: i32, out: *mut *mut comms_buf_t) -> comms_status_t;
    pub fn comms_buf_retain(b: *mut comms_buf_t) -> comms_status_t;
    pub fn comms_buf_release(b: *mut comms_buf_t) -> comms_status_t;
    pub fn comms_buf_ptr(b: *const comms_buf_t) -> *mut c_void;
}
