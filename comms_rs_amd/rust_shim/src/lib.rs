//! comms-rs nodes whose `run()` executes on an MI355X through libcomms_hip.so.
//!
//! UNTESTED SOURCE (no Rust toolchain where this was written).  Each struct keeps the reference
//! node's name, TYPE PARAMETER, constructor signature and public `input` / `output` fields, and
//! derives `Node` with the reference's own `node_derive` macro, so `connect_nodes!` /
//! `start_nodes!` / `Graph` work unchanged and a graph switches by its `use` lines alone:
//!
//! ```ignore
//! use comms_rs::prelude::*;
//! // instead of comms_rs::filter::fir_node::BatchFirNode, comms_rs::util::resample_node::DecimateNode, ...
//! use comms_rs_hip::{BatchFirNode, DecimateNode, FMDemodNode};
//! // examples/fm_radio.rs:146-152, unchanged:
//! let mut dec1: DecimateNode<Complex<f32>> = DecimateNode::new(5);
//! let mut filt1: BatchFirNode<f32> = BatchFirNode::new(taps.clone(), None);
//! let mut fm = FMDemodNode::new();
//! let mut dec2: DecimateNode<f32> = DecimateNode::new(5);
//! connect_nodes!(filt1, output, dec1, input);
//! ```
//!
//! The reference's nodes are generic over any `T: Num + Copy + Send`; the kernels exist for the
//! sample types below (sealed traits -- another `T` is a compile error here, not a run-time one):
//!
//! | node | reference | `T` here |
//! |---|---|---|
//! | `FirNode<T>`, `BatchFirNode<T>`, `PulseNode<T>` | fir_node.rs:45,148; pulse.rs:38 | `f32`, `f64`, `i16` (`FirSample`) |
//! | `MixerNode<T>` | mixer.rs:93 | `f32`, `f64` (`MixerSample`) |
//! | `FFTBatchNode<T>`, `FFTSampleNode<T>`, `FMDemodNode<T>` | fft_node.rs:28,104; analog_node.rs:20 | `f32`, `f64` (`FloatSample`) |
//! | `DecimateNode<T>`, `UpsampleNode<T>` | resample_node.rs:10,74 | any `Copy + Send` type of 1, 2, 4, 8 or 16 bytes |
pub mod ffi;

use comms_rs::prelude::*;
use ffi::*;
use num::{Complex, Num, Zero};
use std::marker::PhantomData;
use std::mem::size_of;
use std::os::raw::c_void;
use std::ptr;

fn to_err(st: comms_status_t) -> NodeError {
    // COMMS_ERR_ARG -> DataError, COMMS_ERR_DEVICE -> PermanentError (include/comms_hip.h)
    if st == COMMS_ERR_ARG { NodeError::DataError } else { NodeError::PermanentError }
}

mod sealed {
    pub trait Sealed {}
    impl Sealed for f32 {}
    impl Sealed for f64 {}
    impl Sealed for i16 {}
}

/// Sample types the FIR and pulse-shaping kernels are built for: `f32` (every BASELINE config), `f64` (the type of the
/// reference's batch_fir doc example and timing estimator; bit-identical outputs) and `i16` (the type of the reference's own
/// FIR / pulse goldens; wrapping arithmetic as in a release build).
pub trait FirSample: sealed::Sealed + Num + Copy + Send + 'static {
    #[doc(hidden)] type Fir;
    #[doc(hidden)] type Pulse;
    #[doc(hidden)] unsafe fn fir_create(taps: *const Complex<Self>, n_taps: usize, state: *const Complex<Self>, n_state: usize,
                                        out: *mut *mut Self::Fir) -> comms_status_t;
    #[doc(hidden)] unsafe fn fir_run(h: *mut Self::Fir, x: *const Complex<Self>, n: usize, y: *mut Complex<Self>) -> comms_status_t;
    #[doc(hidden)] unsafe fn fir_destroy(h: *mut Self::Fir);
    #[doc(hidden)] unsafe fn pulse_create(taps: *const Complex<Self>, n_taps: usize, sps: usize, out: *mut *mut Self::Pulse) -> comms_status_t;
    #[doc(hidden)] unsafe fn pulse_run(h: *mut Self::Pulse, sym: *const Complex<Self>, n: usize, y: *mut Complex<Self>) -> comms_status_t;
    #[doc(hidden)] unsafe fn pulse_destroy(h: *mut Self::Pulse);
}
impl FirSample for f32 {
    type Fir = comms_fir_t;
    type Pulse = comms_pulse_t;
    unsafe fn fir_create(t: *const Complex<f32>, n: usize, s: *const Complex<f32>, ns: usize, out: *mut *mut comms_fir_t) -> comms_status_t {
        comms_fir_create(t, n, s, ns, 0, out)
    }
    unsafe fn fir_run(h: *mut comms_fir_t, x: *const Complex<f32>, n: usize, y: *mut Complex<f32>) -> comms_status_t { comms_fir_run(h, x, n, y) }
    unsafe fn fir_destroy(h: *mut comms_fir_t) { comms_fir_destroy(h); }
    unsafe fn pulse_create(t: *const Complex<f32>, n: usize, sps: usize, out: *mut *mut comms_pulse_t) -> comms_status_t {
        comms_pulse_create(t, n, sps, 0, out)
    }
    unsafe fn pulse_run(h: *mut comms_pulse_t, s: *const Complex<f32>, n: usize, y: *mut Complex<f32>) -> comms_status_t { comms_pulse_run(h, s, n, y) }
    unsafe fn pulse_destroy(h: *mut comms_pulse_t) { comms_pulse_destroy(h); }
}
impl FirSample for i16 {
    type Fir = comms_fir_i16_t;
    type Pulse = comms_pulse_i16_t;
    unsafe fn fir_create(t: *const Complex<i16>, n: usize, s: *const Complex<i16>, ns: usize, out: *mut *mut comms_fir_i16_t) -> comms_status_t {
        comms_fir_i16_create(t, n, s, ns, 0, out)
    }
    unsafe fn fir_run(h: *mut comms_fir_i16_t, x: *const Complex<i16>, n: usize, y: *mut Complex<i16>) -> comms_status_t { comms_fir_i16_run(h, x, n, y) }
    unsafe fn fir_destroy(h: *mut comms_fir_i16_t) { comms_fir_i16_destroy(h); }
    unsafe fn pulse_create(t: *const Complex<i16>, n: usize, sps: usize, out: *mut *mut comms_pulse_i16_t) -> comms_status_t {
        comms_pulse_i16_create(t, n, sps, 0, out)
    }
    unsafe fn pulse_run(h: *mut comms_pulse_i16_t, s: *const Complex<i16>, n: usize, y: *mut Complex<i16>) -> comms_status_t { comms_pulse_i16_run(h, s, n, y) }
    unsafe fn pulse_destroy(h: *mut comms_pulse_i16_t) { comms_pulse_i16_destroy(h); }
}

impl FirSample for f64 {  // round 5: the reference's arithmetic operation for operation, bit-identical outputs (comms_fir_f64_*)
    type Fir = comms_fir_f64_t;
    type Pulse = comms_pulse_f64_t;
    unsafe fn fir_create(t: *const Complex<f64>, n: usize, s: *const Complex<f64>, ns: usize, out: *mut *mut comms_fir_f64_t) -> comms_status_t {
        comms_fir_f64_create(t, n, s, ns, 0, out)
    }
    unsafe fn fir_run(h: *mut comms_fir_f64_t, x: *const Complex<f64>, n: usize, y: *mut Complex<f64>) -> comms_status_t { comms_fir_f64_run(h, x, n, y) }
    unsafe fn fir_destroy(h: *mut comms_fir_f64_t) { comms_fir_f64_destroy(h); }
    unsafe fn pulse_create(t: *const Complex<f64>, n: usize, sps: usize, out: *mut *mut comms_pulse_f64_t) -> comms_status_t {
        comms_pulse_f64_create(t, n, sps, 0, out)
    }
    unsafe fn pulse_run(h: *mut comms_pulse_f64_t, s: *const Complex<f64>, n: usize, y: *mut Complex<f64>) -> comms_status_t { comms_pulse_f64_run(h, s, n, y) }
    unsafe fn pulse_destroy(h: *mut comms_pulse_f64_t) { comms_pulse_f64_destroy(h); }
}

/// Sample types of `MixerNode<T>`: `f32`, and `f64` -- the instantiation the reference's own mixer tests use
/// (mixer.rs:160-336), met at their 1e-6.
pub trait MixerSample: sealed::Sealed + Num + Copy + Send + 'static {
    #[doc(hidden)] unsafe fn mix(h: *mut comms_mixer_t, x: *const Complex<Self>, n: usize, y: *mut Complex<Self>) -> comms_status_t;
}
impl MixerSample for f32 {
    unsafe fn mix(h: *mut comms_mixer_t, x: *const Complex<f32>, n: usize, y: *mut Complex<f32>) -> comms_status_t { comms_mixer_run(h, x, n, y) }
}
impl MixerSample for f64 {
    unsafe fn mix(h: *mut comms_mixer_t, x: *const Complex<f64>, n: usize, y: *mut Complex<f64>) -> comms_status_t { comms_mixer_run_f64(h, x, n, y) }
}

/// Sample types of the FFT and FM-demodulation nodes: `f32` (the reference casts to f64 inside and back, fft/mod.rs:78-94; the
/// kernels compute in f32 within the north star's 1e-5) and `f64` (the instantiation of the reference's doc examples,
/// fft_node.rs:24,99: plain FP64 kernels, comms_fft_f64_* / comms_fmdemod_f64_*).
pub trait FloatSample: sealed::Sealed + Num + Copy + Send + Default + 'static {
    #[doc(hidden)] type Fft;
    #[doc(hidden)] type Fm;
    #[doc(hidden)] unsafe fn fft_create(fft_size: usize, ifft: i32, out: *mut *mut Self::Fft) -> comms_status_t;
    #[doc(hidden)] unsafe fn fft(h: *mut Self::Fft, x: *const Complex<Self>, n: usize, y: *mut Complex<Self>) -> comms_status_t;
    #[doc(hidden)] unsafe fn fft_destroy(h: *mut Self::Fft);
    #[doc(hidden)] unsafe fn fm_create(out: *mut *mut Self::Fm) -> comms_status_t;
    #[doc(hidden)] unsafe fn fm(h: *mut Self::Fm, x: *const Complex<Self>, n: usize, y: *mut Self) -> comms_status_t;
    #[doc(hidden)] unsafe fn fm_destroy(h: *mut Self::Fm);
}
impl FloatSample for f32 {
    type Fft = comms_fft_t;
    type Fm = comms_fmdemod_t;
    unsafe fn fft_create(fft_size: usize, ifft: i32, out: *mut *mut comms_fft_t) -> comms_status_t { comms_fft_create(fft_size, ifft, 0, out) }
    unsafe fn fft(h: *mut comms_fft_t, x: *const Complex<f32>, n: usize, y: *mut Complex<f32>) -> comms_status_t { comms_fft_run(h, x, n, y) }
    unsafe fn fft_destroy(h: *mut comms_fft_t) { comms_fft_destroy(h); }
    unsafe fn fm_create(out: *mut *mut comms_fmdemod_t) -> comms_status_t { comms_fmdemod_create(0, out) }
    unsafe fn fm(h: *mut comms_fmdemod_t, x: *const Complex<f32>, n: usize, y: *mut f32) -> comms_status_t { comms_fmdemod_run(h, x, n, y) }
    unsafe fn fm_destroy(h: *mut comms_fmdemod_t) { comms_fmdemod_destroy(h); }
}
impl FloatSample for f64 {
    type Fft = comms_fft_f64_t;
    type Fm = comms_fmdemod_f64_t;
    unsafe fn fft_create(fft_size: usize, ifft: i32, out: *mut *mut comms_fft_f64_t) -> comms_status_t { comms_fft_f64_create(fft_size, ifft, 0, out) }
    unsafe fn fft(h: *mut comms_fft_f64_t, x: *const Complex<f64>, n: usize, y: *mut Complex<f64>) -> comms_status_t { comms_fft_f64_run(h, x, n, y) }
    unsafe fn fft_destroy(h: *mut comms_fft_f64_t) { comms_fft_f64_destroy(h); }
    unsafe fn fm_create(out: *mut *mut comms_fmdemod_f64_t) -> comms_status_t { comms_fmdemod_f64_create(0, out) }
    unsafe fn fm(h: *mut comms_fmdemod_f64_t, x: *const Complex<f64>, n: usize, y: *mut f64) -> comms_status_t { comms_fmdemod_f64_run(h, x, n, y) }
    unsafe fn fm_destroy(h: *mut comms_fmdemod_f64_t) { comms_fmdemod_f64_destroy(h); }
}

fn czeros<T: Num + Copy>(n: usize) -> Vec<Complex<T>> { vec![Complex::new(T::zero(), T::zero()); n] }

/// `impl Node` for a per-sample node whose `run` costs a device launch: instead of the derive macro's
/// one-message `call` (node_derive/src/lib.rs:200-211) the node drains what is ALREADY queued behind
/// the first message (`recv`, then `try_recv` -- it never waits for more), runs the block in one launch
/// (`run_block`) and sends the outputs one by one in order.  Every receiver sees exactly the message
/// sequence the derived loop would produce; the C++ host runtime does the same (`DeriveNode::call`,
/// comms_rs_amd/host/comms/node.hpp) and is where this behaviour is tested: 16.5 Msamples/s through
/// MixerNode -> FirNode against ~40 ksamples/s with a launch per sample.
macro_rules! drained_node {
    ($name:ident, $bound:ident) => {
        impl<T: $bound> Node for $name<T> {
            fn start(&mut self) {
                for (send, val) in &self.output {
                    if let Some(v) = val { send.send(v.clone()).unwrap(); }
                }
                loop { if self.call().is_err() { break; } }
            }
            fn call(&mut self) -> Result<(), NodeError> {
                let block: Vec<Complex<T>> = match self.input {
                    Some(ref r) => {
                        let mut b = vec![r.recv().or(Err(NodeError::DataEnd))?];
                        while b.len() < (1 << 16) {
                            match r.try_recv() { Ok(v) => b.push(v), Err(_) => break }
                        }
                        b
                    }
                    None => return Err(NodeError::PermanentError),
                };
                let outs: Vec<Complex<T>> = self.run_block(&block)?;
                for res in outs {
                    for (send, _) in &self.output {
                        if send.send(res.clone()).is_err() { return Err(NodeError::CommError); }
                    }
                }
                Ok(())
            }
            fn is_connected(&self) -> bool { self.input.is_some() && !self.output.is_empty() }
        }
    };
}

/// fir_node.rs:148-221
#[derive(Node)]
#[pass_by_ref]
pub struct BatchFirNode<T>
where
    T: FirSample,
{
    pub input: NodeReceiver<Vec<Complex<T>>>,
    h: *mut T::Fir,
    pub output: NodeSender<Vec<Complex<T>>>,
}
// a handle is used by one thread at a time but not its creator: Send, never Sync
unsafe impl<T: FirSample> Send for BatchFirNode<T> {}
impl<T: FirSample> Drop for BatchFirNode<T> {
    fn drop(&mut self) { unsafe { T::fir_destroy(self.h) } }
}
impl<T> BatchFirNode<T>
where
    T: FirSample,
{
    pub fn new(taps: Vec<Complex<T>>, state: Option<Vec<Complex<T>>>) -> Self {
        let mut h = ptr::null_mut();
        let (sp, sn) = match &state { Some(s) => (s.as_ptr(), s.len()), None => (ptr::null(), 0) };
        let st = unsafe { T::fir_create(taps.as_ptr(), taps.len(), sp, sn, &mut h) };
        assert_eq!(st, COMMS_OK, "comms_fir_create failed");   // the reference panics on a bad state too
        BatchFirNode { input: Default::default(), h, output: Default::default() }
    }
    pub fn run(&mut self, input: &[Complex<T>]) -> Result<Vec<Complex<T>>, NodeError> {
        let mut out = czeros::<T>(input.len());
        let st = unsafe { T::fir_run(self.h, input.as_ptr(), input.len(), out.as_mut_ptr()) };
        if st == COMMS_OK { Ok(out) } else { Err(to_err(st)) }
    }
}

/// fir_node.rs:45-114 (one sample per message; queued samples run as one launch, see `drained_node!`)
pub struct FirNode<T>
where
    T: FirSample,
{
    pub input: NodeReceiver<Complex<T>>,
    h: *mut T::Fir,
    pub output: NodeSender<Complex<T>>,
}
unsafe impl<T: FirSample> Send for FirNode<T> {}
impl<T: FirSample> Drop for FirNode<T> {
    fn drop(&mut self) { unsafe { T::fir_destroy(self.h) } }
}
impl<T> FirNode<T>
where
    T: FirSample,
{
    pub fn new(taps: Vec<Complex<T>>, state: Option<Vec<Complex<T>>>) -> Self {
        let mut h = ptr::null_mut();
        let (sp, sn) = match &state { Some(s) => (s.as_ptr(), s.len()), None => (ptr::null(), 0) };
        let st = unsafe { T::fir_create(taps.as_ptr(), taps.len(), sp, sn, &mut h) };
        assert_eq!(st, COMMS_OK, "comms_fir_create failed");
        FirNode { input: Default::default(), h, output: Default::default() }
    }
    pub fn run(&mut self, input: &Complex<T>) -> Result<Complex<T>, NodeError> {
        let mut out = Complex::new(T::zero(), T::zero());
        let st = unsafe { T::fir_run(self.h, input, 1, &mut out) };
        if st == COMMS_OK { Ok(out) } else { Err(to_err(st)) }
    }
    pub fn run_block(&mut self, input: &[Complex<T>]) -> Result<Vec<Complex<T>>, NodeError> {
        let mut out = czeros::<T>(input.len());
        let st = unsafe { T::fir_run(self.h, input.as_ptr(), input.len(), out.as_mut_ptr()) };
        if st == COMMS_OK { Ok(out) } else { Err(to_err(st)) }
    }
}
drained_node!(FirNode, FirSample);

/// mixer.rs:93-148 -- argument order (dphase, phase) as in MixerNode::new (drained, see `drained_node!`)
pub struct MixerNode<T>
where
    T: MixerSample,
{
    pub input: NodeReceiver<Complex<T>>,
    h: *mut comms_mixer_t,
    pub output: NodeSender<Complex<T>>,
}
unsafe impl<T: MixerSample> Send for MixerNode<T> {}
impl<T: MixerSample> Drop for MixerNode<T> {
    fn drop(&mut self) { unsafe { comms_mixer_destroy(self.h); } }
}
impl<T> MixerNode<T>
where
    T: MixerSample,
{
    pub fn new(dphase: f64, phase: Option<f64>) -> Self {
        let mut h = ptr::null_mut();
        let st = unsafe { comms_mixer_create(dphase, phase.unwrap_or(0.0), 0, &mut h) };
        assert_eq!(st, COMMS_OK, "comms_mixer_create failed");
        MixerNode { input: Default::default(), h, output: Default::default() }
    }
    pub fn run(&mut self, input: &Complex<T>) -> Result<Complex<T>, NodeError> {
        let mut out = Complex::new(T::zero(), T::zero());
        let st = unsafe { T::mix(self.h, input, 1, &mut out) };
        if st == COMMS_OK { Ok(out) } else { Err(to_err(st)) }
    }
    pub fn run_block(&mut self, input: &[Complex<T>]) -> Result<Vec<Complex<T>>, NodeError> {
        let mut out = czeros::<T>(input.len());
        let st = unsafe { T::mix(self.h, input.as_ptr(), input.len(), out.as_mut_ptr()) };
        if st == COMMS_OK { Ok(out) } else { Err(to_err(st)) }
    }
    /// oscillator phase of the next sample: checkpoint hook / start phase of a stream shard
    pub fn phase(&self) -> f64 {
        let mut p = 0.0;
        let st = unsafe { comms_mixer_get_phase(self.h, &mut p) };
        assert_eq!(st, COMMS_OK, "comms_mixer_get_phase failed");
        p
    }
    pub fn set_phase(&mut self, phase: f64) {
        let st = unsafe { comms_mixer_set_phase(self.h, phase) };
        assert_eq!(st, COMMS_OK, "comms_mixer_set_phase failed");
    }
}
drained_node!(MixerNode, MixerSample);

/// pulse.rs:38-93
#[derive(Node)]
#[pass_by_ref]
pub struct PulseNode<T>
where
    T: FirSample,
{
    pub input: NodeReceiver<Complex<T>>,
    h: *mut T::Pulse,
    sam_per_sym: usize,
    pub output: NodeSender<Vec<Complex<T>>>,
}
unsafe impl<T: FirSample> Send for PulseNode<T> {}
impl<T: FirSample> Drop for PulseNode<T> {
    fn drop(&mut self) { unsafe { T::pulse_destroy(self.h) } }
}
impl<T> PulseNode<T>
where
    T: FirSample,
{
    pub fn new(taps: Vec<Complex<T>>, sam_per_sym: usize) -> Self {
        let mut h = ptr::null_mut();
        let st = unsafe { T::pulse_create(taps.as_ptr(), taps.len(), sam_per_sym, &mut h) };
        assert_eq!(st, COMMS_OK, "comms_pulse_create failed");
        PulseNode { input: Default::default(), h, sam_per_sym, output: Default::default() }
    }
    pub fn run(&mut self, input: &Complex<T>) -> Result<Vec<Complex<T>>, NodeError> {
        let mut out = czeros::<T>(self.sam_per_sym);
        let st = unsafe { T::pulse_run(self.h, input, 1, out.as_mut_ptr()) };
        if st == COMMS_OK { Ok(out) } else { Err(to_err(st)) }
    }
}
impl PulseNode<f32> {
    /// Transmit chain in one launch: fuses the `MixerNode::new(dphase, phase)` that follows (an addition; f32 only).
    pub fn with_mixer(self, dphase: f64, phase: Option<f64>) -> Self {
        let st = unsafe { comms_pulse_set_mixer(self.h, dphase, phase.unwrap_or(0.0)) };
        assert_eq!(st, COMMS_OK, "comms_pulse_set_mixer failed");
        self
    }
}

/// decimate / upsample of any element type by its size (include/comms_hip.h: elem = 1, 2, 4, 8 or 16 bytes)
fn resample<T: Copy>(up: bool, signal: &[T], rate: usize) -> Result<Vec<T>, NodeError> {
    let mut n_out = 0usize;
    let st = unsafe {
        if up { comms_upsample_out_len(signal.len(), rate, &mut n_out) } else { comms_decimate_out_len(signal.len(), rate, &mut n_out) }
    };
    if st != COMMS_OK { return Err(to_err(st)); }
    let mut out: Vec<T> = Vec::with_capacity(n_out);
    let (src, dst) = (signal.as_ptr() as *const c_void, out.as_mut_ptr() as *mut c_void);
    let st = unsafe {
        if up { comms_upsample_run(src, signal.len(), size_of::<T>(), rate, dst, &mut n_out, 0) }
        else { comms_decimate_run(src, signal.len(), size_of::<T>(), rate, dst, &mut n_out, 0) }
    };
    if st != COMMS_OK { return Err(to_err(st)); }
    unsafe { out.set_len(n_out) };   // every element was written by the library
    Ok(out)
}

/// resample_node.rs:10-66.  Any `T: Copy + Send` of 1, 2, 4, 8 or 16 bytes (`f32`, `Complex<f32>`, `Complex<f64>`,
/// `i16`, `u8` ...: the kernel copies elements by size, bit for bit); another size is `NodeError::DataError`.
#[derive(Node)]
#[pass_by_ref]
pub struct DecimateNode<T>
where
    T: Copy + Send,
{
    pub input: NodeReceiver<Vec<T>>,
    dec_rate: usize,
    pub output: NodeSender<Vec<T>>,
}
impl<T> DecimateNode<T>
where
    T: Copy + Send,
{
    pub fn new(dec_rate: usize) -> Self {
        DecimateNode { dec_rate, input: Default::default(), output: Default::default() }
    }
    pub fn run(&mut self, signal: &[T]) -> Result<Vec<T>, NodeError> {
        resample(false, signal, self.dec_rate)
    }
    /// resample_node.rs:53-65 (the reference exposes the function itself too, on `&self`)
    pub fn decimate(&self, data: &[T]) -> Vec<T> {
        resample(false, data, self.dec_rate).expect("comms_decimate_run failed")
    }
}

/// resample_node.rs:74-132.  `T::zero()` must be the all-zero bit pattern (true of every primitive number and of
/// `Complex` of them): the kernel fills the gaps with zero bytes.
#[derive(Node)]
#[pass_by_ref]
pub struct UpsampleNode<T>
where
    T: Copy + Send + Zero,
{
    pub input: NodeReceiver<Vec<T>>,
    ups_rate: usize,
    pub output: NodeSender<Vec<T>>,
}
impl<T> UpsampleNode<T>
where
    T: Copy + Send + Zero,
{
    pub fn new(ups_rate: usize) -> Self {
        UpsampleNode { input: Default::default(), ups_rate, output: Default::default() }
    }
    pub fn run(&mut self, signal: &[T]) -> Result<Vec<T>, NodeError> {
        resample(true, signal, self.ups_rate)
    }
    /// resample_node.rs:120-131
    pub fn upsample(&self, data: &[T]) -> Vec<T> {
        resample(true, data, self.ups_rate).expect("comms_upsample_run failed")
    }
}

/// analog_node.rs:20-52
#[derive(Node)]
#[pass_by_ref]
pub struct FMDemodNode<T>
where
    T: FloatSample,
{
    pub input: NodeReceiver<Vec<Complex<T>>>,
    h: *mut T::Fm,
    pub output: NodeSender<Vec<T>>,
}
unsafe impl<T: FloatSample> Send for FMDemodNode<T> {}
impl<T: FloatSample> Drop for FMDemodNode<T> {
    fn drop(&mut self) { unsafe { T::fm_destroy(self.h); } }
}
impl<T> FMDemodNode<T>
where
    T: FloatSample,
{
    pub fn new() -> Self {
        let mut h = ptr::null_mut();
        let st = unsafe { T::fm_create(&mut h) };
        assert_eq!(st, COMMS_OK, "comms_fmdemod_create failed");
        FMDemodNode { input: Default::default(), h, output: Default::default() }
    }
    pub fn run(&mut self, samples: &[Complex<T>]) -> Result<Vec<T>, NodeError> {
        let mut out = vec![T::zero(); samples.len()];
        let st = unsafe { T::fm(self.h, samples.as_ptr(), samples.len(), out.as_mut_ptr()) };
        if st == COMMS_OK { Ok(out) } else { Err(to_err(st)) }
    }
}
impl<T: FloatSample> Default for FMDemodNode<T> {
    fn default() -> Self { Self::new() }   // the reference derives Default (analog_node.rs:18)
}

/// fft_node.rs:28-84
#[derive(Node)]
#[pass_by_ref]
pub struct FFTBatchNode<T>
where
    T: FloatSample,
{
    pub input: NodeReceiver<Vec<Complex<T>>>,
    h: *mut T::Fft,
    pub output: NodeSender<Vec<Complex<T>>>,
}
unsafe impl<T: FloatSample> Send for FFTBatchNode<T> {}
impl<T: FloatSample> Drop for FFTBatchNode<T> {
    fn drop(&mut self) { unsafe { T::fft_destroy(self.h); } }
}
impl<T> FFTBatchNode<T>
where
    T: FloatSample,
{
    pub fn new(fft_size: usize, ifft: bool) -> Self {
        let mut h = ptr::null_mut();
        let st = unsafe { T::fft_create(fft_size, ifft as i32, &mut h) };
        assert_eq!(st, COMMS_OK, "comms_fft_create failed");
        FFTBatchNode { input: Default::default(), h, output: Default::default() }
    }
    pub fn run(&mut self, data: &[Complex<T>]) -> Result<Vec<Complex<T>>, NodeError> {
        let mut out = czeros::<T>(data.len());
        let st = unsafe { T::fft(self.h, data.as_ptr(), data.len(), out.as_mut_ptr()) };
        if st == COMMS_OK { Ok(out) } else { Err(to_err(st)) }   // rustfft would panic on a wrong length
    }
}

/// fft_node.rs:104-168: `#[aggregate]` -- a sample per message in, a spectrum out every `fft_size` samples
/// (`run` returns `Ok(None)` in between and the derived `call` sends nothing, node_derive/src/lib.rs:139-151).
#[derive(Node)]
#[aggregate]
#[pass_by_ref]
pub struct FFTSampleNode<T>
where
    T: FloatSample,
{
    pub input: NodeReceiver<Complex<T>>,
    h: *mut T::Fft,
    fft_size: usize,
    samples: Vec<Complex<T>>,
    pub output: NodeSender<Vec<Complex<T>>>,
}
unsafe impl<T: FloatSample> Send for FFTSampleNode<T> {}
impl<T: FloatSample> Drop for FFTSampleNode<T> {
    fn drop(&mut self) { unsafe { T::fft_destroy(self.h); } }
}
impl<T> FFTSampleNode<T>
where
    T: FloatSample,
{
    pub fn new(fft_size: usize, ifft: bool) -> Self {
        let mut h = ptr::null_mut();
        let st = unsafe { T::fft_create(fft_size, ifft as i32, &mut h) };
        assert_eq!(st, COMMS_OK, "comms_fft_create failed");
        FFTSampleNode { input: Default::default(), h, fft_size, samples: Vec::with_capacity(fft_size), output: Default::default() }
    }
    pub fn run(&mut self, sample: &Complex<T>) -> Result<Option<Vec<Complex<T>>>, NodeError> {
        self.samples.push(*sample);
        if self.samples.len() != self.fft_size { return Ok(None); }
        let mut out = czeros::<T>(self.fft_size);
        let st = unsafe { T::fft(self.h, self.samples.as_ptr(), self.fft_size, out.as_mut_ptr()) };
        self.samples.clear();
        if st == COMMS_OK { Ok(Some(out)) } else { Err(to_err(st)) }
    }
}

// ------------------------------------------------------------------ additions (no counterpart in the reference)

macro_rules! handle_node {
    ($name:ident, $destroy:ident) => {
        unsafe impl Send for $name {}
        impl Drop for $name {
            fn drop(&mut self) { unsafe { $destroy(self.h); } }
        }
    };
}

/// An additional node: MixerNode -> BatchFirNode -> DecimateNode [-> FMDemodNode] (or
/// BatchFirNode -> MixerNode -> DecimateNode with `mixer_after_fir`) as one launch
/// (comms_chain_*).  `ChainNode` emits Complex<f32>, `FmChainNode` the demodulated f32.
#[derive(Node)]
#[pass_by_ref]
pub struct ChainNode {
    pub input: NodeReceiver<Vec<Complex<f32>>>,
    h: *mut comms_chain_t,
    rate: usize,
    pub output: NodeSender<Vec<Complex<f32>>>,
}
handle_node!(ChainNode, comms_chain_destroy);
impl ChainNode {
    pub fn new(dphase: f64, phase: Option<f64>, taps: Vec<Complex<f32>>, rate: usize, mixer_after_fir: bool) -> Self {
        let mut h = ptr::null_mut();
        let flags = if mixer_after_fir { COMMS_CHAIN_MIXER_AFTER_FIR } else { 0 };
        let st = unsafe { comms_chain_create_ex(dphase, phase.unwrap_or(0.0), taps.as_ptr(), taps.len(), rate, flags, 0, &mut h) };
        assert_eq!(st, COMMS_OK, "comms_chain_create_ex failed");
        ChainNode { input: Default::default(), h, rate, output: Default::default() }
    }
    pub fn run(&mut self, input: &[Complex<f32>]) -> Result<Vec<Complex<f32>>, NodeError> {
        if self.rate == 0 || input.len() % self.rate != 0 { return Err(NodeError::DataError); }
        let mut out = vec![Complex::new(0.0f32, 0.0); input.len() / self.rate];
        let st = unsafe { comms_chain_run(self.h, input.as_ptr(), input.len(), out.as_mut_ptr() as *mut _) };
        if st == COMMS_OK { Ok(out) } else { Err(to_err(st)) }
    }
}

#[derive(Node)]
#[pass_by_ref]
pub struct FmChainNode {
    pub input: NodeReceiver<Vec<Complex<f32>>>,
    h: *mut comms_chain_t,
    rate: usize,
    pub output: NodeSender<Vec<f32>>,
}
handle_node!(FmChainNode, comms_chain_destroy);
impl FmChainNode {
    pub fn new(dphase: f64, phase: Option<f64>, taps: Vec<Complex<f32>>, rate: usize) -> Self {
        let mut h = ptr::null_mut();
        let st = unsafe {
            comms_chain_create_ex(dphase, phase.unwrap_or(0.0), taps.as_ptr(), taps.len(), rate, COMMS_CHAIN_FM_DEMOD, 0, &mut h)
        };
        assert_eq!(st, COMMS_OK, "comms_chain_create_ex failed");
        FmChainNode { input: Default::default(), h, rate, output: Default::default() }
    }
    pub fn run(&mut self, input: &[Complex<f32>]) -> Result<Vec<f32>, NodeError> {
        if self.rate == 0 || input.len() % self.rate != 0 { return Err(NodeError::DataError); }
        let mut out = vec![0.0f32; input.len() / self.rate];
        let st = unsafe { comms_chain_run(self.h, input.as_ptr(), input.len(), out.as_mut_ptr() as *mut _) };
        if st == COMMS_OK { Ok(out) } else { Err(to_err(st)) }
    }
    /// The chain's whole cross-call state (FIR history newest first, oscillator phase, FM.prev): what a
    /// checkpoint stores and what the node of the next stream shard starts from (analog.rs:9,31; fir_node.rs:193-211).
    pub fn state(&mut self, n_taps: usize) -> Result<(Vec<Complex<f32>>, f64, Complex<f32>), NodeError> {
        let mut hist = vec![Complex::new(0.0f32, 0.0); n_taps];
        let (mut phase, mut prev) = (0.0f64, Complex::new(0.0f32, 0.0));
        for st in [unsafe { comms_chain_get_fir_state(self.h, hist.as_mut_ptr(), n_taps) },
                   unsafe { comms_chain_get_phase(self.h, &mut phase) },
                   unsafe { comms_chain_get_fm_prev(self.h, &mut prev) }].iter() {
            if *st != COMMS_OK { return Err(to_err(*st)); }
        }
        Ok((hist, phase, prev))
    }
    pub fn set_state(&mut self, hist: &[Complex<f32>], phase: f64, prev: Complex<f32>) -> Result<(), NodeError> {
        for st in [unsafe { comms_chain_set_fir_state(self.h, hist.as_ptr(), hist.len()) },
                   unsafe { comms_chain_set_phase(self.h, phase) },
                   unsafe { comms_chain_set_fm_prev(self.h, &prev) }].iter() {
            if *st != COMMS_OK { return Err(to_err(*st)); }
        }
        Ok(())
    }
}

/// The literal front end of examples/fm_radio.rs:82-90,144-152: RTL-SDR bytes in, demodulated audio-rate
/// f32 out, one launch per block -- the u8 -> f32 conversion happens in the kernel's load stage (2 B/sample
/// from HBM).  `input` carries the raw interleaved (re, im) bytes exactly as the radio delivers them.
#[derive(Node)]
#[pass_by_ref]
pub struct RtlFmChainNode {
    pub input: NodeReceiver<Vec<u8>>,
    h: *mut comms_chain_t,
    rate: usize,
    pub output: NodeSender<Vec<f32>>,
}
handle_node!(RtlFmChainNode, comms_chain_destroy);
impl RtlFmChainNode {
    pub fn new(taps: Vec<Complex<f32>>, rate: usize) -> Self {
        let mut h = ptr::null_mut();
        let st = unsafe { comms_chain_create_ex(0.0, 0.0, taps.as_ptr(), taps.len(), rate, COMMS_CHAIN_FM_DEMOD, 0, &mut h) };
        assert_eq!(st, COMMS_OK, "comms_chain_create_ex failed");
        assert_eq!(unsafe { comms_chain_set_input_format(h, COMMS_IQ_U8, 1.0) }, COMMS_OK);
        RtlFmChainNode { input: Default::default(), h, rate, output: Default::default() }
    }
    pub fn run(&mut self, bytes: &[u8]) -> Result<Vec<f32>, NodeError> {
        let n = bytes.len() / 2;
        if self.rate == 0 || n % self.rate != 0 { return Err(NodeError::DataError); }
        let mut out = vec![0.0f32; n / self.rate];
        let st = unsafe { comms_chain_run(self.h, bytes.as_ptr() as *const _, n, out.as_mut_ptr() as *mut _) };
        if st == COMMS_OK { Ok(out) } else { Err(to_err(st)) }
    }
}

/// demodulation/timing_estimator.rs:116-136
#[derive(Node)]
#[pass_by_ref]
pub struct TimingEstimatorNode {
    pub input: NodeReceiver<Vec<Complex<f64>>>,
    h: *mut comms_timing_t,
    pub output: NodeSender<f64>,
}
handle_node!(TimingEstimatorNode, comms_timing_destroy);
impl TimingEstimatorNode {
    /// `Err(())` stands for the reference's `MathError::InvalidRolloffError`.
    pub fn new(n: u32, d: u32, alpha: f64) -> Result<Self, ()> {
        let mut h = ptr::null_mut();
        let st = unsafe { comms_timing_create(n, d, alpha, 0, &mut h) };
        if st != COMMS_OK { return Err(()); }
        Ok(TimingEstimatorNode { input: Default::default(), h, output: Default::default() })
    }
    pub fn run(&mut self, input: &[Complex<f64>]) -> Result<f64, NodeError> {
        let mut est = 0.0f64;
        let st = unsafe { comms_timing_push(self.h, input.as_ptr() as *const f64, input.len(), &mut est) };
        if st == COMMS_OK { Ok(est) } else { Err(to_err(st)) }
    }
}

/// demodulation/nco.rs:118-133 in block form: a vector of phase errors per message.
#[derive(Node)]
#[pass_by_ref]
pub struct BatchNcoNode {
    pub input: NodeReceiver<Vec<f64>>,
    h: *mut comms_nco_t,
    pub output: NodeSender<Vec<Complex<f64>>>,
}
handle_node!(BatchNcoNode, comms_nco_destroy);
impl BatchNcoNode {
    pub fn new(dphase: f64, phase: Option<f64>) -> Self {
        let mut h = ptr::null_mut();
        let st = unsafe { comms_nco_create(dphase, phase.unwrap_or(0.0), 0, &mut h) };
        assert_eq!(st, COMMS_OK, "comms_nco_create failed");
        BatchNcoNode { input: Default::default(), h, output: Default::default() }
    }
    pub fn run(&mut self, perr: &[f64]) -> Result<Vec<Complex<f64>>, NodeError> {
        let mut out = vec![Complex::new(0.0f64, 0.0); perr.len()];
        let st = unsafe { comms_nco_run(self.h, perr.as_ptr(), perr.len(), out.as_mut_ptr() as *mut f64) };
        if st == COMMS_OK { Ok(out) } else { Err(to_err(st)) }
    }
}

/// demodulation/frequency_estimator.rs:27-42, phase_estimator.rs:26-33, :58-65 (free functions there too;
/// they cannot fail there, so a device failure panics here)
pub fn frequency_offset_estimate(samples: &[Complex<f64>]) -> f64 {
    let mut out = 0.0f64;
    let st = unsafe { comms_frequency_offset_estimate(samples.as_ptr() as *const f64, samples.len(), &mut out, 0) };
    assert_eq!(st, COMMS_OK, "comms_frequency_offset_estimate failed");
    out
}
pub fn psk_phase_estimate(symbols: &[Complex<f64>], m: u32) -> f64 {
    let mut out = 0.0f64;
    let st = unsafe { comms_psk_phase_estimate(symbols.as_ptr() as *const f64, symbols.len(), m, &mut out, 0) };
    assert_eq!(st, COMMS_OK, "comms_psk_phase_estimate failed");
    out
}
pub fn qam_phase_estimate(symbols: &[Complex<f64>]) -> f64 {
    let mut out = 0.0f64;
    let st = unsafe { comms_qam_phase_estimate(symbols.as_ptr() as *const f64, symbols.len(), &mut out, 0) };
    assert_eq!(st, COMMS_OK, "comms_qam_phase_estimate failed");
    out
}

/// Device-resident message: `Clone` bumps the library's refcount (the derive macro clones once per sender,
/// node_derive/src/lib.rs:156).  `T` is the element type the bytes are read as.
pub struct DeviceBuf<T> { b: *mut comms_buf_t, pub len: usize, _t: PhantomData<T> }
unsafe impl<T: Send> Send for DeviceBuf<T> {}
impl<T> Clone for DeviceBuf<T> {
    fn clone(&self) -> Self { unsafe { comms_buf_retain(self.b) }; DeviceBuf { b: self.b, len: self.len, _t: PhantomData } }
}
impl<T> Drop for DeviceBuf<T> {
    fn drop(&mut self) { unsafe { comms_buf_release(self.b) }; }
}

/// One stream over several GPUs (host arithmetic of `include/comms_hip.h`, "stream shards"):
/// contiguous shards, and per neighbour pair one hand-over of the raw samples in front of the shard.  The
/// library moves no samples between GPUs itself: the host passes the halo with its own transport (RCCL
/// send / recv, MPI, a channel between node threads) and calls these three around it.
pub mod shard {
    use super::*;
    /// `[start, stop)` of `rank`'s share of `total` units (samples, or FFT transforms)
    pub fn range(total: usize, world: u32, rank: u32) -> (usize, usize) {
        let (mut a, mut b) = (0usize, 0usize);
        let st = unsafe { comms_shard_range(total, world, rank, &mut a, &mut b) };
        assert_eq!(st, COMMS_OK, "comms_shard_range failed");
        (a, b)
    }
    /// the samples before a shard (time order, as many as taps) -> `BatchFirNode::new(taps, Some(state))`
    pub fn state_from_halo(halo: &[Complex<f32>]) -> Vec<Complex<f32>> {
        let mut state = vec![Complex::new(0.0f32, 0.0); halo.len()];
        let st = unsafe { comms_state_from_halo(halo.as_ptr(), halo.len(), state.as_mut_ptr()) };
        assert_eq!(st, COMMS_OK, "comms_state_from_halo failed");
        state
    }
    /// oscillator phase of stream sample `first_index` -> `MixerNode::new(dphase, Some(phase))` of the shard's node
    pub fn mixer_phase(phase0: f64, dphase: f64, first_index: i64) -> f64 {
        let mut ph = 0.0f64;
        let st = unsafe { comms_shard_mixer_phase(phase0, dphase, first_index, &mut ph) };
        assert_eq!(st, COMMS_OK, "comms_shard_mixer_phase failed");
        ph
    }
    /// raw samples a fused chain shard runs through first (outputs dropped): FIR history + the sample FM.prev comes from
    pub fn chain_prefix_len(n_taps: usize, rate: usize, fm_demod: bool) -> usize {
        let mut n = 0usize;
        let st = unsafe { comms_chain_prefix_len(n_taps, rate, fm_demod as i32, &mut n) };
        assert_eq!(st, COMMS_OK, "comms_chain_prefix_len failed");
        n
    }
}
