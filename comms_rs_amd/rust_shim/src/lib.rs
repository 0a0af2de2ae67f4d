//! comms-rs nodes whose `run()` executes on an MI355X through libcomms_hip.so.
//!
//! UNTESTED SOURCE (no Rust toolchain where this was written).  Each struct keeps
//! the reference node's name, constructor signature and public `input` / `output`
//! fields, and derives `Node` with the reference's own `node_derive` macro, so
//! `connect_nodes!` / `start_nodes!` / `Graph` work unchanged:
//!
//! ```ignore
//! use comms_rs::prelude::*;
//! use comms_rs_hip::BatchFirNode;          // instead of comms_rs::filter::fir_node::BatchFirNode
//! let mut filt: BatchFirNode = BatchFirNode::new(taps, None);
//! connect_nodes!(src, output, filt, input);
//! start_nodes!(src, filt);
//! ```
pub mod ffi;

use comms_rs::prelude::*;
use ffi::*;
use num::Complex;
use std::ptr;

fn to_err(st: comms_status_t) -> NodeError {
    // COMMS_ERR_ARG -> DataError, COMMS_ERR_DEVICE -> PermanentError (include/comms_hip.h)
    if st == COMMS_ERR_ARG { NodeError::DataError } else { NodeError::PermanentError }
}

macro_rules! handle_node {
    ($name:ident, $h:ty, $destroy:ident) => {
        // a handle is used by one thread at a time but not its creator: Send, never Sync
        unsafe impl Send for $name {}
        impl Drop for $name {
            fn drop(&mut self) { unsafe { $destroy(self.h); } }
        }
    };
}

/// `impl Node` for a per-sample node whose `run` costs a device launch: instead of the derive macro's
/// one-message `call` (node_derive/src/lib.rs:200-211) the node drains what is ALREADY queued behind
/// the first message (`recv`, then `try_recv` -- it never waits for more), runs the block in one launch
/// (`run_block`) and sends the outputs one by one in order.  Every receiver sees exactly the message
/// sequence the derived loop would produce; the C++ host runtime does the same (`DeriveNode::call`,
/// comms_rs_amd/host/comms/node.hpp) and is where this behaviour is tested: 16.5 Msamples/s through
/// MixerNode -> FirNode against ~40 ksamples/s with a launch per sample.
macro_rules! drained_node {
    ($name:ident, $in:ty, $out:ty) => {
        impl Node for $name {
            fn start(&mut self) {
                for (send, val) in &self.output {
                    if let Some(v) = val { send.send(v.clone()).unwrap(); }
                }
                loop { if self.call().is_err() { break; } }
            }
            fn call(&mut self) -> Result<(), NodeError> {
                let block: Vec<$in> = match self.input {
                    Some(ref r) => {
                        let mut b = vec![r.recv().or(Err(NodeError::DataEnd))?];
                        while b.len() < (1 << 16) {
                            match r.try_recv() { Ok(v) => b.push(v), Err(_) => break }
                        }
                        b
                    }
                    None => return Err(NodeError::PermanentError),
                };
                let outs: Vec<$out> = self.run_block(&block)?;
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
pub struct BatchFirNode {
    pub input: NodeReceiver<Vec<Complex<f32>>>,
    h: *mut comms_fir_t,
    pub output: NodeSender<Vec<Complex<f32>>>,
}
handle_node!(BatchFirNode, comms_fir_t, comms_fir_destroy);
impl BatchFirNode {
    pub fn new(taps: Vec<Complex<f32>>, state: Option<Vec<Complex<f32>>>) -> Self {
        let mut h = ptr::null_mut();
        let (sp, sn) = match &state { Some(s) => (s.as_ptr(), s.len()), None => (ptr::null(), 0) };
        let st = unsafe { comms_fir_create(taps.as_ptr(), taps.len(), sp, sn, 0, &mut h) };
        assert_eq!(st, COMMS_OK, "comms_fir_create failed");   // the reference panics on a bad state too
        BatchFirNode { input: Default::default(), h, output: Default::default() }
    }
    pub fn run(&mut self, input: &[Complex<f32>]) -> Result<Vec<Complex<f32>>, NodeError> {
        let mut out = vec![Complex::new(0.0f32, 0.0); input.len()];
        let st = unsafe { comms_fir_run(self.h, input.as_ptr(), input.len(), out.as_mut_ptr()) };
        if st == COMMS_OK { Ok(out) } else { Err(to_err(st)) }
    }
}

/// BatchFirNode<i16>: the reference's node instantiated on the sample type of its own tests (fir_node.rs:342-449);
/// wrapping i16 arithmetic as in a release build.
#[derive(Node)]
#[pass_by_ref]
pub struct BatchFirNodeI16 {
    pub input: NodeReceiver<Vec<Complex<i16>>>,
    h: *mut comms_fir_i16_t,
    pub output: NodeSender<Vec<Complex<i16>>>,
}
handle_node!(BatchFirNodeI16, comms_fir_i16_t, comms_fir_i16_destroy);
impl BatchFirNodeI16 {
    pub fn new(taps: Vec<Complex<i16>>, state: Option<Vec<Complex<i16>>>) -> Self {
        let mut h = ptr::null_mut();
        let (sp, sn) = match &state { Some(s) => (s.as_ptr(), s.len()), None => (ptr::null(), 0) };
        let st = unsafe { comms_fir_i16_create(taps.as_ptr(), taps.len(), sp, sn, 0, &mut h) };
        assert_eq!(st, COMMS_OK, "comms_fir_i16_create failed");
        BatchFirNodeI16 { input: Default::default(), h, output: Default::default() }
    }
    pub fn run(&mut self, input: &[Complex<i16>]) -> Result<Vec<Complex<i16>>, NodeError> {
        let mut out = vec![Complex::new(0i16, 0); input.len()];
        let st = unsafe { comms_fir_i16_run(self.h, input.as_ptr(), input.len(), out.as_mut_ptr()) };
        if st == COMMS_OK { Ok(out) } else { Err(to_err(st)) }
    }
}

/// fir_node.rs:45-114 (one sample per message; queued samples run as one launch, see `drained_node!`)
pub struct FirNode {
    pub input: NodeReceiver<Complex<f32>>,
    h: *mut comms_fir_t,
    pub output: NodeSender<Complex<f32>>,
}
handle_node!(FirNode, comms_fir_t, comms_fir_destroy);
impl FirNode {
    pub fn new(taps: Vec<Complex<f32>>, state: Option<Vec<Complex<f32>>>) -> Self {
        let mut h = ptr::null_mut();
        let (sp, sn) = match &state { Some(s) => (s.as_ptr(), s.len()), None => (ptr::null(), 0) };
        let st = unsafe { comms_fir_create(taps.as_ptr(), taps.len(), sp, sn, 0, &mut h) };
        assert_eq!(st, COMMS_OK, "comms_fir_create failed");
        FirNode { input: Default::default(), h, output: Default::default() }
    }
    pub fn run(&mut self, input: &Complex<f32>) -> Result<Complex<f32>, NodeError> {
        let mut out = Complex::new(0.0f32, 0.0);
        let st = unsafe { comms_fir_run(self.h, input, 1, &mut out) };
        if st == COMMS_OK { Ok(out) } else { Err(to_err(st)) }
    }
    pub fn run_block(&mut self, input: &[Complex<f32>]) -> Result<Vec<Complex<f32>>, NodeError> {
        let mut out = vec![Complex::new(0.0f32, 0.0); input.len()];
        let st = unsafe { comms_fir_run(self.h, input.as_ptr(), input.len(), out.as_mut_ptr()) };
        if st == COMMS_OK { Ok(out) } else { Err(to_err(st)) }
    }
}
drained_node!(FirNode, Complex<f32>, Complex<f32>);

/// mixer.rs:93-148 -- argument order (dphase, phase) as in MixerNode::new (drained, see `drained_node!`)
pub struct MixerNode {
    pub input: NodeReceiver<Complex<f32>>,
    h: *mut comms_mixer_t,
    pub output: NodeSender<Complex<f32>>,
}
handle_node!(MixerNode, comms_mixer_t, comms_mixer_destroy);
impl MixerNode {
    pub fn new(dphase: f64, phase: Option<f64>) -> Self {
        let mut h = ptr::null_mut();
        let st = unsafe { comms_mixer_create(dphase, phase.unwrap_or(0.0), 0, &mut h) };
        assert_eq!(st, COMMS_OK, "comms_mixer_create failed");
        MixerNode { input: Default::default(), h, output: Default::default() }
    }
    pub fn run(&mut self, input: &Complex<f32>) -> Result<Complex<f32>, NodeError> {
        let mut out = Complex::new(0.0f32, 0.0);
        let st = unsafe { comms_mixer_run(self.h, input, 1, &mut out) };
        if st == COMMS_OK { Ok(out) } else { Err(to_err(st)) }
    }
    pub fn run_block(&mut self, input: &[Complex<f32>]) -> Result<Vec<Complex<f32>>, NodeError> {
        let mut out = vec![Complex::new(0.0f32, 0.0); input.len()];
        let st = unsafe { comms_mixer_run(self.h, input.as_ptr(), input.len(), out.as_mut_ptr()) };
        if st == COMMS_OK { Ok(out) } else { Err(to_err(st)) }
    }
    /// oscillator phase of the next sample: checkpoint hook / start phase of a stream shard
    pub fn phase(&self) -> f64 { let mut p = 0.0; unsafe { comms_mixer_get_phase(self.h, &mut p) }; p }
    pub fn set_phase(&mut self, phase: f64) { unsafe { comms_mixer_set_phase(self.h, phase) }; }
}
drained_node!(MixerNode, Complex<f32>, Complex<f32>);

/// `MixerNode<f64>` (mixer.rs:93-148 with T = f64 -- the instantiation the reference's own mixer tests use)
pub struct MixerNode64 {
    pub input: NodeReceiver<Complex<f64>>,
    h: *mut comms_mixer_t,
    pub output: NodeSender<Complex<f64>>,
}
handle_node!(MixerNode64, comms_mixer_t, comms_mixer_destroy);
impl MixerNode64 {
    pub fn new(dphase: f64, phase: Option<f64>) -> Self {
        let mut h = ptr::null_mut();
        let st = unsafe { comms_mixer_create(dphase, phase.unwrap_or(0.0), 0, &mut h) };
        assert_eq!(st, COMMS_OK, "comms_mixer_create failed");
        MixerNode64 { input: Default::default(), h, output: Default::default() }
    }
    pub fn run(&mut self, input: &Complex<f64>) -> Result<Complex<f64>, NodeError> {
        let mut out = Complex::new(0.0f64, 0.0);
        let st = unsafe { comms_mixer_run_f64(self.h, input, 1, &mut out) };
        if st == COMMS_OK { Ok(out) } else { Err(to_err(st)) }
    }
    pub fn run_block(&mut self, input: &[Complex<f64>]) -> Result<Vec<Complex<f64>>, NodeError> {
        let mut out = vec![Complex::new(0.0f64, 0.0); input.len()];
        let st = unsafe { comms_mixer_run_f64(self.h, input.as_ptr(), input.len(), out.as_mut_ptr()) };
        if st == COMMS_OK { Ok(out) } else { Err(to_err(st)) }
    }
}
drained_node!(MixerNode64, Complex<f64>, Complex<f64>);

/// pulse.rs:38-93
#[derive(Node)]
#[pass_by_ref]
pub struct PulseNode {
    pub input: NodeReceiver<Complex<f32>>,
    h: *mut comms_pulse_t,
    sam_per_sym: usize,
    pub output: NodeSender<Vec<Complex<f32>>>,
}
handle_node!(PulseNode, comms_pulse_t, comms_pulse_destroy);
impl PulseNode {
    pub fn new(taps: Vec<Complex<f32>>, sam_per_sym: usize) -> Self {
        let mut h = ptr::null_mut();
        let st = unsafe { comms_pulse_create(taps.as_ptr(), taps.len(), sam_per_sym, 0, &mut h) };
        assert_eq!(st, COMMS_OK, "comms_pulse_create failed");
        PulseNode { input: Default::default(), h, sam_per_sym, output: Default::default() }
    }
    /// Transmit chain in one launch: fuses the `MixerNode::new(dphase, phase)` that follows.
    pub fn with_mixer(self, dphase: f64, phase: Option<f64>) -> Self {
        let st = unsafe { comms_pulse_set_mixer(self.h, dphase, phase.unwrap_or(0.0)) };
        assert_eq!(st, COMMS_OK, "comms_pulse_set_mixer failed");
        self
    }
    pub fn run(&mut self, input: &Complex<f32>) -> Result<Vec<Complex<f32>>, NodeError> {
        let mut out = vec![Complex::new(0.0f32, 0.0); self.sam_per_sym];
        let st = unsafe { comms_pulse_run(self.h, input, 1, out.as_mut_ptr()) };
        if st == COMMS_OK { Ok(out) } else { Err(to_err(st)) }
    }
}

/// resample_node.rs:10-66 (T = Complex<f32>; any Copy T works through `elem`)
#[derive(Node)]
#[pass_by_ref]
pub struct DecimateNode {
    pub input: NodeReceiver<Vec<Complex<f32>>>,
    dec_rate: usize,
    pub output: NodeSender<Vec<Complex<f32>>>,
}
impl DecimateNode {
    pub fn new(dec_rate: usize) -> Self {
        DecimateNode { dec_rate, input: Default::default(), output: Default::default() }
    }
    pub fn run(&mut self, signal: &[Complex<f32>]) -> Result<Vec<Complex<f32>>, NodeError> {
        let mut n_out = 0usize;
        unsafe { comms_decimate_out_len(signal.len(), self.dec_rate, &mut n_out) };
        let mut out = vec![Complex::new(0.0f32, 0.0); n_out];
        let st = unsafe {
            comms_decimate_run(signal.as_ptr() as *const _, signal.len(), 8, self.dec_rate,
                               out.as_mut_ptr() as *mut _, ptr::null_mut(), 0)
        };
        if st == COMMS_OK { Ok(out) } else { Err(to_err(st)) }
    }
}

/// analog_node.rs:20-52
#[derive(Node)]
#[pass_by_ref]
pub struct FMDemodNode {
    pub input: NodeReceiver<Vec<Complex<f32>>>,
    h: *mut comms_fmdemod_t,
    pub output: NodeSender<Vec<f32>>,
}
handle_node!(FMDemodNode, comms_fmdemod_t, comms_fmdemod_destroy);
impl FMDemodNode {
    pub fn new() -> Self {
        let mut h = ptr::null_mut();
        let st = unsafe { comms_fmdemod_create(0, &mut h) };
        assert_eq!(st, COMMS_OK, "comms_fmdemod_create failed");
        FMDemodNode { input: Default::default(), h, output: Default::default() }
    }
    pub fn run(&mut self, samples: &[Complex<f32>]) -> Result<Vec<f32>, NodeError> {
        let mut out = vec![0.0f32; samples.len()];
        let st = unsafe { comms_fmdemod_run(self.h, samples.as_ptr(), samples.len(), out.as_mut_ptr()) };
        if st == COMMS_OK { Ok(out) } else { Err(to_err(st)) }
    }
}

/// fft_node.rs:28-84
#[derive(Node)]
#[pass_by_ref]
pub struct FFTBatchNode {
    pub input: NodeReceiver<Vec<Complex<f32>>>,
    h: *mut comms_fft_t,
    pub output: NodeSender<Vec<Complex<f32>>>,
}
handle_node!(FFTBatchNode, comms_fft_t, comms_fft_destroy);
impl FFTBatchNode {
    pub fn new(fft_size: usize, ifft: bool) -> Self {
        let mut h = ptr::null_mut();
        let st = unsafe { comms_fft_create(fft_size, ifft as i32, 0, &mut h) };
        assert_eq!(st, COMMS_OK, "comms_fft_create failed");
        FFTBatchNode { input: Default::default(), h, output: Default::default() }
    }
    pub fn run(&mut self, data: &[Complex<f32>]) -> Result<Vec<Complex<f32>>, NodeError> {
        let mut out = vec![Complex::new(0.0f32, 0.0); data.len()];
        let st = unsafe { comms_fft_run(self.h, data.as_ptr(), data.len(), out.as_mut_ptr()) };
        if st == COMMS_OK { Ok(out) } else { Err(to_err(st)) }   // rustfft would panic on a wrong length
    }
}

/// Device-resident message: `Clone` bumps the library's refcount (the derive macro

/// resample_node.rs:74-132
#[derive(Node)]
#[pass_by_ref]
pub struct UpsampleNode {
    pub input: NodeReceiver<Vec<Complex<f32>>>,
    ups_rate: usize,
    pub output: NodeSender<Vec<Complex<f32>>>,
}
impl UpsampleNode {
    pub fn new(ups_rate: usize) -> Self {
        UpsampleNode { input: Default::default(), ups_rate, output: Default::default() }
    }
    pub fn run(&mut self, signal: &[Complex<f32>]) -> Result<Vec<Complex<f32>>, NodeError> {
        let mut n_out = 0usize;
        unsafe { comms_upsample_out_len(signal.len(), self.ups_rate, &mut n_out) };
        let mut out = vec![Complex::new(0.0f32, 0.0); n_out];
        let st = unsafe {
            comms_upsample_run(signal.as_ptr() as *const _, signal.len(), 8, self.ups_rate,
                               out.as_mut_ptr() as *mut _, &mut n_out, 0)
        };
        if st == COMMS_OK { Ok(out) } else { Err(to_err(st)) }
    }
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
handle_node!(ChainNode, comms_chain_t, comms_chain_destroy);
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
handle_node!(FmChainNode, comms_chain_t, comms_chain_destroy);
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
    pub fn state(&mut self, n_taps: usize) -> (Vec<Complex<f32>>, f64, Complex<f32>) {
        let mut hist = vec![Complex::new(0.0f32, 0.0); n_taps];
        let (mut phase, mut prev) = (0.0f64, Complex::new(0.0f32, 0.0));
        unsafe {
            comms_chain_get_fir_state(self.h, hist.as_mut_ptr(), n_taps);
            comms_chain_get_phase(self.h, &mut phase);
            comms_chain_get_fm_prev(self.h, &mut prev);
        }
        (hist, phase, prev)
    }
    pub fn set_state(&mut self, hist: &[Complex<f32>], phase: f64, prev: Complex<f32>) {
        unsafe {
            comms_chain_set_fir_state(self.h, hist.as_ptr(), hist.len());
            comms_chain_set_phase(self.h, phase);
            comms_chain_set_fm_prev(self.h, &prev);
        }
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
handle_node!(RtlFmChainNode, comms_chain_t, comms_chain_destroy);
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
handle_node!(TimingEstimatorNode, comms_timing_t, comms_timing_destroy);
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
handle_node!(BatchNcoNode, comms_nco_t, comms_nco_destroy);
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

/// demodulation/frequency_estimator.rs:27-42, phase_estimator.rs:26-33, :58-65 (free functions there too)
pub fn frequency_offset_estimate(samples: &[Complex<f64>]) -> f64 {
    let mut out = 0.0f64;
    unsafe { comms_frequency_offset_estimate(samples.as_ptr() as *const f64, samples.len(), &mut out, 0) };
    out
}
pub fn psk_phase_estimate(symbols: &[Complex<f64>], m: u32) -> f64 {
    let mut out = 0.0f64;
    unsafe { comms_psk_phase_estimate(symbols.as_ptr() as *const f64, symbols.len(), m, &mut out, 0) };
    out
}
pub fn qam_phase_estimate(symbols: &[Complex<f64>]) -> f64 {
    let mut out = 0.0f64;
    unsafe { comms_qam_phase_estimate(symbols.as_ptr() as *const f64, symbols.len(), &mut out, 0) };
    out
}

/// clones once per sender, node_derive/src/lib.rs:156).
pub struct DeviceBuf { b: *mut comms_buf_t, pub len: usize }
unsafe impl Send for DeviceBuf {}
impl Clone for DeviceBuf {
    fn clone(&self) -> Self { unsafe { comms_buf_retain(self.b) }; DeviceBuf { b: self.b, len: self.len } }
}
impl Drop for DeviceBuf {
    fn drop(&mut self) { unsafe { comms_buf_release(self.b) }; }
}

/// One stream over several GPUs (host arithmetic of `include/comms_hip.h`, "stream shards"):
/// contiguous shards, and per neighbour pair one hand-over of the raw samples in front of the shard.
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
        let mut st = vec![Complex::new(0.0f32, 0.0); halo.len()];
        unsafe { comms_state_from_halo(halo.as_ptr(), halo.len(), st.as_mut_ptr()) };
        st
    }
    /// oscillator phase of stream sample `first_index` -> `MixerNode::new(dphase, Some(phase))` of the shard's node
    pub fn mixer_phase(phase0: f64, dphase: f64, first_index: i64) -> f64 {
        let mut ph = 0.0f64;
        unsafe { comms_shard_mixer_phase(phase0, dphase, first_index, &mut ph) };
        ph
    }
    /// raw samples a fused chain shard runs through first (outputs dropped): FIR history + the sample FM.prev comes from
    pub fn chain_prefix_len(n_taps: usize, rate: usize, fm_demod: bool) -> usize {
        let mut n = 0usize;
        unsafe { comms_chain_prefix_len(n_taps, rate, fm_demod as i32, &mut n) };
        n
    }
}
