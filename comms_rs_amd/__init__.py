"""comms_rs_amd -- MI355X (gfx950) DSP hot path for comms-rs style node graphs.

Python front end of the C ABI in include/comms_hip.h (hand-written HIP kernels
in csrc/).  The node classes keep the reference constructors' names and
argument order; the compiled host graph runtime lives in
host/ (C++).  No CPU fallback: the HIP library must be built and a gfx950
device present, or construction raises.
"""
from ._lib import (COMMS_ERR_ARG, COMMS_ERR_DEVICE, COMMS_OK, CommsError, FIR_AUTO, FIR_DIRECT, FIR_OVERLAP_SAVE, FIR_OS1024, FIR_OS4096, FIR_OS16K, FIR_OS1024_FIXED, LIB_PATH, build, lib)  # noqa: F401
from .nodes import (BatchFirNode, BatchFirNodeI16, FirNodeI16, PulseNodeI16, BatchFirNodeF64, FirNodeF64, PulseNodeF64, ChainNode, DecimateNode, DeviceBuf, FFTBatchNode, FFTSampleNode, FFTBatchNodeF64, FFTSampleNodeF64, FMDemodNodeF64,  # noqa: F401
                    FirNode, FMDemodNode, KernelTimer, NcoNode, TimingEstimatorNode, qfilt_taps, frequency_offset_estimate, psk_phase_estimate, qam_phase_estimate, MixerNode, PulseNode, UpsampleNode, device_count,
                    gaussian_taps, iq_c32_to_i16, iq_i16_to_c32, iq_u8_to_c32, real_to_c32_dev, c32_re_dev, rc_taps, rect_taps, rrc_taps,
                    synth_iq, synth_iq_dev)
