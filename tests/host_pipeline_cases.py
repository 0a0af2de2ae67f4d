"""Workloads of tests/test_gpu_host_pipeline.py (shared with the single-shot subprocess of that test): name ->
() -> (input, node factory, (input bytes, output bytes) per unit)."""
import numpy as np

def _lpf(n_taps, cutoff):
    k = np.arange(n_taps) - (n_taps - 1) / 2.0
    return (2 * cutoff * np.sinc(2 * cutoff * k) * np.hamming(n_taps)).astype(np.complex64)


def _case_fir_direct():
    import comms_rs_amd as c
    n = (5 << 20) + 12345
    return c.synth_iq(n, 0, 31), lambda: c.BatchFirNode(_lpf(31, 0.1)).set_algo(c.FIR_DIRECT), (8, 8)


def _case_fir_auto():
    import comms_rs_amd as c
    n = (6 << 20) + 777
    return c.synth_iq(n, 0, 32), lambda: c.BatchFirNode(c.rrc_taps(255, 8.0, 0.35)), (8, 8)


def _case_mixer():
    import comms_rs_amd as c
    n = (5 << 20) + 3
    return c.synth_iq(n, 0, 33), lambda: c.MixerNode(0.123, 0.4), (8, 8)


def _case_fmdemod():
    import comms_rs_amd as c
    n = (6 << 20) + 11
    return c.synth_iq(n, 0, 34), lambda: c.FMDemodNode(), (8, 4)


def _case_decimate():
    import comms_rs_amd as c
    n = (9 << 20) + 5  # not a multiple of the rate: the ragged tail is the last chunk's
    return c.synth_iq(n, 0, 35), lambda: c.DecimateNode(3), (24, 8)


def _case_upsample():
    import comms_rs_amd as c
    n = (2 << 20) + 9
    return c.synth_iq(n, 0, 36), lambda: c.UpsampleNode(4), (8, 32)


def _case_fft():
    import comms_rs_amd as c
    n = 4096 * 1300  # 1300 transforms of 4096 points
    return c.synth_iq(n, 0, 37), lambda: c.FFTBatchNode(4096, False), (4096 * 8, 4096 * 8)


def _case_chain_r2():
    import comms_rs_amd as c
    n = 2 * ((6 << 20) + 7)   # mixer -> 63 taps -> keep every 2nd: 16 B in, 8 B out per unit (a chain whose output is worth pipelining)
    return c.synth_iq(n, 0, 38), lambda: c.ChainNode(0.31, 0.2, _lpf(63, 1 / 5.0), 2, False), (16, 8)


CASES = {"fir_direct": _case_fir_direct, "fir_auto": _case_fir_auto, "mixer": _case_mixer, "fmdemod": _case_fmdemod,
         "decimate": _case_decimate, "upsample": _case_upsample, "fft": _case_fft, "chain_r2": _case_chain_r2}
