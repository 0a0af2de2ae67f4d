#!/usr/bin/env python3
"""Kernel timings of the pointwise / reduction nodes (device-resident, torch events on the
current stream; median of 50).  usage: python scripts/bench_pointwise.py [log2 n]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import comms_rs_amd as c
from comms_rs_amd._lib import lib

n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 24)
s = torch.cuda.current_stream().cuda_stream
dev = "cuda:0"


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


def report(name, ms, nbytes, unit_count):
    print("%-44s %8.1f us  %7.1f Gunits/s  %6.0f GB/s algorithmic (%.1f%% of 8 TB/s)"
          % (name, ms * 1e3, unit_count / ms / 1e6, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80.0), flush=True)


x = torch.empty(n, dtype=torch.complex64, device=dev)
c.synth_iq_dev(x.data_ptr(), n, 0)
y = torch.empty(n, dtype=torch.complex64, device=dev)

mx = c.MixerNode(2 * np.pi * 0.1)
report("mixer (16 B/sample)", timeit(lambda: mx.run_dev(x.data_ptr(), n, y.data_ptr(), s)), 16 * n, n)
fm = c.FMDemodNode()
f = torch.empty(n, dtype=torch.float32, device=dev)
report("fm demod (12 B/sample)", timeit(lambda: fm.run_dev(x.data_ptr(), n, f.data_ptr(), s)), 12 * n, n)
for r in (2, 8):
    d = c.DecimateNode(r)
    z = torch.empty(n // r, dtype=torch.complex64, device=dev)
    report("decimate /%d (16 B per output)" % r, timeit(lambda: d.run_dev(x.data_ptr(), n, 8, z.data_ptr(), s)), 16 * (n // r), n)
for r in (4,):
    u = c.UpsampleNode(r)
    z = torch.empty(n, dtype=torch.complex64, device=dev)
    m = n // r
    report("upsample x%d (8 + 8R B per input)" % r, timeit(lambda: u.run_dev(x.data_ptr(), m, 8, z.data_ptr(), s)), (8 + 8 * r) * m, m)
taps = c.rrc_taps(63, 4.0, 0.25)
p = c.PulseNode(taps, 4)
m = n // 4
report("pulse 63 taps x4 (2 B rd + 8 B wr per output)", timeit(lambda: p.run_dev(x.data_ptr(), m, y.data_ptr(), s)), 10 * n, n)
pm = c.PulseNode(taps, 4).set_mixer(2 * np.pi * 0.1)
report("pulse x4 + fused mixer (transmit chain, one launch)", timeit(lambda: pm.run_dev(x.data_ptr(), m, y.data_ptr(), s)), 10 * n, n)
y2 = torch.empty_like(y)
def two_nodes():
    p.run_dev(x.data_ptr(), m, y.data_ptr(), s)
    mx.run_dev(y.data_ptr(), n, y2.data_ptr(), s)
report("pulse x4 -> mixer as two nodes", timeit(two_nodes), 10 * n, n)
m1 = (1 << 20) // 4  # BASELINE config 1: 262144 symbols -> 2^20 samples
report("config 1 size: pulse x4 + fused mixer, 2^20 samples", timeit(lambda: pm.run_dev(x.data_ptr(), m1, y.data_ptr(), s)), 10 * (1 << 20), 1 << 20)
def two_nodes_c1():
    p.run_dev(x.data_ptr(), m1, y.data_ptr(), s)
    mx.run_dev(y.data_ptr(), 1 << 20, y2.data_ptr(), s)
report("config 1 size: pulse x4 -> mixer as two nodes", timeit(two_nodes_c1), 10 * (1 << 20), 1 << 20)
# raw IQ formats
L = lib()
i16 = torch.zeros(n * 2, dtype=torch.int16, device=dev)
u8 = torch.zeros(n * 2, dtype=torch.uint8, device=dev)
report("iq i16 -> c32 (12 B/sample)", timeit(lambda: L.comms_iq_i16_to_c32_dev(i16.data_ptr(), n, 1.0, y.data_ptr(), 0, s)), 12 * n, n)
report("iq c32 -> i16 (12 B/sample)", timeit(lambda: L.comms_iq_c32_to_i16_dev(x.data_ptr(), n, 8192.0, i16.data_ptr(), 0, s)), 12 * n, n)
report("iq u8 -> c32 (10 B/sample)", timeit(lambda: L.comms_iq_u8_to_c32_dev(u8.data_ptr(), n, y.data_ptr(), 0, s)), 10 * n, n)
# estimators (f64 in)
import ctypes as C
xd = torch.randn(n, dtype=torch.complex128, device=dev)
out = C.c_double()
report("frequency_offset_estimate (16 B/sample, f64)", timeit(lambda: L.comms_frequency_offset_estimate_dev(xd.data_ptr(), n, C.byref(out), 0, s), 20), 16 * n, n)
report("psk_phase_estimate m=4 (16 B/sample, f64)", timeit(lambda: L.comms_psk_phase_estimate_dev(xd.data_ptr(), n, 4, C.byref(out), 0, s), 20), 16 * n, n)
te = c.TimingEstimatorNode(10, 5, 0.5)
report("timing estimator n=10 d=5 (16 B/sample, f64)", timeit(lambda: te.run_dev(xd.data_ptr(), n, s), 20), 16 * n, n)
nco = c.NcoNode(0.1)
pe = torch.zeros(n, dtype=torch.float64, device=dev)
yo = torch.empty(n, dtype=torch.complex128, device=dev)
report("nco block (8 B in + 16 B out per sample)", timeit(lambda: nco.run_dev(pe.data_ptr(), n, yo.data_ptr(), s), 20), 24 * n, n)
