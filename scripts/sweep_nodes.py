#!/usr/bin/env python3
"""Every hot-path node over batch sizes 2^12 ... 2^24, device-resident, launch to completion (events around the call,
median of 30 after 10 of warm-up): Gsamples/s consumed.  Shows where a node is launch-bound and where the kernel choice
changes.  usage: python scripts/sweep_nodes.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import comms_rs_amd as c

dev = "cuda:0"
s = torch.cuda.current_stream().cuda_stream
NMAX = 1 << 24
x = torch.empty(NMAX, dtype=torch.complex64, device=dev)
c.synth_iq_dev(x.data_ptr(), NMAX, 0, 1)
y = torch.empty(NMAX, dtype=torch.complex64, device=dev)
f = torch.empty(NMAX, dtype=torch.float32, device=dev)


def lp(n_taps, cut):
    k = np.arange(n_taps) - (n_taps - 1) / 2
    return (2 * cut * np.sinc(2 * cut * k) * np.hamming(n_taps)).astype(np.float32).astype(np.complex64)


def timeit(fn, reps=30):
    for _ in range(10):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3  # us


fir63, fir255, fir1025 = c.BatchFirNode(lp(63, 0.1)), c.BatchFirNode(c.rrc_taps(255, 8.0, 0.35)), c.BatchFirNode(lp(1025, 0.05))
mixer, fm, dec8, up4 = c.MixerNode(0.3), c.FMDemodNode(), c.DecimateNode(8), c.UpsampleNode(4)
pulse = c.PulseNode(c.rrc_taps(63, 4.0, 0.25), 4)
fft1k, fft64k = c.FFTBatchNode(1024, False), c.FFTBatchNode(1 << 16, False)
chain = c.ChainNode(2 * np.pi * 0.05, 0.0, lp(127, 1 / 16), 8, True)
rows = [
    ("BatchFirNode 63 taps", lambda n: fir63.run_dev(x.data_ptr(), n, y.data_ptr(), s), lambda n: fir63.kernel_for(n)),
    ("BatchFirNode 255 taps", lambda n: fir255.run_dev(x.data_ptr(), n, y.data_ptr(), s), lambda n: fir255.kernel_for(n)),
    ("BatchFirNode 1025 taps", lambda n: fir1025.run_dev(x.data_ptr(), n, y.data_ptr(), s), lambda n: fir1025.kernel_for(n)),
    ("MixerNode", lambda n: mixer.run_dev(x.data_ptr(), n, y.data_ptr(), s), None),
    ("FMDemodNode", lambda n: fm.run_dev(x.data_ptr(), n, f.data_ptr(), s), None),
    ("DecimateNode 8", lambda n: dec8.run_dev(x.data_ptr(), n, 8, y.data_ptr(), s), None),
    ("UpsampleNode 4 (n inputs)", lambda n: up4.run_dev(x.data_ptr(), n // 4, 8, y.data_ptr(), s), None),
    ("PulseNode 63 taps x4 (n outputs)", lambda n: pulse.run_dev(x.data_ptr(), n // 4, y.data_ptr(), s), None),
    ("FFTBatchNode 1024", lambda n: fft1k.run_dev(x.data_ptr(), n, y.data_ptr(), s), None),
    ("FFTBatchNode 65536", lambda n: fft64k.run_dev(x.data_ptr(), n, y.data_ptr(), s) if n >= (1 << 16) else None, None),
    ("ChainNode mixer->127 taps->/8->FM", lambda n: chain.run_dev(x.data_ptr(), n, f.data_ptr(), s), None),
]
sizes = [12, 14, 16, 18, 20, 22, 24]
print("%-36s" % "node \\ log2 n" + "".join("%14d" % lg for lg in sizes))
for name, fn, kern in rows:
    cells = []
    for lg in sizes:
        n = 1 << lg
        if name.startswith("FFTBatchNode 65536") and n < (1 << 16):
            cells.append("%14s" % "-")
            continue
        us = timeit(lambda: fn(n))
        cells.append("%7.1fus %5.1fG" % (us, n / us / 1e3))
    print("%-36s" % name + "".join(cells), flush=True)
    if kern:
        print("%-36s" % "  kernel" + "".join("%14s" % kern(1 << lg).replace("fir_", "").replace("_kernel", "")[:13] for lg in sizes))
