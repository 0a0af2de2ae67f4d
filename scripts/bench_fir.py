#!/usr/bin/env python3
"""Kernel-level timing of the FIR variants on one GPU (tuning aid, not the judged bench).
usage: python scripts/bench_fir.py [n_taps] [log2 n] [reps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import comms_rs_amd as c

n_taps = int(sys.argv[1]) if len(sys.argv) > 1 else 255
n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 24)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
algos = {"direct": c.FIR_DIRECT, "os1024": c.FIR_OS1024, "os4096": c.FIR_OS4096, "os16k": c.FIR_OS16K}
x = torch.empty(n, dtype=torch.complex64, device="cuda:0")
y = torch.empty_like(x)
c.synth_iq_dev(x.data_ptr(), n, 0)
taps = c.rrc_taps(n_taps, 8.0, 0.35)
s = torch.cuda.current_stream().cuda_stream
for name in (os.environ.get("ALGOS", "os1024,os4096,direct").split(",")):
    try:
        fir = c.BatchFirNode(taps).set_algo(algos[name])
    except c.CommsError as e:
        print(name, "skipped:", e)
        continue
    for _ in range(20):
        fir.run_dev(x.data_ptr(), n, y.data_ptr(), s)
    t = c.KernelTimer(reps).attach(fir)
    for _ in range(reps):
        fir.run_dev(x.data_ptr(), n, y.data_ptr(), s)
    ms = t.read_ms()
    t.close()
    gbs = 16.0 * n / (np.median(ms) * 1e-3) / 1e9
    print("%-7s taps=%d n=2^%d  median %.2f us  min %.2f us  mean %.2f us  -> %.0f GB/s algorithmic (%.1f%% of 8 TB/s), %.1f Gsamples/s"
          % (name, n_taps, int(np.log2(n)), np.median(ms) * 1e3, ms.min() * 1e3, ms.mean() * 1e3, gbs, gbs / 80.0,
             n / (np.median(ms) * 1e-3) / 1e9), flush=True)
