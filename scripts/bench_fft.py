#!/usr/bin/env python3
"""Kernel-level timing of the FFT node (tuning aid). usage: bench_fft.py [log2N ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import comms_rs_amd as c

sizes = [int(a) for a in sys.argv[1:]] or [10, 12, 16, 20]
total = 1 << int(os.environ.get("FFT_TOTAL_LOG", 26))  # points per launch (512 MiB in, 512 MiB out at 26)
s = torch.cuda.current_stream().cuda_stream
x = torch.empty(total, dtype=torch.complex64, device="cuda:0")
y = torch.empty_like(x)
c.synth_iq_dev(x.data_ptr(), total, 0)
for lg in sizes:
    n = 1 << lg
    node = c.FFTBatchNode(n, False)
    for _ in range(int(os.environ.get("FFT_WARM", 40))):  # past the clock transient of a burst's first ~25 launches
        node.run_dev(x.data_ptr(), total, y.data_ptr(), s)
    t = c.KernelTimer(40).attach(node)
    for _ in range(40):
        node.run_dev(x.data_ptr(), total, y.data_ptr(), s)
    ms = t.read_ms(); t.close()
    med = float(np.median(ms))
    print("N=2^%d batch=%d: median %.3f ms -> %.1f Gpoints/s, %.0f GB/s algorithmic (16 B/pt) = %.1f%% of 8 TB/s"
          % (lg, total // n, med, total / med / 1e6, 16 * total / med / 1e6, 16 * total / med / 1e6 / 80), flush=True)
