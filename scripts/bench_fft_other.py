#!/usr/bin/env python3
"""Timing of non-power-of-two FFT lengths (exact-index DFT up to 4096, Bluestein above)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import comms_rs_amd as c

s = torch.cuda.current_stream().cuda_stream
for n in [int(a) for a in sys.argv[1:]] or [10, 12, 100, 600, 1000, 1200, 1536, 3000, 4095, 5000, 10000, 100000]:
    batch = max(1, (1 << 22) // n)
    total = n * batch
    x = torch.empty(total, dtype=torch.complex64, device="cuda:0")
    y = torch.empty_like(x)
    c.synth_iq_dev(x.data_ptr(), total, 0)
    node = c.FFTBatchNode(n, False)
    for _ in range(2):
        node.run_dev(x.data_ptr(), total, y.data_ptr(), s)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record(); node.run_dev(x.data_ptr(), total, y.data_ptr(), s); b.record()
    torch.cuda.synchronize()
    ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
    print("N=%d batch=%d: %.3f ms -> %.2f Gpoints/s" % (n, batch, ms, total / ms / 1e6), flush=True)
