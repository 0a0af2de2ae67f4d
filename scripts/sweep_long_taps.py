"""Decimating chains with filters longer than 257 taps (and 250-257 taps with FM demod): which kernel the product picks and
what a launch costs, bursts of 20.  usage: python3 scripts/sweep_long_taps.py [kernel: auto|poly|time]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import comms_rs_amd as c

kern = sys.argv[1] if len(sys.argv) > 1 else "auto"
s = torch.cuda.current_stream().cuda_stream
for lg in (20, 24):
    n = 1 << lg
    x = torch.empty(n, dtype=torch.complex64, device="cuda:0")
    c.synth_iq_dev(x.data_ptr(), n, 0)
    for fm in (False, True):
        for rate in (4, 8, 12, 16, 20, 32, 48, 64):
            if fm and rate > 8:
                continue
            row = []
            for nt in (249, 255, 300, 383, 449, 511):
                k = np.arange(nt) - (nt - 1) / 2.0
                taps = (2 / 16 * np.sinc(2 / 16 * k) * np.hamming(nt)).astype(np.complex64)
                m = (n // rate) * rate
                out = torch.empty(m // rate, dtype=torch.float32 if fm else torch.complex64, device="cuda:0")
                try:
                    node = c.ChainNode(2 * np.pi * 0.05, 0.0, taps, rate, fm, mixer_after_fir=not fm, kernel=kern)
                except Exception as e:  # the forced kernel cannot run this chain
                    row.append("%d: -" % nt)
                    continue
                ts = []
                for rep in range(6):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for _ in range(20):
                        node.run_dev(x.data_ptr(), m, out.data_ptr(), s)
                    b.record()
                    torch.cuda.synchronize()
                    ts.append(a.elapsed_time(b) / 20 * 1e3)
                row.append("%d: %.1f (%s)" % (nt, np.median(ts[1:]), node.kernel))
            print("2^%d rate %2d fm %d [%s]  %s" % (lg, rate, fm, kern, "  ".join(row)), flush=True)
