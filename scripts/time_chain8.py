"""Bursts of one rate-8 chain (for A/B runs of knobs of the diagnostic build, one process per setting).
usage: [COMMS_HIP_LIB=...diag.so COMMS_POLY8_NT=0] python3 scripts/time_chain8.py <log2 n> <taps> <fm 0/1> [kernel]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import comms_rs_amd as c

lg, nt, fm = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) != 0
kern = sys.argv[4] if len(sys.argv) > 4 else "poly"
n = 1 << lg
k = np.arange(nt) - (nt - 1) / 2.0
taps = (2 / 16 * np.sinc(2 / 16 * k) * np.hamming(nt)).astype(np.complex64)
x = torch.empty(n, dtype=torch.complex64, device="cuda:0")
c.synth_iq_dev(x.data_ptr(), n, 0)
out = torch.empty(n // 8, dtype=torch.float32 if fm else torch.complex64, device="cuda:0")
node = c.ChainNode(2 * np.pi * 0.05, 0.0, taps, 8, fm, mixer_after_fir=not fm, kernel=kern)
s = torch.cuda.current_stream().cuda_stream
ts = []
for rep in range(10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        node.run_dev(x.data_ptr(), n, out.data_ptr(), s)
    b.record()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) / 20 * 1e3)
print("2^%d taps %d fm %d %s %s: median %.1f us (bursts of 20): %s" % (lg, nt, fm, kern, " ".join("%s=%s" % (k, v) for k, v in os.environ.items() if k.startswith("COMMS_POLY8")),
                                                                  np.median(ts[2:]), " ".join("%.1f" % t for t in ts)), flush=True)
