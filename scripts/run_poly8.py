"""Launches of one fused rate-8 chain for the profilers.  usage: python3 scripts/run_poly8.py [log2 n] [launches] [kernel] [taps] [fm]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import comms_rs_amd as c

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
kern = sys.argv[3] if len(sys.argv) > 3 else "poly"
nt = int(sys.argv[4]) if len(sys.argv) > 4 else 255
fm = len(sys.argv) > 5 and sys.argv[5] == "fm"
n = 1 << lg
taps = c.rrc_taps(nt, 8.0, 0.35)
x = torch.empty(n, dtype=torch.complex64, device="cuda:0")
c.synth_iq_dev(x.data_ptr(), n, 0)
out = torch.empty(n // 8, dtype=torch.float32 if fm else torch.complex64, device="cuda:0")
node = c.ChainNode(2 * np.pi * 0.05, 0.1, taps, 8, fm, mixer_after_fir=not fm, kernel=kern)
s = torch.cuda.current_stream().cuda_stream
for _ in range(reps):
    node.run_dev(x.data_ptr(), n, out.data_ptr(), s)
torch.cuda.synchronize()
print("done", node.kernel)
