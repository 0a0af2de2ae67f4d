"""A few launches of one decimating chain for counter passes.  usage: python3 scripts/run_chain_once.py <taps> <rate> [kernel] [log2 n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import comms_rs_amd as c
nt, rate = int(sys.argv[1]), int(sys.argv[2])
kern = sys.argv[3] if len(sys.argv) > 3 else "auto"
lg = int(sys.argv[4]) if len(sys.argv) > 4 else 24
n = ((1 << lg) // rate) * rate
s = torch.cuda.current_stream().cuda_stream
x = torch.empty(n, dtype=torch.complex64, device="cuda:0")
c.synth_iq_dev(x.data_ptr(), n, 0)
k = np.arange(nt) - (nt - 1) / 2.0
taps = (0.4 / rate * np.sinc(0.4 / rate * k) * np.hamming(nt)).astype(np.complex64)
out = torch.empty(n // rate, dtype=torch.complex64, device="cuda:0")
node = c.ChainNode(2 * np.pi * 0.05, 0.0, taps, rate, False, mixer_after_fir=True, kernel=kern)
for _ in range(12):
    node.run_dev(x.data_ptr(), n, out.data_ptr(), s)
torch.cuda.synchronize()
print(node.kernel)
