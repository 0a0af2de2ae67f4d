"""A few launches of the rate-2 chain (31 real taps, 2^24 samples) for counter passes.  usage: python3 scripts/run_rate2_once.py [taps] [rate]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import comms_rs_amd as c
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 31
rate = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n = ((1 << 24) // rate) * rate
s = torch.cuda.current_stream().cuda_stream
x = torch.empty(n, dtype=torch.complex64, device="cuda:0")
c.synth_iq_dev(x.data_ptr(), n, 0)
k = np.arange(nt) - (nt - 1) / 2.0
taps = (0.4 * np.sinc(0.4 * k) * np.hamming(nt)).astype(np.complex64)
out = torch.empty(n // rate, dtype=torch.complex64, device="cuda:0")
node = c.ChainNode(2 * np.pi * 0.05, 0.0, taps, rate, False, mixer_after_fir=True, kernel="time")
for _ in range(12):
    node.run_dev(x.data_ptr(), n, out.data_ptr(), s)
torch.cuda.synchronize()
