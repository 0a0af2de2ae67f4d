"""Map of decimating chains (mixer after the FIR, no FM demod) at 2^24 samples: us per launch and the kernel the chain ran on, by
rate and tap count; the last column block is the time a plain pass over the chain's bytes would take at 5.5 TB/s.
usage: python3 scripts/map_chains.py [log2 n] [fm 0/1]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import comms_rs_amd as c

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
fm = len(sys.argv) > 2 and int(sys.argv[2]) != 0
n0 = 1 << lg
s = torch.cuda.current_stream().cuda_stream
x = torch.empty(n0, dtype=torch.complex64, device="cuda:0")
c.synth_iq_dev(x.data_ptr(), n0, 0)
short = {"poly": "P", "time": "T", "time_any": "A", "freq": "F", "unfused": "U"}
tapset = (31, 63, 127, 255, 383, 511, 769, 1025)
print("2^%d samples, fm %d; us per launch (bursts of 20) [kernel: P polyphase, T per-rate time-domain, A any-rate, F overlap-save fusion, U series]" % (lg, fm))
print("rate  floor " + " ".join("%9d" % t for t in tapset))
for rate in (2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 16, 20, 24, 32, 50, 64, 100):
    n = (n0 // rate) * rate
    floor = (8 * n + (4 if fm else 8) * n / rate) / 5.5e12 * 1e6
    row = []
    for nt in tapset:
        k = np.arange(nt) - (nt - 1) / 2.0
        taps = (2 / (2.5 * rate) * np.sinc(2 / (2.5 * rate) * k) * np.hamming(nt)).astype(np.complex64)
        out = torch.empty(n // rate, dtype=torch.float32 if fm else torch.complex64, device="cuda:0")
        node = c.ChainNode(2 * np.pi * 0.05, 0.0, taps, rate, fm, mixer_after_fir=not fm)
        ts = []
        for rep in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                node.run_dev(x.data_ptr(), n, out.data_ptr(), s)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 20 * 1e3)
        row.append("%7.1f %s" % (np.median(ts[1:]), short.get(node.kernel, node.kernel)))
    print("%4d %6.1f " % (rate, floor) + " ".join(row), flush=True)
