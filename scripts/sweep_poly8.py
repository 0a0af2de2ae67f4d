"""Rate-8 chains: the time-domain kernels against the polyphase frequency-domain kernel over taps x batch length (bursts of 20
launches, the kernels taking turns).  usage: python3 scripts/sweep_poly8.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import comms_rs_amd as c


def lpf(n_taps, cutoff):
    k = np.arange(n_taps) - (n_taps - 1) / 2.0
    return (2 * cutoff * np.sinc(2 * cutoff * k) * np.hamming(n_taps)).astype(np.complex64)


s = torch.cuda.current_stream().cuda_stream
x = torch.empty(1 << 26, dtype=torch.complex64, device="cuda:0")
c.synth_iq_dev(x.data_ptr(), 1 << 26, 0)
for fm in (False, True):
    for n_taps in (15, 31, 63, 95, 127, 129, 161, 193, 249):
        taps = lpf(n_taps, 1 / 16.0)
        line = []
        for lg in (14, 16, 18, 20, 22, 24, 26):
            n = 1 << lg
            kerns = ("time", "poly")
            nodes = [c.ChainNode(2 * np.pi * 0.05, 0.0, taps, 8, fm, kernel=k) for k in kerns]
            outs = [torch.empty(n // 8, dtype=torch.float32 if fm else torch.complex64, device="cuda:0") for _ in kerns]
            ts = [[] for _ in kerns]
            for rep in range(5):
                for i, nd in enumerate(nodes):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for _ in range(20):
                        nd.run_dev(x.data_ptr(), n, outs[i].data_ptr(), s)
                    b.record()
                    torch.cuda.synchronize()
                    ts[i].append(a.elapsed_time(b) / 20 * 1e3)
            line.append("2^%d %6.1f /%6.1f" % (lg, np.median(ts[0][1:]), np.median(ts[1][1:])))
        print("fm %d taps %3d (time / poly, us):  %s" % (fm, n_taps, "   ".join(line)), flush=True)
