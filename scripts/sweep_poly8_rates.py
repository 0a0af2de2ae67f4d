"""Rates 4 and 8 m: the chain's other kernels (time-domain per-rate / any-rate, overlap-save fusion) against the polyphase
frequency-domain kernel over taps x batch length.  usage: python3 scripts/sweep_poly8_rates.py"""
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
for rate in (4, 16, 32, 64):
    for n_taps in (31, 64, 96, 127, 193, 255):
        taps = lpf(n_taps, 1 / (2.5 * rate))
        line = []
        for lg in (16, 20, 24, 26):
            n = (1 << lg) - (1 << lg) % (rate * 64)
            kerns = ("time", "freq", "poly")
            nodes = [c.ChainNode(2 * np.pi * 0.05, 0.0, taps, rate, False, mixer_after_fir=True, kernel=k) for k in kerns]
            outs = [torch.empty(n // rate, dtype=torch.complex64, device="cuda:0") for _ in kerns]
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
            line.append("2^%d %5.1f /%5.1f /%5.1f" % (lg, *(np.median(t[1:]) for t in ts)))
        print("rate %2d taps %3d (%s / freq / %s, us):  %s" % (rate, n_taps, nodes[0].kernel, nodes[2].kernel, "   ".join(line)), flush=True)
