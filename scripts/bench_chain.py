#!/usr/bin/env python3
"""Timing of the fused chains (tuning aid): config 2 (FIR 255 -> mixer -> /8, 2^24) and
config 3 (mixer -> 127-tap LPF -> /8 -> FM demod, 2^26), fused and node by node.
usage: python scripts/bench_chain.py [reps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import comms_rs_amd as c

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
s = torch.cuda.current_stream().cuda_stream


def lpf(n_taps, cutoff):
    k = np.arange(n_taps) - (n_taps - 1) / 2.0
    return (2 * cutoff * np.sinc(2 * cutoff * k) * np.hamming(n_taps)).astype(np.complex64)


def timeit(fn):
    for _ in range(5):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


for name, logn, taps, after, fm in (("config2 FIR255->mix->/8", 24, c.rrc_taps(255, 8.0, 0.35), True, False),
                                    ("config3 mix->LPF127->/8->FM", 26, lpf(127, 1 / 16.0), False, True),
                                    ("fm_radio literal mix->LPF63->/5->FM", 26, lpf(63, 1 / 10.0), False, True),
                                    ("mix->LPF255->/4 (64 MACs/sample)", 24, lpf(255, 1 / 8.0), False, False),
                                    ("mix->LPF255->/2 (128 MACs/sample)", 24, lpf(255, 1 / 4.0), False, False),
                                    ("mix->cplx127->/8 (32 MACs/sample)", 24, lpf(127, 1 / 16.0) * np.exp(0.2j * np.arange(127)).astype(np.complex64), False, False)):
    if os.environ.get("CHAIN_ONLY", "") not in name:
        continue
    n = 1 << int(os.environ.get("CHAIN_LOGN", logn))
    rate = int(name.split("->/")[1].split("-")[0].split(" ")[0])
    n -= n % (rate * 1024)
    x = torch.empty(n, dtype=torch.complex64, device="cuda:0")
    c.synth_iq_dev(x.data_ptr(), n, 0)
    out = torch.empty(n // rate, dtype=torch.float32 if fm else torch.complex64, device="cuda:0")
    for kernel, unfused in (("time", False), ("freq", False), ("auto", True)):
        if os.environ.get("KERNELS") and ("unfused" if unfused else kernel) not in os.environ["KERNELS"].split(","):
            continue
        ch = c.ChainNode(2 * np.pi * 0.05, 0.0, taps, rate, fm, mixer_after_fir=after, unfused=unfused, kernel=kernel)
        ms = timeit(lambda: ch.run_dev(x.data_ptr(), n, out.data_ptr(), s))
        bytes_alg = n * 8 + out.numel() * out.element_size()
        print("%-36s %-8s %.1f us  %.1f Gsamples/s  %.0f GB/s algorithmic (%.1f%% of 8 TB/s)"
              % (name, ch.kernel, ms * 1e3, n / ms / 1e6, bytes_alg / ms / 1e6, bytes_alg / ms / 1e6 / 80.0), flush=True)
