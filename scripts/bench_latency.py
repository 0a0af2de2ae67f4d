#!/usr/bin/env python3
"""Per-call latency of the host-vector (`*_run`) path vs message length (ctypes front end)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import comms_rs_amd as c

taps = c.rrc_taps(32, 4.0, 0.25)
rng = np.random.default_rng(0)
nodes = {"BatchFirNode(32 taps)": c.BatchFirNode(taps), "MixerNode": c.MixerNode(0.1), "DecimateNode(4)": c.DecimateNode(4),
         "FMDemodNode": c.FMDemodNode(), "FFTBatchNode(64)": c.FFTBatchNode(64, False), "PulseNode(x4)": c.PulseNode(taps, 4),
         "ChainNode(mix->fir->/4->fm)": c.ChainNode(0.1, 0.0, taps, 4, True)}
for name, node in nodes.items():
    line = []
    for n in (64, 1024, 8192, 16384, 65536, 262144):
        xs = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        for _ in range(20):
            node.run(xs)
        ts = []
        for _ in range(100):  # per call, and the median: a 100-call span caught a 50-70 ms interpreter pause at one size every run
            t = time.perf_counter()
            node.run(xs)
            ts.append(time.perf_counter() - t)
        line.append("%d: %.1f us" % (n, float(np.median(ts)) * 1e6))
    print("%-30s %s" % (name, "   ".join(line)), flush=True)
