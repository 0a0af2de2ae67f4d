#!/usr/bin/env python3
"""What would a hipGraph of the metric's three launches (FIR -> mixer -> decimate) save?

Timing probe only: the captured graph replays the SAME kernel arguments (mixer phase, history
ping-pong index), so its outputs are those of the first step -- good for the dispatch-gap question,
not a product path.  Prints ms per step for plain launches and for graph replays on the same box.
Usage: python scripts/probe_graph.py [log2_samples=24] [steps=200]
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np
import torch

import comms_rs_amd as c


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    n = 1 << lg
    dev = torch.device("cuda", 0)
    taps = c.rrc_taps(255, 8.0, 0.35)
    x = torch.empty(n, dtype=torch.complex64, device=dev)
    y = torch.empty(n, dtype=torch.complex64, device=dev)
    z = torch.empty(n // 8, dtype=torch.complex64, device=dev)
    side = torch.cuda.Stream()
    s = side.cuda_stream
    c.synth_iq_dev(x.data_ptr(), n, 0, 0xC0FFEE, device=0, stream=s)
    fir = c.BatchFirNode(taps, device=0)
    mixer = c.MixerNode(2 * np.pi * 0.1, 0.0, device=0)
    dec = c.DecimateNode(8, device=0)

    def step():
        fir.run_dev(x.data_ptr(), n, y.data_ptr(), s)
        mixer.run_dev(y.data_ptr(), n, y.data_ptr(), s)
        dec.run_dev(y.data_ptr(), n, 8, z.data_ptr(), s)

    def timed(fn, k):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k * 1e3

    for rep in range(3):
        print("plain launches     : %.4f ms/step" % timed(step, steps), flush=True)
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            g.capture_begin(capture_error_mode="relaxed")
            step()
            g.capture_end()
        for rep in range(3):
            with torch.cuda.stream(side):
                print("graph of 3 launches: %.4f ms/step" % timed(g.replay, steps), flush=True)
        g10 = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            g10.capture_begin(capture_error_mode="relaxed")
            for _ in range(10):
                step()
            g10.capture_end()
        for rep in range(3):
            with torch.cuda.stream(side):
                print("graph of 30 launches: %.4f ms/step" % (timed(g10.replay, max(steps // 10, 1)) / 10), flush=True)
    except Exception as e:  # capture refused: say why, the plain figures above stand
        print("capture failed: %r" % (e,), flush=True)
    for rep in range(2):
        print("plain launches     : %.4f ms/step" % timed(step, steps), flush=True)


if __name__ == "__main__":
    main()
