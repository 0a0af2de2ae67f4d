"""Decimating chains of 1600 ... 4097 taps at 2^24 samples: the one launch (16384-point kernel, mixer and decimator in its store
stage) against the series of launches (unfused=True), bursts of 20."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import comms_rs_amd as c
n0 = 1 << 24
s = torch.cuda.current_stream().cuda_stream
x = torch.empty(n0, dtype=torch.complex64, device="cuda:0")
c.synth_iq_dev(x.data_ptr(), n0, 0)
for fm in (False, True):
    for rate in (5, 8, 16, 100):
        n = (n0 // rate) * rate
        row = []
        for nt in (1600, 2049, 3073, 4097):
            k = np.arange(nt) - (nt - 1) / 2.0
            taps = (2 / (2.5 * rate) * np.sinc(2 / (2.5 * rate) * k) * np.hamming(nt)).astype(np.complex64)
            out = torch.empty(n // rate, dtype=torch.float32 if fm else torch.complex64, device="cuda:0")
            cell = []
            for kw in (dict(), dict(unfused=True)):
                node = c.ChainNode(2 * np.pi * 0.05, 0.0, taps, rate, fm, mixer_after_fir=not fm, **kw)
                ts = []
                for rep in range(5):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for _ in range(20):
                        node.run_dev(x.data_ptr(), n, out.data_ptr(), s)
                    b.record(); torch.cuda.synchronize()
                    ts.append(a.elapsed_time(b) / 20 * 1e3)
                cell.append("%.1f %s" % (np.median(ts[1:]), node.kernel))
            row.append("%d: %s" % (nt, " / ".join(cell)))
        print("rate %3d fm %d  %s" % (rate, fm, "   ".join(row)), flush=True)
