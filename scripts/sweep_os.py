"""os4096 against os16k over tap counts and lengths (kernel events; tuning aid for fir_pick's threshold).
usage: python scripts/sweep_os.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import comms_rs_amd as c

s = torch.cuda.current_stream().cuda_stream
for lg in (20, 22, 24, 26):
    n = 1 << lg
    x = torch.empty(n, dtype=torch.complex64, device="cuda:0"); y = torch.empty_like(x)
    c.synth_iq_dev(x.data_ptr(), n, 0)
    for taps_n in (300, 513, 769, 1025, 1281, 1537, 1793, 2049):
        taps = c.rrc_taps(taps_n, 8.0, 0.35)
        row = []
        for name, algo in (("os4096", c.FIR_OS4096), ("os16k", c.FIR_OS16K)):
            fir = c.BatchFirNode(taps).set_algo(algo)
            for _ in range(5):
                fir.run_dev(x.data_ptr(), n, y.data_ptr(), s)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
            for a, b in ev:
                a.record(); fir.run_dev(x.data_ptr(), n, y.data_ptr(), s); b.record()
            torch.cuda.synchronize()
            row.append(np.median([a.elapsed_time(b) for a, b in ev]) * 1e3)
        print("n=2^%d taps=%4d  os4096 %8.1f us   os16k %8.1f us   %s" % (lg, taps_n, row[0], row[1], "os16k" if row[1] < row[0] else "os4096"), flush=True)
