import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import comms_rs_amd as c
s = torch.cuda.current_stream().cuda_stream
for rate in (2, 4):
    n = ((1 << 24) + (1 << 20) + 128 * rate * 777)
    n -= n % rate
    x = torch.empty(n, dtype=torch.complex64, device="cuda:0")
    c.synth_iq_dev(x.data_ptr(), n, 0)
    row = []
    for nt in (15, 31, 47, 63):
        k = np.arange(nt) - (nt - 1) / 2.0
        taps = (0.4 * np.sinc(0.4 * k) * np.hamming(nt)).astype(np.complex64)
        out = torch.empty(n // rate, dtype=torch.complex64, device="cuda:0")
        node = c.ChainNode(2 * np.pi * 0.05, 0.0, taps, rate, False, mixer_after_fir=True, kernel="time")
        ts = []
        for rep in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                node.run_dev(x.data_ptr(), n, out.data_ptr(), s)
            b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 20 * 1e3)
        row.append("%d: %.1f" % (nt, np.median(ts[1:])))
    print("unbalanced batch (workgroup kernel) rate", rate, " ".join(row), flush=True)
