import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import comms_rs_amd as c
n0 = 1 << 24
s = torch.cuda.current_stream().cuda_stream
x = torch.empty(n0, dtype=torch.complex64, device="cuda:0")
c.synth_iq_dev(x.data_ptr(), n0, 0)
for rate in (11, 12):
    n = (n0 // rate) * rate
    for fm in (False, True):
        row = []
        for nt in (31, 63, 127, 255):
            k = np.arange(nt) - (nt - 1) / 2.0
            taps = (2 / (2.5 * rate) * np.sinc(2 / (2.5 * rate) * k) * np.hamming(nt)).astype(np.complex64)
            out = torch.empty(n // rate, dtype=torch.float32 if fm else torch.complex64, device="cuda:0")
            node = c.ChainNode(2 * np.pi * 0.05, 0.0, taps, rate, fm, mixer_after_fir=False, kernel="time")
            ts = []
            for rep in range(6):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(20):
                    node.run_dev(x.data_ptr(), n, out.data_ptr(), s)
                b.record(); torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) / 20 * 1e3)
            row.append("%d: %.1f (%s)" % (nt, np.median(ts[1:]), node.kernel))
        print(os.path.basename(os.environ.get("COMMS_HIP_LIB", "product")), "rate", rate, "mixer first, fm", int(fm), "  ".join(row), flush=True)
