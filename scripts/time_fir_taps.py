import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import comms_rs_amd as c
n = 1 << 24
s = torch.cuda.current_stream().cuda_stream
x = torch.empty(n, dtype=torch.complex64, device="cuda:0"); y = torch.empty_like(x)
c.synth_iq_dev(x.data_ptr(), n, 0)
for nt in (255, 300, 511, 769, 1025, 1537, 2049):
    k = np.arange(nt) - (nt - 1) / 2.0
    taps = (0.1 * np.sinc(0.1 * k) * np.hamming(nt)).astype(np.complex64)
    node = c.BatchFirNode(taps)
    ts = []
    for rep in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            node.run_dev(x.data_ptr(), n, y.data_ptr(), s)
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 20 * 1e3)
    print("BatchFirNode %d taps 2^24: %.1f us (algo %s)" % (nt, np.median(ts[1:]), node.algo_for(n)), flush=True)
