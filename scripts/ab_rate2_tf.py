"""Rate-2 (and 3) chains: the per-rate time-domain kernel against the overlap-save fusion, the two taking turns (bursts of 20).
usage: python3 scripts/ab_rate2_tf.py [rate]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import comms_rs_amd as c
rate = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = ((1 << 24) // rate) * rate
s = torch.cuda.current_stream().cuda_stream
x = torch.empty(n, dtype=torch.complex64, device="cuda:0")
c.synth_iq_dev(x.data_ptr(), n, 0)
for fm in (False, True):
    for nt in (31, 47, 63, 79, 95, 127):
        k = np.arange(nt) - (nt - 1) / 2.0
        taps = (0.4 / rate * 2 * np.sinc(0.4 / rate * 2 * k) * np.hamming(nt)).astype(np.complex64)
        out = torch.empty(n // rate, dtype=torch.float32 if fm else torch.complex64, device="cuda:0")
        nodes = [c.ChainNode(2 * np.pi * 0.05, 0.0, taps, rate, fm, mixer_after_fir=not fm, kernel=kk) for kk in ("time", "freq")]
        ts = [[], []]
        for rep in range(8):
            for i, node in enumerate(nodes):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(20):
                    node.run_dev(x.data_ptr(), n, out.data_ptr(), s)
                b.record(); torch.cuda.synchronize()
                ts[i].append(a.elapsed_time(b) / 20 * 1e3)
        auto = c.ChainNode(2 * np.pi * 0.05, 0.0, taps, rate, fm, mixer_after_fir=not fm).kernel
        print("rate %d fm %d taps %3d: time %.1f  freq %.1f  (auto picks %s)" % (rate, fm, nt, np.median(ts[0][2:]), np.median(ts[1][2:]), auto), flush=True)
