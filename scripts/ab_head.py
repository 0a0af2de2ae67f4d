"""The headline FIR launch (255 taps, 2^24 samples by default) under the variants that could move it, interleaved in
ONE process on ONE box: segment dealing (contiguous shares / chunks of 2^k) x in-kernel stamps (off / on).  Two
figures per variant: event pairs around single launches (isolated) and the wall time of back-to-back bursts.
usage: python scripts/with_lib.py comms_rs_amd/lib/libcomms_hip_diag.so scripts/ab_head.py [log2 n] [rounds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import comms_rs_amd as c
from comms_rs_amd._lib import lib

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
n = 1 << lg
burst = max(4, min(200, (1 << 31) // n // 8))
set_chunk = lib().comms_debug_os1024_chunk_log2
x = torch.empty(n, dtype=torch.complex64, device="cuda:0")
y = torch.empty_like(x)
c.synth_iq_dev(x.data_ptr(), n, 0)
s = torch.cuda.current_stream().cuda_stream
fir = c.BatchFirNode(c.rrc_taps(255, 8.0, 0.35))
assert fir.kernel_for(n) == "fir_os1024_dyn_kernel"
stamper = c.KernelTimer(1 << 16, stamps=True)
variants = [(k, st) for k in (32, 1, 2, 3) for st in (False, True)]


def setup(k, st):
    set_chunk(k)
    if st:
        stamper.attach(fir)
    else:
        lib().comms_fir_set_timer(fir._h, None)


for _ in range(300):
    fir.run_dev(x.data_ptr(), n, y.data_ptr(), s)
torch.cuda.synchronize()
iso = {v: [] for v in variants}
bb = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        setup(*v)
        for _ in range(3):
            fir.run_dev(x.data_ptr(), n, y.data_ptr(), s)
        ev = []
        for _ in range(8):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fir.run_dev(x.data_ptr(), n, y.data_ptr(), s)
            b.record()
            ev.append((a, b))
        torch.cuda.synchronize()
        iso[v] += [a.elapsed_time(b) * 1e3 for a, b in ev]
        t0 = time.perf_counter()
        for _ in range(burst):
            fir.run_dev(x.data_ptr(), n, y.data_ptr(), s)
        torch.cuda.synchronize()
        bb[v].append((time.perf_counter() - t0) / burst * 1e6)
        if stamper.read_stamps_ms().size > (1 << 16) - 1024:
            stamper.reset()
print("n = 2^%d, %d rounds, bursts of %d launches; us per launch" % (lg, rounds, burst))
for v in variants:
    i, b = np.array(iso[v]), np.array(bb[v])
    print("dealing %-7s stamps %-3s: isolated median %7.2f  p10 %7.2f | back-to-back median %7.2f  min %7.2f" % (
        "contig" if v[0] == 32 else "2^%d" % v[0], "on" if v[1] else "off", np.median(i), np.percentile(i, 10), np.median(b), b.min()))
