"""The ticketed FIR kernel's segment dealing (contiguous shares against round-robin chunks of 2^k segments), variants
interleaved launch by launch in one process; at several placements of the output buffer relative to the input.
usage: python scripts/with_lib.py comms_rs_amd/lib/libcomms_hip_diag.so scripts/probe_chunks.py [log2 n] [reps] [n_taps]
(the diagnostic build: comms_debug_os1024_chunk_log2 switches the dealing at run time)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import comms_rs_amd as c
from comms_rs_amd._lib import lib

set_chunk = lib().comms_debug_os1024_chunk_log2

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 28
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 9
n_taps = int(sys.argv[3]) if len(sys.argv) > 3 else 255
n = 1 << lg
dev = torch.device("cuda", 0)
x = torch.empty(n, dtype=torch.complex64, device=dev)
ybig = torch.empty(n + (1 << 23), dtype=torch.complex64, device=dev)
c.synth_iq_dev(x.data_ptr(), n, 0)
s = torch.cuda.current_stream().cuda_stream
fir = c.BatchFirNode(c.rrc_taps(n_taps, 8.0, 0.35))
print("n = 2^%d, %d taps, kernel %s; x at 0x%x, y at 0x%x" % (lg, n_taps, fir.kernel_for(n), x.data_ptr(), ybig.data_ptr()))
variants = [32, 0, 1, 2, 3, 4, 5, 6]
offs = [0, 1 << 20, (1 << 22) + 4096]
for _ in range(10):
    fir.run_dev(x.data_ptr(), n, ybig.data_ptr(), s)
torch.cuda.synchronize()
ref = None
for off in offs:
    yp = ybig.data_ptr() + off
    ev = {v: [] for v in variants}
    for r in range(reps + 2):
        for v in variants:
            set_chunk(v)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fir.run_dev(x.data_ptr(), n, yp, s)
            b.record()
            if r >= 2:
                ev[v].append((a, b))
    torch.cuda.synchronize()
    line = []
    for v in variants:
        t = np.array([a.elapsed_time(b) for a, b in ev[v]])
        line.append("%s %.4f" % ("contig" if v == 32 else "2^%d" % v, np.median(t)))
    print("y offset 0x%07x: median ms  " % off + " | ".join(line))
# results must not depend on the dealing
set_chunk(32)
y0 = torch.empty(1 << 22, dtype=torch.complex64, device=dev)
m = min(n, 1 << 22)
f2 = c.BatchFirNode(c.rrc_taps(n_taps, 8.0, 0.35))
f2.run_dev(x.data_ptr(), m, y0.data_ptr(), s)
for v in (0, 3, 5):
    set_chunk(v)
    f3 = c.BatchFirNode(c.rrc_taps(n_taps, 8.0, 0.35))
    y1 = torch.empty_like(y0)
    f3.run_dev(x.data_ptr(), m, y1.data_ptr(), s)
    torch.cuda.synchronize()
    print("chunks 2^%d vs contiguous on 2^22 samples: bit-identical %s" % (v, bool(torch.equal(y0, y1))))
