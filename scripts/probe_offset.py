"""Does the headline FIR kernel's HBM-resident rate depend on where its output buffer sits relative to its input?
Both streams advance together (a segment's rows are read and its outputs written by the same wave), so bases that are
congruent modulo the memory system's interleaving period put every read and its write on the same channel / bank group.
usage: python scripts/probe_offset.py [log2 n] [n_taps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import comms_rs_amd as c

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 28
n_taps = int(sys.argv[2]) if len(sys.argv) > 2 else 255
n = 1 << lg
PAD = 1 << 26  # bytes of slack behind the output buffer
dev = torch.device("cuda", 0)
x = torch.empty(n, dtype=torch.complex64, device=dev)
ybig = torch.empty(n + PAD // 8, dtype=torch.complex64, device=dev)
c.synth_iq_dev(x.data_ptr(), n, 0)
s = torch.cuda.current_stream().cuda_stream
fir = c.BatchFirNode(c.rrc_taps(n_taps, 8.0, 0.35))
print("x at 0x%x, y at 0x%x: difference 0x%x (mod 2^32: 0x%x)" % (x.data_ptr(), ybig.data_ptr(), ybig.data_ptr() - x.data_ptr(),
                                                               (ybig.data_ptr() - x.data_ptr()) % (1 << 32)))
print("kernel:", fir.kernel_for(n))


def time_at(off_bytes, reps=7):
    yp = ybig.data_ptr() + off_bytes
    for _ in range(2):
        fir.run_dev(x.data_ptr(), n, yp, s)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fir.run_dev(x.data_ptr(), n, yp, s)
        b.record()
    torch.cuda.synchronize()
    v = np.array([a.elapsed_time(b) for a, b in ev])
    return float(np.median(v)), float(v.min())


# a long warm-up: the chip's clocks settle over the first tens of milliseconds of a burst
for _ in range(10):
    fir.run_dev(x.data_ptr(), n, ybig.data_ptr(), s)
torch.cuda.synchronize()
offs = [0, 8, 64, 256, 1024, 4096, 8192, 16384, 32768, 65536, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 21, 1 << 22, 1 << 23,
        1 << 24, 1 << 25, 4096 + 256, (1 << 20) + 4096, (1 << 21) + 8192 + 512, 3 << 19, 5 << 18, 0]
for o in offs:
    med, mn = time_at(o)
    print("offset %10d B (0x%08x): median %8.3f ms  min %8.3f ms  = %6.0f GB/s algorithmic" % (o, o, med, mn, 16.0 * n / med / 1e6))
