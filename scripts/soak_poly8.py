"""Soak of fir_poly8_kernel: thousands of launches over a few batch lengths, rates and both store forms; every output buffer must
equal the first launch's bit for bit (segments are drawn in whatever order the waves arrive).  usage: python3 scripts/soak_poly8.py [seconds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import comms_rs_amd as c

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
s = torch.cuda.current_stream().cuda_stream
x = torch.empty(1 << 25, dtype=torch.complex64, device="cuda:0")
c.synth_iq_dev(x.data_ptr(), 1 << 25, 0, 5)
def lp(nt):
    k = np.arange(nt) - (nt - 1) / 2.0
    return (2 / 20 * np.sinc(2 / 20 * k) * np.hamming(nt)).astype(np.complex64)


# (rate, FM demod, log2 n, taps): 249 taps = four halo rows; 313 / 385 / 505 / 513 = five / six / eight
cases = []
for rate, fm, lg, nt in [(8, False, 24, 249), (8, True, 25, 249), (4, False, 23, 249), (4, True, 22, 249), (32, False, 24, 249), (8, True, 18, 249),
                         (8, False, 21, 249), (16, False, 20, 249), (12, False, 21, 249), (8, True, 23, 313), (8, False, 24, 385), (4, True, 22, 505),
                         (8, False, 20, 513), (20, False, 22, 400)]:
    taps = lp(nt)
    n = (1 << lg) - 8 * 64 * 3 + rate * 5 * 8
    n -= n % (rate * 8)
    ref_node = c.ChainNode(0.4, 0.1, taps, rate, fm, mixer_after_fir=not fm, kernel="poly")
    ref = torch.empty(n // rate, dtype=torch.float32 if fm else torch.complex64, device="cuda:0")
    ref_node.run_dev(x.data_ptr(), n, ref.data_ptr(), s)
    torch.cuda.synchronize()
    cases.append((rate, fm, n, ref, taps))
t0, launches, bad, rounds = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    for rate, fm, n, ref, taps in cases:
        node = c.ChainNode(0.4, 0.1, taps, rate, fm, mixer_after_fir=not fm, kernel="poly")  # fresh state: the same call as the reference's
        out = torch.empty_like(ref)
        for i in range(3):  # (calls 2 and 3 continue the stream: only the first is compared)
            node.run_dev(x.data_ptr(), n, out.data_ptr(), s)
            launches += 1
            if i == 0:
                torch.cuda.synchronize()
                if not torch.equal(torch.view_as_real(out) if not fm else out, torch.view_as_real(ref) if not fm else ref):
                    bad += 1
                    print("MISMATCH rate %d fm %d n %d" % (rate, fm, n), flush=True)
        torch.cuda.synchronize()
    rounds += 1
    if rounds % 100 == 0:
        print("%.0f s: %d launches, %d mismatches" % (time.time() - t0, launches, bad), flush=True)
print("soak done: %d launches in %.0f s, %d mismatches" % (launches, time.time() - t0, bad))
assert bad == 0
