"""Soak of the chain kernels added in round 5's second half: the decimating 4096- / 16384-point kernels and the wave-private kernel at
rates 2 and 4 -- every first launch of a fresh node must equal the reference launch bit for bit.  usage: python3 scripts/soak_chains.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import comms_rs_amd as c

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
s = torch.cuda.current_stream().cuda_stream
x = torch.empty(1 << 24, dtype=torch.complex64, device="cuda:0")
c.synth_iq_dev(x.data_ptr(), 1 << 24, 0, 9)


def lp(nt, rate):
    k = np.arange(nt) - (nt - 1) / 2.0
    return (0.8 / rate * np.sinc(0.8 / rate * k) * np.hamming(nt)).astype(np.complex64)


cases = []
for rate, fm, n_out, nt in [(5, False, 1 << 21, 769), (7, True, 1 << 20, 1100), (5, False, 1 << 21, 2500), (16, True, 1 << 19, 4097), (2, False, 128 * 8192, 31),
                            (2, True, 128 * 8000 - 5, 63), (4, False, 128 * 8192, 100), (3, False, 1 << 21, 300), (100, False, 1 << 17, 600)]:
    n = rate * n_out
    taps = lp(nt, rate)
    mk = lambda: c.ChainNode(0.4, 0.1, taps, rate, fm, mixer_after_fir=not fm)
    ref = torch.empty(n_out, dtype=torch.float32 if fm else torch.complex64, device="cuda:0")
    node = mk()
    node.run_dev(x.data_ptr(), n, ref.data_ptr(), s)
    torch.cuda.synchronize()
    cases.append((rate, fm, n, ref, taps, node.kernel))
print("cases:", [(r, f, k) for r, f, _, _, _, k in cases], flush=True)
t0, launches, bad, rounds = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    for rate, fm, n, ref, taps, _ in cases:
        node = c.ChainNode(0.4, 0.1, taps, rate, fm, mixer_after_fir=not fm)
        out = torch.empty_like(ref)
        for i in range(3):
            node.run_dev(x.data_ptr(), n, out.data_ptr(), s)
            launches += 1
            if i == 0:
                torch.cuda.synchronize()
                if not torch.equal(out if fm else torch.view_as_real(out), ref if fm else torch.view_as_real(ref)):
                    bad += 1
                    print("MISMATCH rate %d fm %d n %d" % (rate, fm, n), flush=True)
        torch.cuda.synchronize()
    rounds += 1
    if rounds % 200 == 0:
        print("%.0f s: %d launches, %d mismatches" % (time.time() - t0, launches, bad), flush=True)
print("soak done: %d launches in %.0f s, %d mismatches" % (launches, time.time() - t0, bad))
assert bad == 0
