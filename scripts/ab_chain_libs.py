"""A/B of builds of the library on one fused chain, bursts of 20 launches, the builds taking turns.
usage: python3 scripts/ab_chain_libs.py <libA.so> <libB.so> ... [log2 n] [taps] [rate] [fm]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import comms_rs_amd as c

paths = [a for a in sys.argv[1:] if a.endswith(".so")]
nums = [int(a) for a in sys.argv[1:] if not a.endswith(".so")]
lg, nt, rate, fm = (nums + [24, 255, 8, 0][len(nums):])[:4]
n = 1 << lg
k = np.arange(nt) - (nt - 1) / 2.0
taps = np.ascontiguousarray((2 / 16 * np.sinc(2 / 16 * k) * np.hamming(nt)).astype(np.complex64))
x = torch.empty(n, dtype=torch.complex64, device="cuda:0")
c.synth_iq_dev(x.data_ptr(), n, 0)
s = torch.cuda.current_stream().cuda_stream
hs, outs = [], []
for p in paths:
    l = C.CDLL(os.path.abspath(p))
    h = C.c_void_p()
    l.comms_chain_create_ex.argtypes = [C.c_double, C.c_double, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    assert l.comms_chain_create_ex(2 * np.pi * 0.05, 0.0, taps.ctypes.data, taps.size, rate, (1 if fm else 2) | 32, 0, C.byref(h)) == 0
    l.comms_chain_run_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    hs.append((l, h))
    outs.append(torch.empty(n // rate, dtype=torch.float32 if fm else torch.complex64, device="cuda:0"))
run = lambda i: hs[i][0].comms_chain_run_dev(hs[i][1], x.data_ptr(), n, outs[i].data_ptr(), s)
for i in range(len(paths)):
    for _ in range(5):
        assert run(i) == 0
torch.cuda.synchronize()
ts = [[] for _ in paths]
for rep in range(10):
    for i in range(len(paths)):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            run(i)
        b.record()
        torch.cuda.synchronize()
        ts[i].append(a.elapsed_time(b) / 20 * 1e3)
for i, p in enumerate(paths):
    print("%-32s 2^%d taps %d rate %d fm %d: median %.2f us  (%s)" % (os.path.basename(p), lg, nt, rate, fm, np.median(ts[i][2:]), " ".join("%.1f" % t for t in ts[i])), flush=True)
