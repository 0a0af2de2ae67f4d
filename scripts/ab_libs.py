"""A/B timing of two or more builds of the library in ONE process, launch by launch (FIR node; ALGO=os1024|os4096|os16k|direct|auto).
usage: python scripts/ab_libs.py <libA.so> <libB.so> [more.so ...] [n_taps] [log2 n] [reps]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import comms_rs_amd as c  # synthetic input + taps from the default build

paths = [a for a in sys.argv[1:] if a.endswith(".so")]
nums = [a for a in sys.argv[1:] if not a.endswith(".so")]
n_taps = int(nums[0]) if len(nums) > 0 else 255
n = 1 << (int(nums[1]) if len(nums) > 1 else 24)
reps = int(nums[2]) if len(nums) > 2 else 300
x = torch.empty(n, dtype=torch.complex64, device="cuda:0"); y = torch.empty_like(x)
c.synth_iq_dev(x.data_ptr(), n, 0)
taps = np.ascontiguousarray(c.rrc_taps(n_taps, 8.0, 0.35))
s = torch.cuda.current_stream().cuda_stream
hs = []
for p in paths:
    l = C.CDLL(os.path.abspath(p))
    h = C.c_void_p()
    l.comms_fir_create.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int32, C.POINTER(C.c_void_p)]
    assert l.comms_fir_create(taps.ctypes.data, taps.size, None, 0, 0, C.byref(h)) == 0
    l.comms_fir_set_algo.argtypes = [C.c_void_p, C.c_int32]
    assert l.comms_fir_set_algo(h, {"os1024": c.FIR_OS1024, "os4096": c.FIR_OS4096, "os16k": c.FIR_OS16K, "auto": c.FIR_AUTO, "direct": c.FIR_DIRECT}[os.environ.get("ALGO", "os1024")]) == 0
    l.comms_fir_run_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    hs.append((l, h))
run = lambda i: hs[i][0].comms_fir_run_dev(hs[i][1], x.data_ptr(), n, y.data_ptr(), s)
for _ in range(20):
    for i in range(len(paths)):
        run(i)
ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)] for _ in paths]
for r in range(reps):
    for i in range(len(paths)):
        a, b = ev[i][r]
        a.record(); run(i); b.record()
torch.cuda.synchronize()
for i, p in enumerate(paths):
    v = np.array([a.elapsed_time(b) for a, b in ev[i]]) * 1e3
    print("%-40s taps=%d n=2^%d: median %.2f us  mean %.2f  p10 %.2f  p90 %.2f (events around the launch)"
          % (os.path.basename(p), n_taps, int(np.log2(n)), np.median(v), v.mean(), *np.percentile(v, [10, 90])))
