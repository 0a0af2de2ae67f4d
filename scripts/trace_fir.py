"""Diagnostic: when do the waves of fir_os1024_kernel start, finish set-up and end?  (traced build,
production geometry: 256 workgroups of 16 waves, 255 taps, 2^24 samples).  s_memrealtime ticks at 100 MHz."""
import os as _os; _os.environ.setdefault("COMMS_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "comms_rs_amd", "lib", "libcomms_hip_diag.so"))  # diagnostic build (`make -C comms_rs_amd/csrc diag`)
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import comms_rs_amd as c
l = c.lib()
f = l.comms_fir_run_fused_dev; f.restype = C.c_int32
f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
x = torch.empty(n, dtype=torch.complex64, device="cuda:0"); y = torch.empty_like(x)
c.synth_iq_dev(x.data_ptr(), n, 0)
fir = c.BatchFirNode(c.rrc_taps(255, 8.0, 0.35))
slots = 16 * 256
dbg = torch.zeros(slots * 8, dtype=torch.int64, device="cuda:0")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
warm = int(os.environ.get("WARM", "3"))  # launches before the traced one (a long run shows the sustained clock)
for it in range(warm + 1):
    if it == warm: ev[0].record()
    st = f(fir._h, x.data_ptr(), n, y.data_ptr(), 32, 0, 0, 1, None, dbg.data_ptr(), None)
    assert st == 0, l.comms_last_error()
ev[1].record()
torch.cuda.synchronize()
print("event-timed launch: %.1f us" % (ev[0].elapsed_time(ev[1]) * 1e3))
d = dbg.cpu().numpy().reshape(slots, 8)
t0 = d[:, 0].min()
start, setup, end = [(d[:, i] - t0) / 100.0 for i in range(3)]  # us
cyc, hwid, xcc, segs = d[:, 3], d[:, 4], d[:, 5], d[:, 6]
life = end - start
print("kernel span (first wave start -> last wave end): %.1f us" % end.max())
print("shader clock from s_memtime / s_memrealtime: %.3f GHz" % np.median(cyc / (life * 1e3)))
q = lambda v: "min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" % (v.min(), *np.percentile(v, [10, 50, 90]), v.max())
print("wave start   [us]:", q(start))
print("set-up done  [us]:", q(setup))
print("set-up time  [us]:", q(setup - start))
print("wave end     [us]:", q(end))
print("lifetime     [us]:", q(life))
for k in sorted(set(segs)):
    m = segs == k
    print("  waves with %d segments: %4d   compute time %s" % (k, m.sum(), q((end - setup)[m])))
    if k: print("     per segment [us]: %.2f" % np.median((end - setup)[m] / k))
rank = (np.arange(slots) % 16) // 4
print("median end by launch rank on the SIMD (waves 0-3, 4-7, 8-11, 12-15): " + "  ".join("%.1f" % np.median(end[rank == r]) for r in range(4)))
print("segments by rank: " + "  ".join("%.2f" % segs[rank == r].mean() for r in range(4)))
print("mean residency: %.1f%% of the span" % (100 * life.sum() / (slots * end.max())))
cu = (hwid >> 8) & 0xF; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 0x7; simd = (hwid >> 4) & 0x3
place = xcc * 1000 + se * 100 + sh * 50 + cu
wg = np.arange(slots) // 16
per_cu = {}
for w in range(256): per_cu.setdefault(int(place[w * 16]), []).append(w)
print("distinct (xcc,se,sh,cu) hosting a workgroup: %d; most on one: %d" % (len(per_cu), max(len(v) for v in per_cu.values())))
same = sum(1 for w in range(256) if len(set(place[w * 16:(w + 1) * 16])) == 1)
print("workgroups whose 16 waves share one CU: %d; waves per SIMD in workgroup 0: %s" % (same, np.bincount(simd[:16], minlength=4)))
if len(sys.argv) > 2:
    np.save(sys.argv[2], d)
