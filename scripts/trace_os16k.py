"""Where does a segment of fir_os16k_kernel spend its time?  The diagnostic build stamps s_memtime at eight points of
every segment (first 40 segments of every wave); this prints the phases per wave position and the shape of the two
waits.  4097 taps, 2^27 samples by default (BASELINE config 5).
usage: python scripts/trace_os16k.py [log2 n] [n_taps] [save.npy]
MODES=0,1,2 (environment): wave-priority trial schemes (comms_debug_os16k_fault bits 1...), timed interleaved first"""
import os as _os; _os.environ.setdefault("COMMS_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "comms_rs_amd", "lib", "libcomms_hip_diag.so"))  # diagnostic build (`make -C comms_rs_amd/csrc diag`)
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import comms_rs_amd as c

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 27
n_taps = int(sys.argv[2]) if len(sys.argv) > 2 else 4097
n = 1 << lg
l = c.lib()
l.comms_debug_os16k_trace.restype = C.c_uint
l.comms_debug_os16k_trace.argtypes = [C.c_void_p]
x = torch.empty(n, dtype=torch.complex64, device="cuda:0")
y = torch.empty_like(x)
c.synth_iq_dev(x.data_ptr(), n, 0)
s = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(5)
fir = c.BatchFirNode((rng.standard_normal(n_taps) + 1j * rng.standard_normal(n_taps)).astype(np.complex64) / n_taps)
assert fir.kernel_for(n) == "fir_os16k_kernel", fir.kernel_for(n)
G = 256
for _ in range(10):
    fir.run_dev(x.data_ptr(), n, y.data_ptr(), s)
torch.cuda.synchronize()
modes = [int(m) for m in os.environ.get("MODES", "0").split(",")]
if len(modes) > 1:
    tm = {m: [] for m in modes}
    for r in range(8):
        for m in modes:
            l.comms_debug_os16k_fault(m << 1)
            fir.run_dev(x.data_ptr(), n, y.data_ptr(), s)
            e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(4)]
            for a, b in e:
                a.record()
                fir.run_dev(x.data_ptr(), n, y.data_ptr(), s)
                b.record()
            torch.cuda.synchronize()
            tm[m] += [a.elapsed_time(b) * 1e3 for a, b in e]
    for m in modes:
        v = np.array(tm[m])
        print("priority scheme %d: median %.1f us  min %.1f  p90 %.1f  (%d launches, interleaved)" % (m, np.median(v), v.min(), np.percentile(v, 90), v.size))
    l.comms_debug_os16k_fault(int(os.environ.get("TRACE_MODE", str(modes[-1]))) << 1)
    print("traced below: scheme %s" % os.environ.get("TRACE_MODE", str(modes[-1])))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
ev[0].record()
fir.run_dev(x.data_ptr(), n, y.data_ptr(), s)
ev[1].record()
torch.cuda.synchronize()
segs = 40
buf = torch.zeros(G * 16 * segs * 16, dtype=torch.int64, device="cuda:0")
got = l.comms_debug_os16k_trace(buf.data_ptr())
assert got == segs, got
ev[2].record()
fir.run_dev(x.data_ptr(), n, y.data_ptr(), s)
ev[3].record()
torch.cuda.synchronize()
l.comms_debug_os16k_trace(None)
l.comms_debug_os16k_fault(0)
print("launch untraced %.1f us, traced %.1f us" % (ev[0].elapsed_time(ev[1]) * 1e3, ev[2].elapsed_time(ev[3]) * 1e3))
raw = buf.cpu().numpy().reshape(G, 16, segs, 16).astype(np.float64)
# marks in program order: 0 segment start, 1 rows arrived, 8 radix-16 done, 2 slices written + signalled, 3 slice_in passed,
# 4 slice done + signalled, 5 next rows requested, 6 slice_out passed, 9 slice points read, 10 radix-16 back done, 7 stores issued
order = [0, 1, 8, 2, 3, 4, 5, 6, 9, 10, 7]
d = raw[..., order]
ok = d[:, 0, :, 0] > 0
print("workgroups that wrote a trace: %d, segments traced per workgroup: %s" % (ok[:, 0].sum(), sorted(set(ok.sum(axis=1)))[:4]))
# the shader clock: the traced launch's span in s_memtime ticks against its event time is not known per wave; print ticks
# and, from the launch time, what a tick is if the workgroups ran the whole launch
per_wg = (n // 12288 + 1) / G
span = d[:, :, 1:, 0] - d[:, :, :-1, 0]  # start of a segment to the start of the next: [G][16][segs-1]
seg_ticks = np.median(span[:, :, 4:])
tick_ns = (ev[2].elapsed_time(ev[3]) * 1e6 / per_wg) / seg_ticks
print("a segment: median %.0f ticks start to start; with %.1f segments per workgroup in the launch a tick is ~%.3f ns (%.2f GHz)" % (
    seg_ticks, per_wg, tick_ns, 1 / tick_ns))
names = ["rows arrive (load wait)", "radix-16", "16 slice writes + signal", "wait slice_in", "slice work (FFT, H, IFFT)", "request next rows",
         "wait slice_out", "16 LDS reads arrive", "radix-16 back", "12 stores issued"]
NP = len(names)
ph = d[:, :, 4:, 1:] - d[:, :, 4:, :-1]  # [G][16][segs-4][NP]
tot = d[:, :, 4:, NP] - d[:, :, 4:, 0]
print("\nphase                       median   mean    p10    p90   (ticks; share of the segment's mean)")
for i, nm in enumerate(names):
    v = ph[..., i].ravel()
    print("%-26s %7.0f %7.0f %6.0f %6.0f   %5.1f %%" % (nm, np.median(v), v.mean(), *np.percentile(v, [10, 90]), 100 * v.mean() / tot.mean()))
print("%-26s %7.0f %7.0f" % ("segment (mark 0 -> 7)", np.median(tot), tot.mean()))
print("\nby wave of the workgroup (mean ticks), the phases above in order")
for w in range(16):
    print("  wave %2d: " % w + " ".join("%6.0f" % ph[:, w, :, i].mean() for i in range(NP)))
# how far apart are the sixteen waves when they reach the marks?
for i, nm in ((3, "stage 1 done (signal slice_in)"), (5, "slice done (signal slice_out)"), (NP, "stores issued")):
    t = d[:, :, 4:, i]
    spread = t.max(axis=1) - t.min(axis=1)
    print("spread of the sixteen waves at '%s': median %.0f ticks, p90 %.0f" % (nm, np.median(spread), np.percentile(spread, 90)))
# order of arrival by SIMD slot: waves w, w+4, w+8, w+12 usually share a SIMD
t4 = d[:, :, 4:, 5]
rank = np.argsort(np.argsort(t4, axis=1), axis=1)
print("mean finishing rank of the slice work by wave: " + " ".join("%.1f" % rank[:, w].mean() for w in range(16)))
g = 0
for sgi in (10, 11):
    t = d[g, :, sgi, :]
    t0 = t[:, 0].min()
    print("workgroup %d, segment %d: marks in program order, ticks after the first wave's start" % (g, sgi))
    for w in range(16):
        print("  wave %2d " % w + " ".join("%6d" % (x - t0) for x in t[w]))
if len(sys.argv) > 3:
    np.save(sys.argv[3], d)
