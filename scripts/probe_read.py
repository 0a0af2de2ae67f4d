import os as _os; _os.environ.setdefault("COMMS_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "comms_rs_amd", "lib", "libcomms_hip_diag.so"))  # diagnostic build (`make -C comms_rs_amd/csrc diag`)
"""Read-only probes (comms_debug_read, diagnostic build): what the decimating chain's input side reaches without its
arithmetic.  usage: python scripts/probe_read.py [log2 n]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import comms_rs_amd as c
l = c.lib()
f = l.comms_debug_read; f.restype = C.c_int32; f.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 26)
x = torch.empty(n, dtype=torch.complex64, device="cuda:0")
c.synth_iq_dev(x.data_ptr(), n, 0)
sink = torch.zeros(16, dtype=torch.float32, device="cuda:0")
out = torch.zeros(n // 8, dtype=torch.float32, device="cuda:0")
names = {0: "grid-stride float4", 1: "grid-stride float2"}
for fl in range(32):
    names[100 + fl] = "tiles" + (" +prefetch" if fl & 1 else "") + (" +LDS,2 barriers" if fl & 2 else "") + (" +nt" if fl & 4 else "") + (" 16B/lane" if fl & 8 else "") + (" +store n/16" if fl & 16 else "")
plan = [(0, 0), (1, 0)]
for wg in (4, 3, 2, 6, 8):
    for fl in (0, 1, 2, 3, 4, 5, 8, 9, 12, 13, 16, 17, 18, 19):
        if wg in (6, 8) and fl & 2:
            continue  # (37 KiB of LDS per workgroup: four per CU at most)
        plan.append((100 + fl, wg))
plan += [(0, 0), (1, 0)]
for mode, wg in plan:
    for _ in range(5):
        assert f(x.data_ptr(), n, mode, wg, sink.data_ptr(), out.data_ptr(), None) == 0
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
    for a, b in ev:
        a.record(); f(x.data_ptr(), n, mode, wg, sink.data_ptr(), out.data_ptr(), None); b.record()
    torch.cuda.synchronize()
    ms = np.array([a.elapsed_time(b) for a, b in ev])
    print("mode %3d %-48s wgs/CU %d: median %.1f us -> %.0f GB/s read" % (mode, names[mode], wg, np.median(ms) * 1e3, 8 * n / np.median(ms) / 1e6), flush=True)
