#!/usr/bin/env python3
"""Write-only rate (fill of 2^k Complex<f32>) beside a copy: the floor of UpsampleNode's zero stores.
Usage: python scripts/probe_fill.py [log2 n = 24]   (diagnostic build)"""
import os as _os; _os.environ.setdefault("COMMS_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "comms_rs_amd", "lib", "libcomms_hip_diag.so"))
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import comms_rs_amd as c
l = c.lib()
f = l.comms_debug_copy; f.restype = C.c_int32; f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 24)
x = torch.empty(n, dtype=torch.complex64, device="cuda:0"); y = torch.empty_like(x)
c.synth_iq_dev(x.data_ptr(), n, 0)
for mode, nm, bytes_ in [(1, "copy float2", 16), (9, "fill float2", 8), (10, "fill float4", 8), (9, "fill float2", 8), (1, "copy float2", 16)]:
    for _ in range(10): f(x.data_ptr(), y.data_ptr(), n, mode, 0, None)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(100)]
    for a, b in ev:
        a.record(); f(x.data_ptr(), y.data_ptr(), n, mode, 0, None); b.record()
    torch.cuda.synchronize()
    ms = np.median([a.elapsed_time(b) for a, b in ev])
    print("%s: median %.1f us -> %.0f GB/s" % (nm, ms * 1e3, bytes_ * n / ms / 1e6))
up = c.UpsampleNode(4)
xi = x[: n // 4]
for _ in range(10): up.run_dev(xi.data_ptr(), n // 4, 8, y.data_ptr())
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(100)]
for a, b in ev:
    a.record(); up.run_dev(xi.data_ptr(), n // 4, 8, y.data_ptr()); b.record()
torch.cuda.synchronize()
ms = np.median([a.elapsed_time(b) for a, b in ev])
print("upsample x4 (%d outputs): median %.1f us -> %.0f GB/s" % (n, ms * 1e3, (8 * n + 2 * n) / ms / 1e6))
