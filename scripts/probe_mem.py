import os as _os; _os.environ.setdefault("COMMS_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "comms_rs_amd", "lib", "libcomms_hip_diag.so"))  # diagnostic build (`make -C comms_rs_amd/csrc diag`)
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import comms_rs_amd as c
l = c.lib()
f = l.comms_debug_copy; f.restype = C.c_int32; f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 24)
x = torch.empty(n, dtype=torch.complex64, device="cuda:0"); y = torch.empty_like(x)
c.synth_iq_dev(x.data_ptr(), n, 0)
for mode, wpc in [(0, 0), (1, 0), (2, 8), (2, 12), (2, 16), (2, 32), (3, 8), (3, 16), (3, 32), (4, 256), (4, 512), (5, 256), (5, 512), (2, 16), (6, 16), (7, 16), (8, 16), (2, 16), (8, 16),
                  (1, 0), (101, 16), (101, 32), (102, 16), (104, 16), (112, 16), (112, 32), (116, 16), (2, 16), (1, 0)]:
    for _ in range(10): f(x.data_ptr(), y.data_ptr(), n, mode, wpc, None)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(100)]
    for a, b in ev:
        a.record(); f(x.data_ptr(), y.data_ptr(), n, mode, wpc, None); b.record()
    torch.cuda.synchronize()
    ms = np.array([a.elapsed_time(b) for a, b in ev])
    print("mode %d waves/CU %2d: median %.1f us -> %.0f GB/s" % (mode, wpc, np.median(ms) * 1e3, 2 * 8 * n / np.median(ms) / 1e6))
