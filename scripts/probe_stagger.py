import os as _os; _os.environ.setdefault("COMMS_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "comms_rs_amd", "lib", "libcomms_hip_diag.so"))  # diagnostic build (`make -C comms_rs_amd/csrc diag`)
"""Wave-private read streams (comms_debug_read mode 203: nontemporal loads + the FM chain's store), 16 tiles of 8 KiB per
wave, every wave starting at tile 0 of its run (all waves at the same phase: requests 128 KiB apart) against waves that
start at different tiles of their run and wrap."""
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
plan = []
for rep in range(2):
    for ntc in (16, 4):
        for mul in (0, 1, 3, 5, 7):
            plan.append((203, ((mul * 16 + 1) << 12) + ntc * 16 + 4, "runs of %d tiles, nt loads, nt store, %s" % (ntc, "all waves in phase" if mul == 0 else "wave w starts at tile (%d w) %% %d" % (mul, ntc))))
for mode, wg, name in plan:
    for _ in range(5):
        assert f(x.data_ptr(), n, mode, wg, sink.data_ptr(), out.data_ptr(), None) == 0
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
    for a, b in ev:
        a.record(); f(x.data_ptr(), n, mode, wg, sink.data_ptr(), out.data_ptr(), None); b.record()
    torch.cuda.synchronize()
    ms = np.array([a.elapsed_time(b) for a, b in ev])
    print("%-80s median %.1f us -> %.0f GB/s read" % (name, np.median(ms) * 1e3, 8 * n / np.median(ms) / 1e6), flush=True)
