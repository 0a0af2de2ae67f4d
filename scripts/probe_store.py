import os as _os; _os.environ.setdefault("COMMS_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "comms_rs_amd", "lib", "libcomms_hip_diag.so"))  # diagnostic build (`make -C comms_rs_amd/csrc diag`)
"""What the FM chain's small output stream (1/16 of the input bytes) costs next to a read stream, by store form
(comms_debug_read modes 203: wave-private tiles, nontemporal loads, 512 B stored per 8 KiB read)."""
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
names = ["plain store", "nontemporal store", "sc0 sc1 store", "sc1 store", "sc0 store", "store into a 512-KiB window", "eight tiles' stores in one burst"]
plan = [(201, 4 * 16 + 4, "no store")]
for rep in range(2):
    for stm in range(7):
        for mode in (203, 202):
            plan.append((mode, (stm << 12) + 4 * 16 + 4, names[stm] + (" (nt loads)" if mode == 203 else " (plain loads)")))
plan.append((201, 4 * 16 + 4, "no store"))
for mode, wg, name in plan:
    for _ in range(5):
        assert f(x.data_ptr(), n, mode, wg, sink.data_ptr(), out.data_ptr(), None) == 0
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
    for a, b in ev:
        a.record(); f(x.data_ptr(), n, mode, wg, sink.data_ptr(), out.data_ptr(), None); b.record()
    torch.cuda.synchronize()
    ms = np.array([a.elapsed_time(b) for a, b in ev])
    print("mode %3d %-56s median %.1f us -> %.0f GB/s read" % (mode, name, np.median(ms) * 1e3, 8 * n / np.median(ms) / 1e6), flush=True)
