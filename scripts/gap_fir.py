import os as _os; _os.environ.setdefault("COMMS_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "comms_rs_amd", "lib", "libcomms_hip_diag.so"))  # diagnostic build (`make -C comms_rs_amd/csrc diag`)
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import comms_rs_amd as c
l = c.lib()
f = l.comms_fir_run_fused_dev; f.restype = C.c_int32
f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
n = 1 << 24
x = torch.empty(n, dtype=torch.complex64, device="cuda:0"); y = torch.empty_like(x)
c.synth_iq_dev(x.data_ptr(), n, 0)
fir = c.BatchFirNode(c.rrc_taps(255, 8.0, 0.35))
slots = 16 * 256
bufs = [torch.zeros(slots * 8, dtype=torch.int64, device="cuda:0") for _ in range(6)]
for it in range(3):
    for b in bufs:
        assert f(fir._h, x.data_ptr(), n, y.data_ptr(), 32, 0, 0, 1, None, b.data_ptr(), None) == 0
torch.cuda.synchronize()
d = [b.cpu().numpy().reshape(slots, 8) for b in bufs]
for i in range(1, 6):
    gap = (d[i][:, 0].min() - d[i - 1][:, 2].max()) / 100.0
    span = (d[i][:, 2].max() - d[i][:, 0].min()) / 100.0
    print("launch %d: span %.1f us, gap after the previous launch's last wave %.1f us" % (i, span, gap))
