"""Diagnostic: per-phase cycle shares of one wave of fir_os1024_kernel (stamped build)."""
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
runs = 12 * 256
dbg = torch.zeros(runs * 10, dtype=torch.int64, device="cuda:0")
for _ in range(3):
    st = f(fir._h, x.data_ptr(), n, y.data_ptr(), 16, 0, 0, 1, None, dbg.data_ptr(), None)
    assert st == 0, l.comms_last_error()
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(runs, 10).astype(np.float64)
segs = 21846 / runs
names = ["global loads landed", "R16+tw+E1 writes", "E1 reads", "R16+tw (stage 2)", "R4 across lanes", "H + inverse R4 across lanes",
         "conj tw (stage 2)", "R16+E4 writes", "E4 reads+tw", "R16+stores retired"]
tot = d.sum(1).mean()
print("cycles per segment per wave: %.0f (s_memtime ticks)" % (tot / segs))
for i, nm in enumerate(names):
    print("  %-22s %7.0f  %5.1f%%" % (nm, d[:, i].mean() / segs, 100 * d[:, i].mean() / tot))
