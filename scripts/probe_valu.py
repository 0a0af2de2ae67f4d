import os as _os; _os.environ.setdefault("COMMS_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "comms_rs_amd", "lib", "libcomms_hip_diag.so"))  # diagnostic build (`make -C comms_rs_amd/csrc diag`)
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import comms_rs_amd as c
l = c.lib()
f = l.comms_debug_valu; f.restype = C.c_int32; f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
out = torch.empty(256 * 8 * 256, dtype=torch.float32, device="cuda:0")
iters = 20000
KINDS = [int(k) for k in os.environ.get("KINDS", "0,3,4,5,6,7,8,9,10,11,12,40,41,42,43").split(",")]
for kind, nm in [(k, n) for k, n in [(0, "v_fma_f32"), (3, "v_pk_fma_f32"), (7, "MAC pk_fma acc+=u*SGPRpair, 2 chains"), (8, "MAC pk_fma acc+=u*VGPRpair, 2 chains"),
                 (20, "form0 acc=u*tV+acc plain"), (21, "form1 +op_sel_hi[1,0,1] on tap"), (22, "form2 +op_sel hi-broadcast on tap"),
                 (23, "form3 acc=acc*tV+u (dst==src0) plain"), (24, "form4 dst==src0 + op_sel_hi[1,0,1]"), (25, "form5 plain, SGPR tap"),
                 (26, "form6 2x v_fma_f32 SGPR tap (2 instr)"), (27, "form7 pk_mul(op_sel)+pk_add (2 instr)"), (28, "form8 acc=u*u+acc (2 VGPR pairs)"),
                 (40, "v_fma_f64"), (41, "v_add_f64"), (42, "v_mul_f64"), (43, "v_fmac_f64 SGPR factor"),
                 (29, "form9 pk_mul only"), (30, "form10 lo/hi alternating, same VGPR pair twice"), (31, "form11 lo/hi alternating, new pair each"),
                 (32, "form12 lo,lo same pair twice"), (33, "form13 lo/hi alternating SGPR pair"),
                 (9, "MAC SGPR taps, 16 chains"), (10, "MAC VGPR taps, 16 chains"), (11, "MAC SGPR taps, 4 chains"), (12, "MAC VGPR taps, 4 chains"), (4, "cmul=pk_mul+pk_fma (2 instr)"), (5, "v_permlane32_swap (8 per 16 slots)"), (6, "v_permlane16_swap (8 per 16 slots)")] if k in KINDS]:
    for wps in [int(w) for w in os.environ.get('WPS', '1,2,4,8').split(',')]:   # waves per SIMD
        blocks = 256 * wps
        f(out.data_ptr(), kind, 100, blocks, None)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(out.data_ptr(), kind, iters, blocks, None); b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b)
        instr_per_simd = iters * 16 * wps * (2 if kind in (4, 26, 27) else 0.5 if kind in (5, 6) else 1)
        print("%s waves/SIMD=%d: %.2f ms -> %.2f ns per wave-instr per SIMD = %.2f cycles @2.4GHz" %
              (nm, wps, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4))

g = l.comms_debug_swap_semantics; g.restype = C.c_int32; g.argtypes = [C.c_void_p, C.c_void_p]
o = torch.zeros(256, dtype=torch.int32, device="cuda:0")
g(o.data_ptr(), None); torch.cuda.synchronize()
o = o.cpu().numpy()
print("v_permlane32_swap a:", o[:64].tolist()); print("                  b:", o[64:128].tolist())
print("v_permlane16_swap a:", o[128:192].tolist()); print("                  b:", o[192:].tolist())
