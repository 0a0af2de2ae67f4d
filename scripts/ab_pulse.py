"""A/B timing of two builds of the library in ONE process, launch by launch: the pulse shaper with its fused mixer
(config 1's launch).  usage: python scripts/ab_pulse.py <libA.so> <libB.so> [n_taps] [sps] [log2 n_out] [reps] [mix]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import comms_rs_amd as c

paths = sys.argv[1:3]
n_taps = int(sys.argv[3]) if len(sys.argv) > 3 else 63
sps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
n_out = 1 << (int(sys.argv[5]) if len(sys.argv) > 5 else 24)
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 300
mix = int(sys.argv[7]) if len(sys.argv) > 7 else 1
n_sym = n_out // sps
x = torch.empty(n_sym, dtype=torch.complex64, device="cuda:0"); y = torch.empty(n_sym * sps, dtype=torch.complex64, device="cuda:0")
c.synth_iq_dev(x.data_ptr(), n_sym, 0)
taps = np.ascontiguousarray(c.rrc_taps(n_taps, float(sps), 0.25))
s = torch.cuda.current_stream().cuda_stream
hs = []
for p in paths:
    l = C.CDLL(os.path.abspath(p))
    h = C.c_void_p()
    l.comms_pulse_create.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int32, C.POINTER(C.c_void_p)]
    assert l.comms_pulse_create(taps.ctypes.data, taps.size, sps, 0, C.byref(h)) == 0
    if mix:
        l.comms_pulse_set_mixer.argtypes = [C.c_void_p, C.c_double, C.c_double]
        assert l.comms_pulse_set_mixer(h, 2 * np.pi * 0.1, 0.0) == 0
    l.comms_pulse_run_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    hs.append((l, h))
run = lambda i: hs[i][0].comms_pulse_run_dev(hs[i][1], x.data_ptr(), n_sym, y.data_ptr(), s)
for _ in range(30):
    assert run(0) == 0 and run(1) == 0
ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)] for _ in paths]
for r in range(reps):
    for i in range(len(paths)):
        a, b = ev[i][r]
        a.record(); run(i); b.record()
torch.cuda.synchronize()
for i, p in enumerate(paths):
    v = np.array([a.elapsed_time(b) for a, b in ev[i]]) * 1e3
    print("%-32s taps=%d sps=%d outputs=2^%d mix=%d: median %.2f us  mean %.2f  p10 %.2f  p90 %.2f (events around the launch)"
          % (os.path.basename(p), n_taps, sps, int(np.log2(n_out)), mix, np.median(v), v.mean(), *np.percentile(v, [10, 90])))
