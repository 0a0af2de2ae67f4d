"""A/B timing of fir_os1024 variants, interleaved launch by launch so that clock and box drift cancel.
usage: python scripts/ab_fir.py [n_taps] [log2 n] [reps]   (modes: 0 fixed runs, 1 ticketed segments)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import comms_rs_amd as c
n_taps = int(sys.argv[1]) if len(sys.argv) > 1 else 255
n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 24)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
modes = [int(m) for m in os.environ.get("MODES", "0,1").split(",")]
x = torch.empty(n, dtype=torch.complex64, device="cuda:0"); y = torch.empty_like(x)
c.synth_iq_dev(x.data_ptr(), n, 0)
firs = {0: c.BatchFirNode(c.rrc_taps(n_taps, 8.0, 0.35)).set_algo(c.FIR_OS1024_FIXED),
        1: c.BatchFirNode(c.rrc_taps(n_taps, 8.0, 0.35)).set_algo(c.FIR_OS1024)}
tm = c.KernelTimer(reps * len(modes))
s = torch.cuda.current_stream().cuda_stream
for _ in range(30):
    for m in modes:
        firs[m].run_dev(x.data_ptr(), n, y.data_ptr(), s)
for m in modes:
    c.lib().comms_fir_set_timer(firs[m]._h, tm._h)   # one event pool shared by both nodes, in launch order
for _ in range(reps):
    for m in modes:
        firs[m].run_dev(x.data_ptr(), n, y.data_ptr(), s)
ms = tm.read_ms().reshape(reps, len(modes))
for m in modes:
    c.lib().comms_fir_set_timer(firs[m]._h, None)
tm.close()
for i, m in enumerate(modes):
    v = ms[:, i] * 1e3
    print("mode %d  taps=%d n=2^%d: median %.2f us  mean %.2f  min %.2f  p10 %.2f  p90 %.2f" % (m, n_taps, int(np.log2(n)), np.median(v), v.mean(), v.min(), *np.percentile(v, [10, 90])))
