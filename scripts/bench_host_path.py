"""PCIe-inclusive rate of the host-pointer entry points (what a Vec<Complex<f32>> node pays)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import comms_rs_amd as c
n = 1 << 24
x = c.synth_iq(n)
fir = c.BatchFirNode(c.rrc_taps(255, 8.0, 0.35))
fir.run(x)
t = []
for _ in range(5):
    t0 = time.perf_counter(); y = fir.run(x); t.append(time.perf_counter() - t0)
dt = min(t)
print("comms_fir_run (H2D + kernel + D2H, pageable host memory), 2^24 samples: %.2f ms -> %.2f Gsamples/s, %.1f GB/s over PCIe (both ways)"
      % (dt * 1e3, n / dt / 1e9, 16 * n / dt / 1e9))
