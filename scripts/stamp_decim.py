#!/usr/bin/env python3
"""Per-phase cycle stamps of fir_decim_kernel on config 3 (diagnostic; every stamp drains the
memory counters, so read the SHARES, not the total)."""
import os as _os; _os.environ.setdefault("COMMS_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "comms_rs_amd", "lib", "libcomms_hip_diag.so"))  # diagnostic build (`make -C comms_rs_amd/csrc diag`)
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import comms_rs_amd as c

if os.environ.get("CASE", "c3") == "c2":   # the metric's chain as one launch: 255 taps -> mixer -> /8, 2^24
    n = 1 << 24
    taps = c.rrc_taps(255, 8.0, 0.35)
    x = torch.empty(n, dtype=torch.complex64, device="cuda:0")
    c.synth_iq_dev(x.data_ptr(), n, 0)
    out = torch.empty(n // 8, dtype=torch.complex64, device="cuda:0")
    ch = c.ChainNode(2 * np.pi * 0.05, 0.0, taps, 8, False, mixer_after_fir=True, kernel="time")
else:                                      # config 3 (needs COMMS_DECIM_WAVE=0: the stamps are the workgroup kernel's)
    n = 1 << 26
    k = np.arange(127) - 63.0
    taps = (2 / 16 * np.sinc(2 / 16 * k) * np.hamming(127)).astype(np.complex64)
    x = torch.empty(n, dtype=torch.complex64, device="cuda:0")
    c.synth_iq_dev(x.data_ptr(), n, 0)
    out = torch.empty(n // 8, dtype=torch.float32, device="cuda:0")
    ch = c.ChainNode(2 * np.pi * 0.05, 0.0, taps, 8, True, kernel="time")
s = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    ch.run_dev(x.data_ptr(), n, out.data_ptr(), s)
buf = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda:0")
f = c.lib().comms_debug_decim_stamps
f.restype = None
f.argtypes = [C.c_void_p]
f(buf.data_ptr())
ch.run_dev(x.data_ptr(), n, out.data_ptr(), s)
torch.cuda.synchronize()
f(None)
st = buf.cpu().numpy().reshape(-1, 8)[:, :6].astype(np.float64)
st = st[st.sum(axis=1) > 0]
tot = st.sum(axis=1).mean()
names = ["global loads landed", "mixer + LDS writes", "barrier 1", "(unused)", "filter loop", "barrier 2 + epilogue + stores"]
print("waves %d, ticks per wave %.0f (s_memtime, 100 MHz)" % (st.shape[0], tot))
for i, nm in enumerate(names):
    print("  %-32s %8.0f  %5.1f%%" % (nm, st[:, i].mean(), 100 * st[:, i].mean() / tot))
