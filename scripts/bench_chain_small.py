"""The FM chain at radio-sized blocks (2^16 ... 2^22 samples): time-domain kernel against the overlap-save path,
launch to completion (events around the call), for the two chains the reference's fm_radio example maps to."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import comms_rs_amd as c
s = torch.cuda.current_stream().cuda_stream
for n_taps, rate in ((63, 5), (127, 8), (255, 8)):
    taps = c.rrc_taps(n_taps, 8.0, 0.35)
    for lg in (16, 18, 20, 22):
        n = (1 << lg) // (rate * 16) * (rate * 16)
        x = torch.empty(n, dtype=torch.complex64, device="cuda:0")
        c.synth_iq_dev(x.data_ptr(), n, 0)
        row = []
        for kern in ("time", "freq"):
            node = c.ChainNode(0.3, 0.0, taps, rate, True, kernel=kern)
            out = torch.empty(n // rate, dtype=torch.float32, device="cuda:0")
            for _ in range(10): node.run_dev(x.data_ptr(), n, out.data_ptr(), s)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
            for a, b in ev:
                a.record(); node.run_dev(x.data_ptr(), n, out.data_ptr(), s); b.record()
            torch.cuda.synchronize()
            row.append("%s: %.1f us" % (node.kernel, np.median([a.elapsed_time(b) for a, b in ev]) * 1e3))
        print("%3d taps /%d  n=2^%d  %s" % (n_taps, rate, lg, "   ".join(row)), flush=True)
