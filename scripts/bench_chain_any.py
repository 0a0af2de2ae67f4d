"""The chain at rates the per-rate time-domain kernel is not built for: the any-rate kernel (fir_decim_any_kernel,
kernel="time") against the overlap-save launch (kernel="freq") and the automatic choice, 2^24-ish samples.
usage: python scripts/bench_chain_any.py [nofm]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import comms_rs_amd as c
FM = not (len(sys.argv) > 1 and sys.argv[1] == "nofm")
s = torch.cuda.current_stream().cuda_stream
for rate in (7, 9, 11, 13, 14, 15, 17, 20, 24, 25, 32, 40, 47, 48, 64, 100, 200, 256, 1000, 4096):
    n = (1 << 24) // rate * rate
    x = torch.empty(n, dtype=torch.complex64, device="cuda:0")
    c.synth_iq_dev(x.data_ptr(), n, 0)
    for n_taps in (63, 127, 255):
        taps = c.rrc_taps(n_taps, 8.0, 0.35)
        row = []
        for kern in ("time", "freq"):
            node = c.ChainNode(0.3, 0.0, taps, rate, FM, kernel=kern)
            out = torch.empty(n // rate, dtype=torch.float32 if FM else torch.complex64, device="cuda:0")
            for _ in range(5): node.run_dev(x.data_ptr(), n, out.data_ptr(), s)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
            for a, b in ev:
                a.record(); node.run_dev(x.data_ptr(), n, out.data_ptr(), s); b.record()
            torch.cuda.synchronize()
            row.append("%s(%s): %6.1f us" % (kern, node.kernel, np.median([a.elapsed_time(b) for a, b in ev]) * 1e3))
        auto = c.ChainNode(0.3, 0.0, taps, rate, FM)
        print("rate %4d taps %3d  %s   auto -> %s" % (rate, n_taps, "   ".join(row), auto.kernel), flush=True)
    del x
