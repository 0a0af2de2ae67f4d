"""Decimating FM chain (mixer -> FIR -> /R -> FM demod) on the time-domain kernel, the overlap-save fusion and what
ChainNode picks by itself, over the decimation rates the time-domain kernel is instantiated for (2^24 samples)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import comms_rs_amd as c
FM = not (len(sys.argv) > 1 and sys.argv[1] == "nofm")   # `nofm`: the chain without the demodulator (complex output)
n = 720 * 23301   # 16776720: a multiple of every rate below
x = torch.empty(n, dtype=torch.complex64, device="cuda:0")
c.synth_iq_dev(x.data_ptr(), n, 0)
s = torch.cuda.current_stream().cuda_stream
for rate in (2, 3, 4, 5, 6, 8, 10, 12, 16):
    for n_taps in (127, 255):
        taps = c.rrc_taps(n_taps, 8.0, 0.35)
        row = []
        for kern in ("time", "freq"):
            try:
                node = c.ChainNode(0.3, 0.0, taps, rate, FM, kernel=kern)
            except Exception as e:
                row.append("%s: n/a" % kern); continue
            out = torch.empty(n // rate, dtype=torch.float32 if FM else torch.complex64, device="cuda:0")
            for _ in range(5): node.run_dev(x.data_ptr(), n, out.data_ptr(), s)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
            for a, b in ev:
                a.record(); node.run_dev(x.data_ptr(), n, out.data_ptr(), s); b.record()
            torch.cuda.synchronize()
            row.append("%s(%s): %.1f us" % (kern, node.kernel, np.median([a.elapsed_time(b) for a, b in ev]) * 1e3))
        auto = c.ChainNode(0.3, 0.0, taps, rate, FM)
        print("rate %2d taps %3d (%.1f MACs/sample)  %s   auto -> %s" % (rate, n_taps, n_taps / rate, "   ".join(row), auto.kernel), flush=True)
