"""One node type at a time, 300 create / destroy cycles each: device memory lost per type (tests/test_gpu_parity.py::
test_handles_create_destroy_many_times holds nine handles at a time; this tells the types apart).  usage: python scripts/probe_leak.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import comms_rs_amd as c
rng = np.random.default_rng(3)
x = (rng.standard_normal(4096) + 1j * rng.standard_normal(4096)).astype(np.complex64)
k = np.arange(63) - 31.0
taps = (0.2 * np.sinc(0.2 * k) * np.hamming(63)).astype(np.complex64)
makers = {"fir": lambda i: c.BatchFirNode(taps), "mixer": lambda i: c.MixerNode(0.1), "fm": lambda i: c.FMDemodNode(), "fft1024": lambda i: c.FFTBatchNode(1024, bool(i & 1)),
          "fft1000": lambda i: c.FFTBatchNode(1000, False), "pulse": lambda i: c.PulseNode(taps, 4), "chain": lambda i: c.ChainNode(0.1, 0.0, taps, 4, True),
          "timing": lambda i: c.TimingEstimatorNode(4, 2, 0.5), "nco": lambda i: c.NcoNode(0.1)}
for name, mk in makers.items():
    def cycle(i):
        nd = mk(i)
        if i % 50 == 0 and name not in ("timing", "nco"):
            nd.run(x[:4000] if not isinstance(nd, c.FFTBatchNode) else x[:nd.fft_size])
        del nd
    for i in range(51): cycle(i)
    torch.cuda.synchronize(); f0, _ = torch.cuda.mem_get_info()
    for i in range(300): cycle(i)
    torch.cuda.synchronize(); f1, _ = torch.cuda.mem_get_info()
    print("%-8s lost %8.1f KiB over 300 create/destroy cycles" % (name, (f0 - f1) / 1024.0), flush=True)
