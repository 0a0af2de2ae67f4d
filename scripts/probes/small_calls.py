import sys, os
sys.path.insert(0, "/root/repo")
import torch, numpy as np
import comms_rs_amd as c
s = torch.cuda.current_stream().cuda_stream
x = torch.empty(1 << 16, dtype=torch.complex64, device="cuda:0"); y = torch.empty_like(x)
c.synth_iq_dev(x.data_ptr(), 1 << 16, 0, 1)
for N in (64, 256, 1024, 4096, 16384):
    node = c.FFTBatchNode(N, False)
    for n in (N, 16 * N if 16 * N <= (1 << 16) else N):
        for _ in range(50):
            node.run_dev(x.data_ptr(), n, y.data_ptr(), s)
torch.cuda.synchronize()
fir = c.BatchFirNode(c.rrc_taps(63, 4.0, 0.25))
mix = c.MixerNode(0.1)
for _ in range(50):
    fir.run_dev(x.data_ptr(), 4096, y.data_ptr(), s); mix.run_dev(x.data_ptr(), 4096, y.data_ptr(), s)
torch.cuda.synchronize()
