"""fir_poly8_kernel at rates 4 (two output phases) and 8 m (every m-th output): oracle parity in ragged calls, then the chain's
kernels taking turns at 2^24 samples.  GPU box only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import comms_rs_amd as c
from oracle import oracle

rng = np.random.default_rng(7)


def rand_c(n):
    return (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)


worst = 0.0
for rate in (4, 12, 20, 28, 60, 16, 24, 32, 40, 64):
    for n_taps, cplx, after in [(255, False, True), (257, True, False), (131, False, False), (100, True, True), (65, False, True)]:
        taps = oracle.rrc_taps(n_taps, 8.0, 0.35)
        if cplx:
            taps = (taps * np.exp(1j * 0.01 * np.arange(n_taps))).astype(np.complex64)
        dphase, phase = 2 * np.pi * 0.1, 0.3
        node = c.ChainNode(dphase, phase, taps, rate, False, mixer_after_fir=after, kernel="poly")
        n = rate * (896 * 37 // rate * 8 // 8 + 11)
        x = rand_c(n)
        ost, om = oracle.default_state(taps), oracle.Mixer(phase, dphase)
        cuts = [0, rate, rate * 9, rate * (2500 // rate), rate * (9000 // rate + 1), n]
        for a, b in zip(cuts[:-1], cuts[1:]):
            if after:
                w = oracle.decimate(om.mix(oracle.batch_fir(x[a:b], taps, ost, norotate=True)), rate)
            else:
                w = oracle.decimate(oracle.batch_fir(om.mix(x[a:b]), taps, ost, norotate=True), rate)
            got = node.run(x[a:b])
            assert node.kernel == "poly", (rate, n_taps, node.kernel)
            err = np.max(np.abs(got - w)) / (np.sum(np.abs(taps)) * np.max(np.abs(x)))
            worst = max(worst, err)
            assert err <= 2e-5, (rate, n_taps, cplx, after, a, b, err)
    print("rate %d: worst so far %.2e" % (rate, worst), flush=True)
print("oracle parity OK", flush=True)

taps = oracle.rrc_taps(255, 8.0, 0.35)
n = 1 << 24
x = torch.empty(n, dtype=torch.complex64, device="cuda:0")
c.synth_iq_dev(x.data_ptr(), n, 0)
s = torch.cuda.current_stream().cuda_stream
for rate in (4, 8, 12, 20, 28, 44, 60, 16, 24, 32, 48, 64):
    nn = n - n % (rate * 1024)
    kerns = ("time", "freq", "auto")
    nodes = [c.ChainNode(2 * np.pi * 0.05, 0.1, taps, rate, False, mixer_after_fir=True, kernel=k) for k in kerns]
    outs = [torch.empty(nn // rate, dtype=torch.complex64, device="cuda:0") for _ in kerns]
    for i, nd in enumerate(nodes):
        nd.run_dev(x.data_ptr(), nn, outs[i].data_ptr(), s)
    torch.cuda.synchronize()
    d = max((outs[0] - outs[2]).abs().max().item(), (outs[1] - outs[2]).abs().max().item())
    ts = [[] for _ in kerns]
    for rep in range(6):
        for i, nd in enumerate(nodes):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                nd.run_dev(x.data_ptr(), nn, outs[i].data_ptr(), s)
            b.record()
            torch.cuda.synchronize()
            ts[i].append(a.elapsed_time(b) / 20 * 1e3)
    print("rate %2d, 255 taps, 2^24: %s   max |diff| %.2e" % (rate, "   ".join("%s (%s) %.1f us" % (k, nodes[i].kernel, np.median(ts[i][1:])) for i, k in enumerate(kerns)), d), flush=True)
