"""fir_poly8_kernel against the oracle (small sizes, ragged calls with state) and against the time-domain chain kernel
(2^22 ... 2^26 samples), then bursts of launches of both, timed.  GPU box only."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import comms_rs_amd as c
from oracle import oracle

rng = np.random.default_rng(5)


def rand_c(n):
    return (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)


def lpf(n_taps, cutoff):
    k = np.arange(n_taps) - (n_taps - 1) / 2.0
    return (2 * cutoff * np.sinc(2 * cutoff * k) * np.hamming(n_taps)).astype(np.complex64)


def circ(d):
    return np.abs((d + np.pi) % (2 * np.pi) - np.pi)


worst = 0.0
for n_taps, cplx, after in [(255, False, True), (255, True, True), (255, False, False), (257, True, False), (131, False, True),
                            (200, True, True), (193, False, False), (129, True, True), (127, False, False), (65, False, True),
                            (33, False, True), (1, False, True), (8, True, False)]:
    taps = oracle.rrc_taps(n_taps, 8.0, 0.35) if n_taps > 8 else np.ones(n_taps, np.complex64)
    if cplx:
        taps = (taps * np.exp(1j * 0.01 * np.arange(n_taps))).astype(np.complex64)
    dphase, phase = 2 * np.pi * 0.1, 0.3
    node = c.ChainNode(dphase, phase, taps, 8, False, mixer_after_fir=after, kernel="poly")
    assert node.kernel == "poly", node.kernel
    n = 896 * 37 + 8 * 11
    x = rand_c(n)
    ost, om = oracle.default_state(taps), oracle.Mixer(phase, dphase)
    cuts = [0, 8, 776, 768 * 3, 768 * 3 + 16, 832 * 20 + 8 * 50, n]
    for a, b in zip(cuts[:-1], cuts[1:]):
        if after:
            w = oracle.decimate(om.mix(oracle.batch_fir(x[a:b], taps, ost, norotate=True)), 8)
        else:
            w = oracle.decimate(oracle.batch_fir(om.mix(x[a:b]), taps, ost, norotate=True), 8)
        got = node.run(x[a:b])
        scale = np.sum(np.abs(taps)) * np.max(np.abs(x))
        err = np.max(np.abs(got - w)) / scale
        worst = max(worst, err)
        assert err <= 2e-5, (n_taps, cplx, after, a, b, err)
    print("taps %3d cplx %d after %d: worst so far %.2e" % (n_taps, cplx, after, worst), flush=True)
print("oracle parity OK, worst %.3e of sum|taps| max|x|" % worst, flush=True)

# ---- with the FM demodulator (mixer -> FIR -> /8 -> FM: config 3's order), state across ragged calls
wfm = 0.0
for n_taps in (127, 63, 121, 122, 185, 186, 249):
    taps = lpf(n_taps, 1 / 16.0)
    n = 896 * 30 + 8 * 7
    t = np.arange(n)
    x = (np.exp(1j * (0.02 * t + 3.0 * np.sin(2 * np.pi * t / 5000.0))) * (1 + 0.1 * rng.standard_normal(n))).astype(np.complex64)
    node = c.ChainNode(0.3, 0.1, taps, 8, True, kernel="poly")
    assert node.kernel == "poly", node.kernel
    ost, om, ofm = oracle.default_state(taps), oracle.Mixer(0.1, 0.3), oracle.FM()
    cuts = [0, 8, 16, 896, 896 * 3 + 24, 832 * 11, n]
    for a, b in zip(cuts[:-1], cuts[1:]):
        y = oracle.decimate(oracle.batch_fir(om.mix(x[a:b]), taps, ost, norotate=True), 8)
        prev = ofm.prev if hasattr(ofm, "prev") else None
        w = ofm.demod(y)
        got = node.run(x[a:b])
        mag = np.minimum(np.abs(y), np.abs(np.concatenate([[1.0], y[:-1]])))
        ok = mag > 0.05
        e = circ(got.astype(np.float64) - w)
        if ok.any():
            wfm = max(wfm, float(np.max(e[ok])))
            assert np.max(e[ok]) <= 1e-4, (n_taps, a, b, float(np.max(e[ok])), int(np.argmax(e * ok)))
    print("FM taps %3d: worst angle error so far %.2e rad" % (n_taps, wfm), flush=True)
assert c.ChainNode(0.3, 0.1, lpf(255, 1 / 16.0), 8, True, kernel="poly").kernel != "poly"  # 250 taps and more: no spare halo position
print("FM parity OK, worst %.3e rad" % wfm, flush=True)

taps = oracle.rrc_taps(255, 8.0, 0.35)
for lg in (20, 22, 24, 26):
    n = 1 << lg
    x = torch.empty(n, dtype=torch.complex64, device="cuda:0")
    c.synth_iq_dev(x.data_ptr(), n, 0)
    s = torch.cuda.current_stream().cuda_stream
    kerns = ("time", "poly")
    nodes = [c.ChainNode(2 * np.pi * 0.05, 0.1, taps, 8, False, mixer_after_fir=True, kernel=k) for k in kerns]
    outs = [torch.empty(n // 8, dtype=torch.complex64, device="cuda:0") for _ in kerns]
    for i, nd in enumerate(nodes):
        nd.run_dev(x.data_ptr(), n, outs[i].data_ptr(), s)
    torch.cuda.synchronize()
    d = (outs[0] - outs[1]).abs().max().item()
    sc = np.sum(np.abs(taps)) * float(x[:1 << 20].abs().max().item())
    print("2^%d poly vs time (first call, zero state): max |diff| %.3e = %.2e of sum|taps| max|x|" % (lg, d, d / sc), flush=True)
    assert d / sc < 2e-5
    ts = [[] for _ in kerns]
    for rep in range(8):
        for i, nd in enumerate(nodes):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                nd.run_dev(x.data_ptr(), n, outs[i].data_ptr(), s)
            b.record()
            torch.cuda.synchronize()
            ts[i].append(a.elapsed_time(b) / 20 * 1e3)
    for i, k in enumerate(kerns):
        print("2^%d %s: median %.1f us per launch (bursts of 20, kernels taking turns): %s" % (lg, k, np.median(ts[i]), " ".join("%.1f" % t for t in ts[i])), flush=True)

# ---- config 3: mixer -> 127 taps -> /8 -> FM at 2^26: the product's kernel (wave-private time-domain form) against the polyphase one
taps = lpf(127, 1 / 16.0)
n = 1 << 26
x = torch.empty(n, dtype=torch.complex64, device="cuda:0")
c.synth_iq_dev(x.data_ptr(), n, 0)
s = torch.cuda.current_stream().cuda_stream
for fm in (True, False):
    kerns = ("time", "poly")
    nodes = [c.ChainNode(2 * np.pi * 0.05, 0.0, taps, 8, fm, kernel=k) for k in kerns]
    outs = [torch.empty(n // 8, dtype=torch.float32 if fm else torch.complex64, device="cuda:0") for _ in kerns]
    for i, nd in enumerate(nodes):
        nd.run_dev(x.data_ptr(), n, outs[i].data_ptr(), s)
    torch.cuda.synchronize()
    d = (outs[0] - outs[1]).abs()
    if fm:
        d = torch.minimum(d, 2 * np.pi - d)
    print("config 3 (fm %d) poly vs time: max |diff| %.3e, mean %.3e" % (fm, d.max().item(), d.mean().item()), flush=True)
    ts = [[] for _ in kerns]
    for rep in range(8):
        for i, nd in enumerate(nodes):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                nd.run_dev(x.data_ptr(), n, outs[i].data_ptr(), s)
            b.record()
            torch.cuda.synchronize()
            ts[i].append(a.elapsed_time(b) / 20 * 1e3)
    for i, k in enumerate(kerns):
        print("config 3 (fm %d) %s: median %.1f us per launch: %s" % (fm, k, np.median(ts[i]), " ".join("%.1f" % t for t in ts[i])), flush=True)
