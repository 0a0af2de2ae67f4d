"""The kernel timer of the C ABI (comms_timer_*, comms_*_set_timer: include/comms_hip.h) on every node kind that takes one --
what bench.py's `roofline.kernel_ms` rests on.  The timer brackets the node's dominant kernel; recording is asynchronous,
reading waits; detaching keeps what was recorded; more launches than pairs keep the newest."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c():
    import comms_rs_amd

    return comms_rs_amd


def _buffers(n, real_out=False):
    import torch

    x = torch.empty(n, dtype=torch.complex64, device="cuda:0")
    y = torch.empty(n, dtype=torch.float32 if real_out else torch.complex64, device="cuda:0")
    return x, y, torch.cuda.current_stream().cuda_stream


def _lp(n_taps):
    k = np.arange(n_taps) - (n_taps - 1) / 2
    return (0.1 * np.sinc(0.1 * k) * np.hamming(n_taps)).astype(np.float32).astype(np.complex64)


def _cases(c):
    n = 1 << 20
    x, y, s = _buffers(n)
    c.synth_iq_dev(x.data_ptr(), n, 0, 9)
    f, _, _ = _buffers(n, real_out=True)
    big = 1 << 23  # enough segments for the ticketed kernel
    xb, yb, _ = _buffers(big)
    c.synth_iq_dev(xb.data_ptr(), big, 0, 9)
    out = []
    for name, algo, taps in (("direct", c.FIR_DIRECT, 32), ("os1024 fixed runs", c.FIR_OS1024, 255), ("os4096", c.FIR_OS4096, 511),
                             ("os16k", c.FIR_OS16K, 2100)):
        node = c.BatchFirNode(_lp(taps)).set_algo(algo)
        out.append(("fir " + name, node, lambda node=node: node.run_dev(x.data_ptr(), n, y.data_ptr(), s)))
    dyn = c.BatchFirNode(_lp(255))
    assert dyn.kernel_for(big) == "fir_os1024_dyn_kernel"
    out.append(("fir os1024 ticketed", dyn, lambda: dyn.run_dev(xb.data_ptr(), big, yb.data_ptr(), s)))
    mx = c.MixerNode(0.3)
    out.append(("mixer", mx, lambda: mx.run_dev(x.data_ptr(), n, y.data_ptr(), s)))
    fm = c.FMDemodNode()
    out.append(("fm demod", fm, lambda: fm.run_dev(x.data_ptr(), n, f.data_ptr(), s)))
    for N in (1024, 1 << 16, 1 << 21, 1000, 10):  # one pass, four-step, gathered columns, Bluestein, direct DFT
        ft = c.FFTBatchNode(N, False)
        src, dst, tot = (x, y, n) if N <= n else (xb, yb, big)
        m = (tot // N) * N
        out.append(("fft %d" % N, ft, lambda ft=ft, m=m, src=src, dst=dst: ft.run_dev(src.data_ptr(), m, dst.data_ptr(), s)))
    for ask, kern, rate, taps in (("time", "time", 8, 127), ("time", "time_any", 20, 127), ("freq", "freq", 8, 127)):
        ch = c.ChainNode(0.2, 0.0, _lp(taps), rate, True, kernel=ask)
        assert ch.kernel == kern
        m = (n // rate) * rate
        out.append(("chain " + kern, ch, lambda ch=ch, m=m: ch.run_dev(x.data_ptr(), m, f.data_ptr(), s)))
    cu = c.ChainNode(0.2, 0.0, _lp(127), 8, True, unfused=True)       # the four kernels in series: the FIR stage is timed
    out.append(("chain unfused", cu, lambda: cu.run_dev(x.data_ptr(), n, f.data_ptr(), s)))
    cl = c.ChainNode(0.2, 0.0, _lp(511), 16, False)                   # beyond 257 taps: FIR launch + mixer/decimator pass
    yl, _, _ = _buffers(n // 16)
    out.append(("chain 511 taps", cl, lambda: cl.run_dev(x.data_ptr(), n, yl.data_ptr(), s)))
    pp = c.PulseNode(_lp(63), 4).set_mixer(0.3, 0.0)
    out.append(("pulse polyphase", pp, lambda: pp.run_dev(x.data_ptr(), n // 4, y.data_ptr(), s)))
    pg = c.PulseNode(_lp(63), 7)
    out.append(("pulse generic", pg, lambda: pg.run_dev(x.data_ptr(), n // 8, y.data_ptr(), s)))
    return out, (x, y, f, xb, yb, yl)


def test_timer_brackets_the_dominant_kernel_of_every_node_kind(c):
    import torch

    cases, keep = _cases(c)
    for name, node, launch in cases:
        launch()  # first call: set-up outside the timer
        t = c.KernelTimer(5).attach(node)
        for _ in range(3):
            launch()
        ms = t.read_ms()
        assert ms.size == 3, name
        assert np.all(ms > 0) and np.all(ms < 100.0), (name, ms)
        # detached: launches are not recorded, what was recorded stays
        t.enable(False)
        launch()
        assert t.read_ms().size == 3, name
        # attached again; more launches than pairs keep the newest five
        t.enable(True)
        for _ in range(4):
            launch()
        ms2 = t.read_ms()
        assert ms2.size == 5 and np.all(ms2 > 0), (name, ms2)
        t.reset()
        assert t.read_ms().size == 0, name
        launch()
        assert t.read_ms().size == 1, name
        t.close()
        launch()  # the node runs on without its timer
    torch.cuda.synchronize()
    del keep


def test_timer_durations_follow_the_work(c):
    """Four times the samples, several times the time (the headline kernel, own begin / end timestamps): the pair
    measures the kernel, not the dispatch."""
    import torch

    node = c.BatchFirNode(_lp(255))
    res = {}
    for lg in (23, 25):
        n = 1 << lg
        x, y, s = _buffers(n)
        c.synth_iq_dev(x.data_ptr(), n, 0, 3)
        for _ in range(30):
            node.run_dev(x.data_ptr(), n, y.data_ptr(), s)
        t = c.KernelTimer(20).attach(node)
        for _ in range(20):
            node.run_dev(x.data_ptr(), n, y.data_ptr(), s)
        res[lg] = float(np.median(t.read_ms()))
        t.close()
        del x, y
        torch.cuda.empty_cache()
    assert 2.0 < res[25] / res[23] < 8.0, res   # (wide: clocks move by tens of per cent over a burst; a dispatch-bound figure would not scale at all)
