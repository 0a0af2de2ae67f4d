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
    cl = c.ChainNode(0.2, 0.0, _lp(4200), 16, False)                  # beyond 4097 taps: FIR launches + mixer/decimator pass
    assert not cl.fused
    yl, _, _ = _buffers(n // 16)
    out.append(("chain 4200 taps", cl, lambda: cl.run_dev(x.data_ptr(), n, yl.data_ptr(), s)))
    c16 = c.ChainNode(0.2, 0.0, _lp(1600), 16, False)                 # 1538 ... 4097 taps: the 16384-point kernel, decimating
    assert c16.kernel == "freq"
    out.append(("chain 1600 taps", c16, lambda: c16.run_dev(x.data_ptr(), n, yl.data_ptr(), s)))
    co = c.ChainNode(0.2, 0.0, _lp(600), 16, False)                   # 514 ... 1537 taps: the 4096-point kernel, decimating
    assert co.kernel == "freq"
    out.append(("chain 600 taps", co, lambda: co.run_dev(x.data_ptr(), n, yl.data_ptr(), s)))
    cp = c.ChainNode(0.2, 0.0, _lp(511), 16, False)                   # 258 ... 513 taps: the polyphase kernel with eight halo rows
    assert cp.kernel == "poly"
    out.append(("chain 511 taps", cp, lambda: cp.run_dev(x.data_ptr(), n, yl.data_ptr(), s)))
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
    """Sixteen times the samples, about eleven times the time (the headline kernel, own begin / end timestamps, fixed
    launch cost included): the pair measures the kernel, not the dispatch.  Long warm-up first -- the chip's clocks
    move by tens of per cent over the first launches of a burst."""
    import torch

    node = c.BatchFirNode(_lp(255))
    res = {}
    for lg in (22, 26):
        n = 1 << lg
        x, y, s = _buffers(n)
        c.synth_iq_dev(x.data_ptr(), n, 0, 3)
        assert node.kernel_for(n) == "fir_os1024_dyn_kernel"
        for _ in range(200 if lg == 22 else 40):
            node.run_dev(x.data_ptr(), n, y.data_ptr(), s)
        t = c.KernelTimer(30).attach(node)
        for _ in range(30):
            node.run_dev(x.data_ptr(), n, y.data_ptr(), s)
        res[lg] = float(np.median(t.read_ms()))
        t.close()
        del x, y
        torch.cuda.empty_cache()
    assert 8.0 < res[26] / res[22] < 14.0, res   # (19 us and 210 us on the boxes seen; a dispatch-bound figure would not scale at all)


def test_timer_stride_brackets_every_kth_launch(c):
    """comms_timer_set_stride: events around launches 0, k, 2k, ... only -- on a node that records around its launch
    (mixer) and on one that hands the pair to the launch itself (ticketed FIR)."""
    n = 1 << 23
    x, y, s = _buffers(n)
    c.synth_iq_dev(x.data_ptr(), n, 0, 5)
    mx, fir = c.MixerNode(0.3), c.BatchFirNode(_lp(255))
    for node, launch in ((mx, lambda: mx.run_dev(x.data_ptr(), n, y.data_ptr(), s)),
                         (fir, lambda: fir.run_dev(x.data_ptr(), n, y.data_ptr(), s))):
        t = c.KernelTimer(8, stride=3).attach(node)
        for _ in range(7):
            launch()
        ms = t.read_ms()
        assert ms.size == 3 and np.all(ms > 0), ms       # launches 0, 3, 6
        t.reset()
        launch()
        assert t.read_ms().size == 1
        t.close()


def test_stamps_timer_times_every_launch_in_the_stream(c):
    """comms_timer_create_stamps / _add_stamps: the kernels that take a KStamp write their own begin / end
    (s_memrealtime) -- ordinary launches, every one timed.  The stamped duration of the headline kernel agrees with
    the event pair's; a 'both' timer holds events for every 4th launch and stamps for all."""
    import torch

    n = 1 << 24
    x, y, s = _buffers(n)
    c.synth_iq_dev(x.data_ptr(), n, 0, 5)
    fir = c.BatchFirNode(_lp(255))
    assert fir.kernel_for(n) == "fir_os1024_dyn_kernel"
    for _ in range(100):
        fir.run_dev(x.data_ptr(), n, y.data_ptr(), s)
    t = c.KernelTimer(20, stamps="both", stride=4).attach(fir)
    for _ in range(20):
        fir.run_dev(x.data_ptr(), n, y.data_ptr(), s)
    ev, st = t.read_ms(), t.read_stamps_ms()
    assert ev.size == 5 and st.size == 20 and np.all(st > 0), (ev, st)
    assert 0.75 < float(np.median(st)) / float(np.median(ev)) < 1.2, (ev, st)
    # more launches than slots: the first 20 stay, nothing is written past them
    fir.run_dev(x.data_ptr(), n, y.data_ptr(), s)
    assert t.read_stamps_ms().size == 20
    t.reset()
    fir.run_dev(x.data_ptr(), n, y.data_ptr(), s)
    assert t.read_stamps_ms().size == 1 and t.read_ms().size == 1
    t.close()
    # the results do not depend on the stamps
    y0 = y.clone()
    fir2 = c.BatchFirNode(_lp(255))
    t2 = c.KernelTimer(4, stamps=True).attach(fir2)
    y1 = torch.empty_like(y)
    fir2.run_dev(x.data_ptr(), n, y1.data_ptr(), s)
    fir3 = c.BatchFirNode(_lp(255))
    fir3.run_dev(x.data_ptr(), n, y0.data_ptr(), s)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    assert t2.read_ms().size == 1  # (a stamps-only timer: _read returns the stamped durations)
    t2.close()


def test_stamps_timer_on_the_other_stamping_kernels(c):
    """16384-point FIR, decimating chain kernel, polyphase pulse shaper: every launch stamped, durations in range;
    a node whose kernel takes no stamp (mixer) leaves a stamps timer empty."""
    n = 1 << 22
    x, y, s = _buffers(n)
    f, _, _ = _buffers(n, real_out=True)
    c.synth_iq_dev(x.data_ptr(), n, 0, 5)
    big = c.BatchFirNode(_lp(2100)).set_algo(c.FIR_OS16K)
    ch = c.ChainNode(0.2, 0.0, _lp(127), 8, True, kernel="time")
    pn = c.PulseNode(c.rrc_taps(63, 4.0, 0.25), 4)
    mx = c.MixerNode(0.1)
    cases = (("os16k", big, lambda: big.run_dev(x.data_ptr(), n, y.data_ptr(), s), 5),
             ("chain", ch, lambda: ch.run_dev(x.data_ptr(), n, f.data_ptr(), s), 5),
             ("pulse", pn, lambda: pn.run_dev(x.data_ptr(), n // 4, y.data_ptr(), s), 5),
             ("mixer", mx, lambda: mx.run_dev(x.data_ptr(), n, y.data_ptr(), s), 0))
    for name, node, launch, want in cases:
        for _ in range(3):
            launch()
        t = c.KernelTimer(8, stamps=True).attach(node)
        for _ in range(5):
            launch()
        ms = t.read_stamps_ms()
        assert ms.size == want, (name, ms)
        assert np.all((ms > 0.002) & (ms < 5.0)), (name, ms)
        t.close()
