"""A/B timing of two or more builds of the library on the fused chains, launch by launch in ONE process.
usage: python scripts/ab_chain.py <libA.so> [libB.so ...] [reps]
Cases: config 3 (mixer -> 127 real taps -> /8 -> FM, 2^26) and the metric chain (255 real taps -> mixer -> /8, 2^24) on the
time-domain kernel; CASES=c3,c2,c2fm,r5 selects.  Every lib also gets an output check against the first lib."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import comms_rs_amd as c  # synthetic input + taps from the default build

paths = [a for a in sys.argv[1:] if ".so" in a]
nums = [a for a in sys.argv[1:] if ".so" not in a]
reps = int(nums[0]) if nums else 60


def lpf(n_taps, cutoff):
    k = np.arange(n_taps) - (n_taps - 1) / 2.0
    return np.ascontiguousarray((2 * cutoff * np.sinc(2 * cutoff * k) * np.hamming(n_taps)).astype(np.complex64))


libs = []
for p in paths:
    # "path.so:ENV=VAL" sets an environment variable for that entry (diagnostic builds read knobs at first use)
    libs.append(C.CDLL(os.path.abspath(p.split(":")[0])))
s = torch.cuda.current_stream().cuda_stream
cases = {"c3": (26, lpf(127, 1 / 16.0), 8, 1 | 16), "c2": (24, np.ascontiguousarray(c.rrc_taps(255, 8.0, 0.35)), 8, 2 | 16),
         "r5": (26, lpf(63, 1 / 10.0), 5, 1 | 16), "c3n": (26, lpf(127, 1 / 16.0), 8, 16),
         "c2b": (26, np.ascontiguousarray(c.rrc_taps(255, 8.0, 0.35)), 8, 2 | 16), "c2c": (25, np.ascontiguousarray(c.rrc_taps(255, 8.0, 0.35)), 8, 2 | 16)}
for name in os.environ.get("CASES", "c3,c2").split(","):
    logn, taps, rate, flags = cases[name]
    n = 1 << logn
    n -= n % (rate * 1024)
    x = torch.empty(n, dtype=torch.complex64, device="cuda:0")
    c.synth_iq_dev(x.data_ptr(), n, 0)
    fm = flags & 1
    outs = [torch.zeros(n // rate, dtype=torch.float32 if fm else torch.complex64, device="cuda:0") for _ in libs]
    hs = []
    for l in libs:
        h = C.c_void_p()
        l.comms_chain_create_ex.argtypes = [C.c_double, C.c_double, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        assert l.comms_chain_create_ex(2 * np.pi * 0.05, 0.0, taps.ctypes.data, taps.size, rate, flags, 0, C.byref(h)) == 0
        l.comms_chain_run_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        f = C.c_int32()
        l.comms_chain_is_fused(h, C.byref(f))
        assert f.value == 2, f.value
        hs.append(h)
    run = lambda i: libs[i].comms_chain_run_dev(hs[i], x.data_ptr(), n, outs[i].data_ptr(), s)
    for i in range(len(libs)):
        assert run(i) == 0
    torch.cuda.synchronize()
    for i in range(1, len(libs)):
        a, b = outs[0], outs[i]
        d = (a - b).abs()
        if fm:
            d = torch.minimum(d, 2 * np.pi - d)  # angles: compare on the circle
        if os.environ.get("AB_WHERE"):  # where two builds disagree (a new kernel's first runs)
            bad = torch.nonzero(d > float(os.environ["AB_WHERE"])).flatten()
            print("   %d of %d outputs differ by more than %s; first: %s; mod 128: %s; mod %d (a run of 16 tiles): %s" % (
                bad.numel(), d.numel(), os.environ["AB_WHERE"], bad[:12].tolist(), sorted(set((bad[:2000] % 128).tolist()))[:20],
                2048, sorted(set((bad[:2000] % 2048).tolist()))[:20]))
        d = d.max().item()
        print("%s: %s vs first: max|d| = %.3e (max|y| %.3e)" % (name, os.path.basename(paths[i]), d, a.abs().max().item()))
    for _ in range(10):
        for i in range(len(libs)):
            run(i)
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)] for _ in libs]
    for r in range(reps):
        for i in range(len(libs)):
            a, b = ev[i][r]
            a.record(); run(i); b.record()
    torch.cuda.synchronize()
    for i, p in enumerate(paths):
        v = np.array([a.elapsed_time(b) for a, b in ev[i]]) * 1e3
        print("%-4s %-36s median %.2f us  mean %.2f  p10 %.2f  p90 %.2f" % (name, os.path.basename(p), np.median(v), v.mean(), *np.percentile(v, [10, 90])), flush=True)
    # back to back: bursts of 20 launches of one lib, the libs taking turns burst by burst (clock drift hits all alike)
    ts = [[] for _ in libs]
    for _ in range(8):
        for i in range(len(libs)):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                run(i)
            b.record()
            torch.cuda.synchronize()
            ts[i].append(a.elapsed_time(b) * 1e3 / 20)
    for i, p in enumerate(paths):
        print("%-4s %-36s back to back %.2f us per launch (median of 8 interleaved bursts of 20; min %.2f)" % (name, os.path.basename(p), np.median(ts[i]), min(ts[i])), flush=True)
    for l, h in zip(libs, hs):
        l.comms_chain_destroy.argtypes = [C.c_void_p]
        l.comms_chain_destroy(h)
