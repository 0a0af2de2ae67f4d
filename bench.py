#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X.

Default (what the driver runs):  --config 2
  Metric : Msamples/s of Complex<f32> through the 255-tap FIR -> mixer -> decimate
           chain (BASELINE.json `metric`), whole job over all ranks.
  Step   : one pass of the chain over one resident batch of 2^24 synthetic IQ
           samples per GPU (BASELINE config 2), node by node through the C ABI
           (comms_fir_run_dev -> comms_mixer_run_dev -> comms_decimate_run_dev),
           device-resident in and out, FIR/mixer state carried across steps.
  N > 1  : the long stream is cut into N contiguous shards, one rank per GPU; the
           only cross-rank step is the one-off 255-sample hand-over to the
           right-hand neighbour over RCCL (setup, untimed).  The data path has no
           collective -> weak scaling (per-GPU work fixed) for `value`.
  Beside `value`, the same JSON line carries `stream_2p30`: the north-star's
  2^30-sample stream cut into N contiguous shards of 2^30/N samples per rank
  (strong scaling; every shard is far past the 256 MiB Infinity Cache, so its FIR
  fraction is the HBM-resident figure).

Other BASELINE configs (same contract, one JSON line each, each with roofline + cpu_baseline at N = 1):
  --config 1   the reference's own CPU case on the GPU: PRBS7 -> BPSK -> 63-tap RRC x4 -> mixer, 2^20 samples per step,
               one launch; cpu_baseline = the whole config through the oracle
  --config 3   mixer -> 127-tap LPF -> decimate-by-8 -> FM demod, 2^26 samples per rank,
               one fused launch; shards are primed by a 136-sample prefix (FIR halo + FM.prev)
  --config 4   2^20-point FFT, batch 4096 sharded over the ranks (4096/N transforms each), no
               communication at all
  --config 5   4097-tap overlap-save FIR, 2^27 samples per rank (2^30 over 8 GPUs), 4096-sample halo
  --variant scatter   SURVEY 8e's secondary variant: the root generates all shards, sends shard r
               to rank r point to point and collects the outputs; both transfers are timed and
               reported separately from the compute (`transfer`).
  --backend gloo --n-log2 K   rehearsal aid: CPU-side messages, ranks may share a GPU, smaller shards.

    python bench.py --gpus N --steps K --warmup W
        N > 1 without RANK in the environment: this process only spawns N rank processes
        (child processes, never exec; it has not touched the GPU) and relays rank 0's line.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
        (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment, one rank per GPU)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_SAMPLES = 1 << 24          # per GPU per step (BASELINE config 2)
N_TAPS = 255
DEC_RATE = 8
MIX_DPHASE = 2.0 * np.pi * 0.1
SEED = 0xC0FFEE
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FIR_BYTES_PER_SAMPLE = 16.0  # 8 B read + 8 B write per sample
C3_TAPS, C3_RATE, C3_DPHASE = 127, 8, 2.0 * np.pi * 0.05
C3_BYTES_PER_SAMPLE = 8.5    # fused config 3, 8 B in + 4/8 B out per input sample
C5_TAPS = 4097
FFT_N, FFT_BATCH = 1 << 20, 4096
FFT_BYTES_PER_POINT = 16.0   # one read + one write per point


def box_info(torch, local_rank):
    """Which machine and GPU produced the line (best effort): HBM-resident rates differ by box."""
    import glob
    import socket

    info = {"host": socket.gethostname()}
    try:
        pr = torch.cuda.get_device_properties(local_rank)
        info.update(gpu=pr.name, cus=pr.multi_processor_count, hbm_GiB=round(pr.total_memory / 2 ** 30, 1))
    except Exception:
        pass
    cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
    if local_rank < len(cards):
        for key in ("unique_id", "current_memory_partition", "current_compute_partition"):
            try:
                with open(os.path.join(cards[local_rank], key)) as f:
                    info[key] = f.read().strip()
            except Exception:
                pass
    return info


def pmc_traffic(kernel, n):
    """HBM bytes per FIR launch from the committed rocprofv3 PMC passes (profiles/), or None."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
        if d.get("kernel") == kernel and d.get("n_samples") == n:
            return d.get("hbm_bytes_per_launch")
        leg = None
        if ":" in kernel:
            kernel, leg = kernel.split(":", 1)
        for e in d.get("others", []):   # configs 3, 4, 5: the dominant kernel at the config's own size
            if kernel.startswith(e["kernel"]) and n in (e.get("n_samples"), e.get("n_points")) and e.get("leg") == leg:
                return e["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def copy_ceiling(torch, src, dst, reps=9):
    """What the memory system gives a plain device copy of this footprint in this run (hipMemcpyDtoD: 8 B read + 8 B written per
    Complex<f32>): the practical ceiling of a kernel that reads its input once and writes as much -- the spec peak the roofline
    fraction is quoted against is not reachable by any such kernel on this chip."""
    n = min(src.numel(), dst.numel())
    for _ in range(2):
        dst[:n].copy_(src[:n])
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        dst[:n].copy_(src[:n])
        b.record()
    torch.cuda.synchronize()
    ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
    gbs = 2.0 * n * src.element_size() / (ms * 1e-3) / 1e9
    return {"copy_GBps": round(gbs, 1), "copy_ms": round(ms, 5), "copy_bytes": 2 * n * src.element_size(),
            "frac_at_copy_rate": round(gbs / HBM_PEAK_GBS, 4),
            "note": "a plain device copy (hipMemcpyDtoD) of the same footprint, timed in this run" +
                    (": about the size of the 256 MiB Infinity Cache, which serves part of it -- above what HBM alone gives"
                     if 2 * n * src.element_size() <= (320 << 20) else ": far past the Infinity Cache, HBM's own rate for this read + write mix")}


def fm_radio_taps():
    """The 63 taps of examples/fm_radio.rs:30-52 (a fixture: data the reference holds)."""
    with open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")) as f:
        g = json.load(f)["fm_radio_taps"]
    return np.asarray(g["taps_re"], np.float32).astype(np.complex64)


def synth_u8(n, first=0):
    """RTL-SDR-shaped input for the literal fm_radio chain: an FM tone on a carrier offset, quantised to u8 pairs."""
    idx = np.arange(first, first + n, dtype=np.float64)
    ph = -2 * np.pi * 0.05 * idx + 8.0 * np.cos(2 * np.pi * idx / 4096)
    return np.clip(np.round(np.stack([np.cos(ph), np.sin(ph)], axis=1) * 100.0 + 127.5), 0, 255).astype(np.uint8)


def lowpass_taps(n_taps, cutoff):
    """Hamming-windowed sinc, f64 design on the host (configs 3 and 5)."""
    k = np.arange(n_taps) - (n_taps - 1) / 2.0
    h = 2 * cutoff * np.sinc(2 * cutoff * k) * np.hamming(n_taps)
    return (h / h.sum()).astype(np.complex64)


def host_cores():
    """(cores this process may run on, the cgroup's CPU quota in cores or None): what `all cores` means on this box."""
    try:
        usable = len(os.sched_getaffinity(0))
    except Exception:
        usable = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            a, b = f.read().split()
            if a != "max":
                quota = round(int(a) / int(b), 2)
    except Exception:
        pass
    return usable, quota


def all_cores_counts():
    """Thread counts of the all-cores baselines: every usable core (SURVEY 8d iii), and 64 beside it when the box has
    more -- thread start-up and the host's memory system can make the smaller pool the faster one; both are reported."""
    usable, _ = host_cores()
    return sorted({max(1, usable), max(1, min(usable, 64))}, reverse=True)


def cpu_baseline(n):
    """The oracle's restatement of the reference chain (literal batch_fir with
    rotate_right per sample, Mixer::mix in f64, decimate), one thread = what one
    reference node thread does.  Reported baseline, not the target."""
    import oracle

    x = np.empty(n, np.complex64)
    from comms_rs_amd import synth_iq

    x[:] = synth_iq(n, 0, SEED)
    taps = oracle.rrc_taps(N_TAPS, 8.0, 0.35)
    best = None
    for _ in range(2):
        st = oracle.default_state(taps)
        mx = oracle.Mixer(0.0, MIX_DPHASE)
        t0 = time.perf_counter()
        y = oracle.decimate(mx.mix(oracle.batch_fir(x, taps, st)), DEC_RATE)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    assert y.size == n // DEC_RATE
    out = {"value": round(n / best / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
           "sample": "full config-2 batch (%d samples) through oracle batch_fir(255 taps, rotate_right per "
                     "sample) -> Mixer::mix (f64) -> decimate(8), best of 2, g++ -O3 -ffp-contract=off, "
                     "1 thread of %d host cores" % (n, os.cpu_count() or 0)}
    # the same work cut into independent chunks over the host's cores (SURVEY 8d iii): an upper
    # bound for what a CPU graph of such nodes could reach on this box; the no-rotate variant
    # (circular index instead of the per-sample memmove) is listed beside it, labelled as such
    try:
        from concurrent.futures import ThreadPoolExecutor

        usable, quota = host_cores()
        out["host"] = {"cores_usable": usable, "cgroup_cpu_quota_cores": quota, "cores_online": os.cpu_count() or 0}

        def run_chunk(i, norotate, chunk):
            st = oracle.default_state(taps)
            seg = x[i * chunk:(i + 1) * chunk]
            oracle.decimate(oracle.Mixer(0.0, MIX_DPHASE).mix(oracle.batch_fir(seg, taps, st, norotate=norotate)), DEC_RATE)

        for key, norot in (("all_cores", False), ("all_cores_norotate", True)):
            runs = []
            for threads in all_cores_counts():
                chunk = n // threads // DEC_RATE * DEC_RATE
                with ThreadPoolExecutor(threads) as ex:
                    list(ex.map(lambda i: run_chunk(i, norot, chunk), range(threads)))  # warm-up: threads, first-touch pages
                    t0 = time.perf_counter()
                    list(ex.map(lambda i: run_chunk(i, norot, chunk), range(threads)))
                    dt = time.perf_counter() - t0
                runs.append({"value": round(chunk * threads / dt / 1e6, 2), "unit": "Msamples/s", "cores": threads})
            out[key] = dict(runs[0], other_thread_counts=runs[1:]) if len(runs) > 1 else runs[0]
        chunk = n // 64 // DEC_RATE * DEC_RATE
        t0 = time.perf_counter()
        run_chunk(0, True, chunk)
        out["one_core_norotate"] = {"value": round(chunk / (time.perf_counter() - t0) / 1e6, 2), "unit": "Msamples/s",
                                    "cores": 1}
        # one thread per node joined by unbounded queues, as start_nodes! runs the
        # reference graph (Vec batches of 2^18 samples); the slowest node (the FIR) sets the rate
        import queue
        import threading

        batch, q1, q2, got = 1 << 18, queue.Queue(), queue.Queue(), []
        st, mx = oracle.default_state(taps), oracle.Mixer(0.0, MIX_DPHASE)

        def fir_node():
            for i in range(0, n, batch):
                q1.put(oracle.batch_fir(x[i:i + batch], taps, st))
            q1.put(None)

        def mixer_node():
            while (v := q1.get()) is not None:
                q2.put(mx.mix(v))
            q2.put(None)

        def decimate_node():
            while (v := q2.get()) is not None:
                got.append(oracle.decimate(v, DEC_RATE).size)

        nodes = [threading.Thread(target=f) for f in (fir_node, mixer_node, decimate_node)]
        t0 = time.perf_counter()
        for t in nodes:
            t.start()
        for t in nodes:
            t.join()
        dt = time.perf_counter() - t0
        assert sum(got) == n // DEC_RATE
        out["thread_per_node"] = {"value": round(n / dt / 1e6, 3), "unit": "Msamples/s", "cores": 3}
    except Exception as e:  # the single-thread figure above is the contract; these are extras
        out["all_cores"] = {"error": str(e)}
    return out


def _best_of(fn, reps=2):
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best


def _pipeline_s(stages, batches):
    """one thread per node joined by unbounded queues, as start_nodes! runs the reference graph
    (src/node/mod.rs:276-284); `stages` are the nodes' run functions (stateful closures), `batches` the source's
    messages.  Returns the wall time from start to the last message leaving the last node."""
    import queue
    import threading

    qs = [queue.Queue() for _ in stages]

    def node(i):
        while (v := qs[i].get()) is not None:
            r = stages[i](v)
            if i + 1 < len(stages):
                qs[i + 1].put(r)
        if i + 1 < len(stages):
            qs[i + 1].put(None)

    th = [threading.Thread(target=node, args=(i,)) for i in range(len(stages))]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for b in batches:
        qs[0].put(b)
    qs[0].put(None)
    for t in th:
        t.join()
    return time.perf_counter() - t0


def _all_cores_s(run_chunk, cap=None):
    """the work cut into independent chunks over ALL the host's usable cores (each chunk starts from
    fresh node state: an upper bound for a CPU graph of such nodes).  run_chunk(i, threads) does chunk i of `threads`.
    Returns (seconds, threads) at every usable core; where the box has more than 64, the 64-thread run is timed as well and
    the faster of the two is what is returned (ALL_CORES_LOG keeps both for the line)."""
    from concurrent.futures import ThreadPoolExecutor

    best = None
    counts = all_cores_counts() if cap is None else sorted({min(t, cap) for t in all_cores_counts()}, reverse=True)  # (cap: units of work there are)
    for threads in counts:
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(lambda i: run_chunk(i, threads), range(threads)))  # warm-up: threads, first-touch pages
            t0 = time.perf_counter()
            list(ex.map(lambda i: run_chunk(i, threads), range(threads)))
            dt = time.perf_counter() - t0
        ALL_CORES_LOG.append({"cores": threads, "seconds": round(dt, 4)})
        if best is None:
            best = (dt, threads)   # the all-usable-cores run is the one reported; the other stays in the log
    return best


ALL_CORES_LOG = []


def _all_cores_runs():
    """The thread counts timed by the last _all_cores_s (all usable cores first) with their wall times."""
    runs = list(ALL_CORES_LOG)
    ALL_CORES_LOG.clear()
    return runs


def cpu_baseline_c1():
    """BASELINE config 1 IS the reference's CPU path (examples/single_thread_bpsk.rs:15-52 made deterministic,
    SURVEY 8d): PRBS7 -> BPSK -> PulseNode(rrc_taps(63, 4, 0.25), 4) -> Mixer, 2^18 symbols = 2^20 samples, run in
    full on one thread; beside it the literal example (32 taps, zero-stuff, batch_fir, x8192 -> i16, no mixer)."""
    import oracle

    nsym = 1 << 18
    bits, _ = oracle.prns_u8(0xC0, 0x01, nsym)
    sym = (bits.astype(np.float32) * 2.0 - 1.0).astype(np.complex64)
    taps = oracle.rrc_taps(63, 4.0, 0.25)

    def chain():
        oracle.Mixer(0.0, MIX_DPHASE).mix(oracle.pulse(sym, taps, 4, oracle.default_state(taps)))

    dt = _best_of(chain)
    taps32 = oracle.rrc_taps(32, 4.0, 0.25)

    def literal():  # single_thread_bpsk.rs:24-47, 4096-symbol blocks, one state across blocks
        st = oracle.default_state(taps32)
        for i in range(0, nsym, 4096):
            oracle.iq_c32_to_i16(oracle.batch_fir(oracle.upsample(sym[i:i + 4096], 4), taps32, st), 8192.0)

    dl = _best_of(literal)
    # (ii) PulseNode and MixerNode on their own threads, symbol blocks of 2^14 as messages; (iii) all cores
    pst, pmx = oracle.default_state(taps), oracle.Mixer(0.0, MIX_DPHASE)
    blk = 1 << 14
    dp = _pipeline_s([lambda v: oracle.pulse(v, taps, 4, pst), lambda v: pmx.mix(v)], [sym[i:i + blk] for i in range(0, nsym, blk)])

    def chunk(i, threads):
        m = nsym // threads
        oracle.Mixer(0.0, MIX_DPHASE).mix(oracle.pulse(sym[i * m:(i + 1) * m], taps, 4, oracle.default_state(taps)))

    da, threads = _all_cores_s(chunk)
    return {"value": round(4 * nsym / dt / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "thread_per_node": {"value": round(4 * nsym / dp / 1e6, 3), "unit": "Msamples/s", "cores": 2},
            "all_cores": {"value": round(4 * (nsym // threads) * threads / da / 1e6, 2), "unit": "Msamples/s", "cores": threads, "runs": _all_cores_runs()},
            "sample": "the whole config: 2^18 PRBS7 symbols -> BPSK -> oracle PulseNode (63-tap RRC, 4 samples per "
                      "symbol: fir() per output sample, rotate_right each) -> Mixer::mix (f64), 2^20 output samples, "
                      "best of 2, 1 thread of %d host cores" % (os.cpu_count() or 0),
            "literal_example": {"value": round(4 * nsym / dl / 1e6, 3), "unit": "Msamples/s", "cores": 1,
                                "what": "single_thread_bpsk.rs as written: 4096-symbol blocks, zero-stuff x4, "
                                        "batch_fir(32-tap RRC), x8192 -> i16; no mixer"}}


def cpu_baseline_c3():
    """examples/fm_radio.rs:144-164 shaped as BASELINE config 3: Mixer::mix -> batch_fir(127) -> decimate(8) ->
    FM::demod on a 2^25-sample piece of the same stream (the rate does not depend on the length)."""
    import oracle
    from comms_rs_amd import synth_iq

    n = 1 << 25
    x = synth_iq(n, 0, SEED)
    taps = lowpass_taps(C3_TAPS, 1.0 / 16)

    def chain():
        y = oracle.batch_fir(oracle.Mixer(0.0, C3_DPHASE).mix(x), taps, oracle.default_state(taps))
        oracle.FM().demod(oracle.decimate(y, C3_RATE))

    dt = _best_of(chain)
    # (ii) the four nodes on their own threads (messages of 2^18 samples: the radio block of examples/fm_radio.rs:144)
    mx, st, fm = oracle.Mixer(0.0, C3_DPHASE), oracle.default_state(taps), oracle.FM()
    blk = 1 << 18
    dp = _pipeline_s([lambda v: mx.mix(v), lambda v: oracle.batch_fir(v, taps, st), lambda v: oracle.decimate(v, C3_RATE),
                      lambda v: fm.demod(v)], [x[i:i + blk] for i in range(0, n, blk)])

    def chunk(i, threads):
        m = n // threads // C3_RATE * C3_RATE
        y = oracle.batch_fir(oracle.Mixer(0.0, C3_DPHASE).mix(x[i * m:(i + 1) * m]), taps, oracle.default_state(taps))
        oracle.FM().demod(oracle.decimate(y, C3_RATE))

    da, threads = _all_cores_s(chunk)
    # the literal example (examples/fm_radio.rs:144-152 with its own 63 taps, tests/golden/reference_kats.json):
    # u8 -> (x - 127.5) / 127.5 -> 63-tap FIR -> /5 -> FM demod -> (re, 0) -> 63-tap FIR -> .re -> /5, on 2^22 samples
    t63 = fm_radio_taps()
    nl = 1 << 22
    u8 = synth_u8(nl)

    def literal():
        a = oracle.FM().demod(oracle.decimate(oracle.batch_fir(oracle.iq_u8_to_c32(u8), t63, oracle.default_state(t63)), 5))
        oracle.decimate(oracle.batch_fir(a.astype(np.complex64), t63, oracle.default_state(t63)).real.copy(), 5)

    dl = _best_of(literal, 1)
    return {"value": round(n / dt / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "thread_per_node": {"value": round(n / dp / 1e6, 3), "unit": "Msamples/s", "cores": 4},
            "all_cores": {"value": round((n // threads // C3_RATE * C3_RATE) * threads / da / 1e6, 2), "unit": "Msamples/s",
                          "cores": threads, "runs": _all_cores_runs()},
            "literal_example": {"value": round(nl / dl / 1e6, 3), "unit": "Msamples/s", "cores": 1,
                                "what": "examples/fm_radio.rs as written (its 63 taps, /5, FM demod, 63 taps, /5; RTL-SDR "
                                        "bytes in) on 2^22 input samples, 1 thread"},
            "sample": "2^25 of the config's 2^26 samples (extrapolates linearly: per-sample work is constant) through "
                      "oracle Mixer::mix (f64) -> batch_fir(127 taps, rotate_right per sample) -> decimate(8) -> "
                      "FM::demod, best of 2, 1 thread of %d host cores" % (os.cpu_count() or 0)}


def cpu_baseline_c4():
    """src/fft/mod.rs:73-96: cast to f64, unnormalised 2^20-point DFT in f64, cast back -- 64 of the 4096 transforms."""
    import oracle
    from comms_rs_amd import synth_iq

    k = 64
    xs = [synth_iq(FFT_N, i * FFT_N, SEED) for i in range(k)]

    def run():
        for x in xs:
            oracle.fft(x, False)

    dt = _best_of(run)

    def chunk(i, threads):
        for x in xs[i::threads]:   # whole transforms per thread
            oracle.fft(x, False)

    da, threads = _all_cores_s(chunk, cap=k)
    return {"value": round(k * FFT_N / dt / 1e6, 3), "unit": "Mpoints/s", "cores": 1, "kind": "port",
            "all_cores": {"value": round(k * FFT_N / da / 1e6, 2), "unit": "Mpoints/s", "cores": threads, "runs": _all_cores_runs()},
            "thread_per_node": "one node: the one-thread figure",
            "sample": "64 of the config's 4096 transforms of 2^20 points (extrapolates linearly: transforms are "
                      "independent) through the oracle's BatchFFT::run_fft (f32 -> f64, radix-2 f64 FFT standing in "
                      "for rustfft 2.1.0, twiddles planned once per length as FFTplanner does, fft_node.rs:66-67; "
                      "f64 -> f32), best of 2, 1 thread of %d host cores" % (os.cpu_count() or 0)}


def cpu_baseline_c5():
    """src/filter/fir.rs:87-102 with 4097 taps: 2^21 samples, literal (rotate_right of 32 KiB per sample)."""
    import oracle
    from comms_rs_amd import synth_iq

    n, m = 1 << 21, 1 << 19
    x = synth_iq(n, 0, SEED)
    taps = lowpass_taps(C5_TAPS, 1.0 / 64)
    dt = _best_of(lambda: oracle.batch_fir(x, taps, oracle.default_state(taps)), 1)
    dn = _best_of(lambda: oracle.batch_fir(x[:m], taps, oracle.default_state(taps), norotate=True), 1) * (n / m)

    def chunk(i, threads):
        q = m // 4
        oracle.batch_fir(x[i * q % n:i * q % n + q], taps, oracle.default_state(taps))

    da, threads = _all_cores_s(chunk)
    return {"value": round(n / dt / 1e6, 4), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "all_cores": {"value": round((m // 4) * threads / da / 1e6, 3), "unit": "Msamples/s", "cores": threads, "runs": _all_cores_runs(),
                          "what": "2^17 samples per thread, literal batch_fir"},
            "thread_per_node": "one node: the one-thread figure",
            "sample": "2^21 of the config's 2^27 samples per GPU (extrapolates linearly: 4097 MACs + a 32 KiB "
                      "rotate_right per sample whatever the length) through the oracle's literal batch_fir, "
                      "1 thread of %d host cores" % (os.cpu_count() or 0),
            "one_core_norotate": {"value": round(n / dn / 1e6, 4), "unit": "Msamples/s", "cores": 1,
                                  "what": "circular index instead of rotate_right: not the reference's algorithm"}}


class Ctx:
    """Rank, device, process group and the timed-loop helper shared by every config."""

    def __init__(self, args):
        # before anything initialises the GPU runtime: the host driver supports dmabuf IPC only, and RCCL /
        # cross-process device memory fails without this (it is exported on the GPU boxes already)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.args = torch, dist, args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        assert self.world == args.gpus, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, self.world)
        ndev = torch.cuda.device_count()
        assert ndev >= 1, "no MI355X visible (no CPU fallback)"
        if args.backend == "gloo":
            local_rank = local_rank % ndev      # rehearsal: ranks may share a GPU
        assert local_rank < ndev, "LOCAL_RANK %d but only %d GPU(s) visible" % (local_rank, ndev)
        self.local_rank = local_rank
        torch.cuda.set_device(local_rank)
        if self.world > 1:
            if args.backend == "nccl":  # RCCL over xGMI
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world,
                                        device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
        self.dev = torch.device("cuda", local_rank)
        self.comm_dev = self.dev if args.backend == "nccl" else torch.device("cpu")
        self.stream = torch.cuda.current_stream().cuda_stream

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def max_over_ranks(self, v):
        if self.world > 1:
            t = self.torch.tensor([v], dtype=self.torch.float64, device=self.comm_dev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            return float(t.item())
        return v

    def over_ranks(self, v):
        """The value of `v` on every rank, as a list indexed by rank (all ranks get it)."""
        if self.world == 1:
            return [float(v)]
        t = self.torch.tensor([float(v)], dtype=self.torch.float64, device=self.comm_dev)
        parts = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(parts, t)
        return [float(p.item()) for p in parts]

    def world_size_seen(self):
        """What the process group itself reports (1 when no group was needed)."""
        return self.dist.get_world_size() if self.world > 1 else 1

    def all_ok(self, ok, what):
        """Fail loudly on every rank when any rank's self-check failed."""
        if self.world > 1:
            t = self.torch.tensor([1.0 if ok else 0.0], dtype=self.torch.float64, device=self.comm_dev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
            ok = t.item() == 1.0
        assert ok, "self-check failed on some rank: %s" % what

    def timed(self, step, steps, warmup):
        """warmup untimed steps, then exactly `steps` steps between barrier + synchronize on
        both sides; returns the MAX over ranks of the elapsed seconds."""
        for _ in range(warmup):
            step()
        self.barrier()
        self.torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        self.torch.cuda.synchronize()
        self.barrier()
        return self.max_over_ranks(time.perf_counter() - t0)

    def timed_transfer(self, fn):
        self.barrier()
        self.torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        self.torch.cuda.synchronize()
        self.barrier()
        return self.max_over_ranks(time.perf_counter() - t0)

    def comm_view(self, t):
        """Real-view tensor of `t` on the device the backend moves (GPU for nccl, CPU for gloo)."""
        v = self.torch.view_as_real(t) if t.is_complex() else t
        return v if self.comm_dev.type == "cuda" else v.cpu()

    def fill_shard(self, x, first_index, transfer):
        """This rank's input shard: generated in place from the counter-based generator (primary
        variant), or -- variant "scatter" -- generated whole on the root and sent shard by shard."""
        import comms_rs_amd as c
        from comms_rs_amd.sharding import scatter_shards

        n = x.numel()
        if self.args.variant != "scatter" or self.world == 1:
            c.synth_iq_dev(x.data_ptr(), n, first_index, SEED, device=self.local_rank, stream=self.stream)
            return
        torch = self.torch
        full = None
        if self.rank == 0:
            full = torch.empty(n * self.world, dtype=torch.complex64, device=self.dev)
            c.synth_iq_dev(full.data_ptr(), full.numel(), first_index, SEED, device=self.local_rank, stream=self.stream)
            torch.cuda.synchronize()
            full = self.comm_view(full)
        shard = self.comm_view(x)

        def go():
            scatter_shards(self.dist, full, shard, self.rank, self.world)

        dt = self.timed_transfer(go)
        if shard.device != x.device:
            torch.view_as_real(x).copy_(shard)
        transfer["scatter_ms"] = round(dt * 1e3, 3)
        transfer["scatter_GBps"] = round(8.0 * n * (self.world - 1) / dt / 1e9, 2)
        del full

    def collect(self, out, transfer):
        """Variant "scatter": every rank's output shard back to the root (timed)."""
        from comms_rs_amd.sharding import gather_shards

        if self.args.variant != "scatter" or self.world == 1:
            return
        torch = self.torch
        shard = self.comm_view(out)
        full = None
        if self.rank == 0:
            full = torch.empty((shard.shape[0] * self.world,) + tuple(shard.shape[1:]), dtype=shard.dtype,
                               device=shard.device)

        def go():
            gather_shards(self.dist, shard, full, self.rank, self.world)

        dt = self.timed_transfer(go)
        transfer["gather_ms"] = round(dt * 1e3, 3)
        transfer["gather_GBps"] = round(out.element_size() * out.numel() * (self.world - 1) / dt / 1e9, 2)
        del full


TIMER_STRIDE = 4
# The kernel timer of the timed regions holds BOTH of its instruments (comms_timer_*):
#  * hipEvent pairs around every TIMER_STRIDE-th launch of the dominant kernel (hipExtLaunchKernelGGL start / stop
#    events: the kernel's own begin / end as the command processor stamps them) -> `roofline.kernel_ms`.  Every 4th only,
#    because a pair idles the stream 6.5 + 4.7 us around the launch it brackets (rocprofv3 trace of the driver's 20-step
#    command, profiles/r03_timer_gaps_trace.txt): with every launch bracketed the region ran 5-9 % slower than the
#    loop it measures.  A bracketed launch therefore runs ISOLATED from its neighbours by those gaps.
#  * in-kernel stamps on EVERY launch (first workgroup in -> last wave's stores acknowledged, s_memrealtime; ordinary
#    launches, nothing idles) -> `roofline.kernel_ms_in_stream`: the kernel back to back with its neighbours in the
#    stream, which is what rocprofv3's --kernel-trace average over all launches sees as well.


def kernel_times(timer):
    """(event-timed ms per bracketed launch, stamped in-stream ms per launch) of a 'both' timer."""
    kms = timer.read_ms()
    sms = timer.read_stamps_ms()
    return kms, sms[sms > 0]


def in_stream_fields(sms):
    if not sms.size:
        return {"kernel_ms_in_stream": None, "launches_stamped": 0}
    return {"kernel_ms_in_stream": round(float(np.mean(sms)), 5), "launches_stamped": int(sms.size),
            "kernel_ms_in_stream_min_max": [round(float(sms.min()), 5), round(float(sms.max()), 5)]}


def rank_report(ctx, kernel_ms):
    """Collective: what the process group saw -- its own world size and every rank's mean kernel time."""
    per = ctx.over_ranks(kernel_ms)
    return {"world_size_seen": ctx.world_size_seen(), "backend": ctx.args.backend if ctx.world > 1 else None,
            "kernel_ms_per_rank": [round(v, 5) for v in per], "kernel_ms_min": round(min(per), 5),
            "kernel_ms_max": round(max(per), 5)}


def fir_chain_pass(ctx, n, first_index, steps, warmup, algo="auto", stride=TIMER_STRIDE):
    """The metric's chain, node by node, over this rank's shard [first_index, first_index + n) of the
    stream: 255-tap BatchFirNode -> MixerNode -> DecimateNode.  Returns timings + the nodes' facts."""
    import comms_rs_amd as c
    from comms_rs_amd.sharding import halo_exchange, shard_mixer_phase, state_from_halo

    torch, rank, world = ctx.torch, ctx.rank, ctx.world
    taps = c.rrc_taps(N_TAPS, 8.0, 0.35)
    x = torch.empty(n, dtype=torch.complex64, device=ctx.dev)
    y = torch.empty(n, dtype=torch.complex64, device=ctx.dev)
    z = torch.empty(n // DEC_RATE, dtype=torch.complex64, device=ctx.dev)
    transfer = {}
    ctx.fill_shard(x, first_index if ctx.args.variant != "scatter" else 0, transfer)

    fir = c.BatchFirNode(taps, device=ctx.local_rank)
    if algo != "auto":
        fir.set_algo({"direct": c.FIR_DIRECT, "os1024": c.FIR_OS1024, "os4096": c.FIR_OS4096}[algo])
    # every rank continues the un-sharded oscillator: closed-form phase of its first sample
    mix_phase = shard_mixer_phase(0.0, MIX_DPHASE, first_index)
    mixer = c.MixerNode(MIX_DPHASE, mix_phase, device=ctx.local_rank)
    dec = c.DecimateNode(DEC_RATE, device=ctx.local_rank)

    # ---- one-off halo hand-over to the right-hand neighbour (RCCL send/recv)
    h = None
    if world > 1:
        tail = ctx.comm_view(x[n - N_TAPS:].clone())
        halo = halo_exchange(ctx.dist, tail, rank, world)
        torch.cuda.synchronize()
        if rank > 0:
            h = torch.view_as_complex(halo.contiguous()).cpu().numpy()
            fir.set_state(state_from_halo(h))
        ok = rank == 0 or np.array_equal(h, c.synth_iq(N_TAPS, first_index - N_TAPS, SEED))
        ctx.all_ok(ok, "halo differs from the stream's samples before the shard")

    s = ctx.stream

    def step():
        fir.run_dev(x.data_ptr(), n, y.data_ptr(), s)
        mixer.run_dev(y.data_ptr(), n, y.data_ptr(), s)
        dec.run_dev(y.data_ptr(), n, 8, z.data_ptr(), s)

    # ---- self-check of the first step (fail loudly): the node-by-node chain (FFT overlap-save
    # FIR, then mixer, then decimate) against the fused chain node, an independent time-domain
    # kernel, on the same shard.  (Parity with the reference's arithmetic is the job of tests/
    # and smoke(); nothing on this path touches the CPU checker.)
    step()
    chk = c.ChainNode(MIX_DPHASE, mix_phase, taps, DEC_RATE, False, device=ctx.local_rank, mixer_after_fir=True)
    if h is not None:
        chk.set_fir_state(state_from_halo(h))
    zc = torch.empty_like(z)
    chk.run_dev(x.data_ptr(), n, zc.data_ptr(), s)
    torch.cuda.synchronize()
    bound = 2e-5 * float(np.sum(np.abs(taps)))
    err = float((z - zc).abs().max())
    ok = err <= bound
    if ok and rank > 0 and ctx.args.variant != "scatter":
        # the shard boundary itself: a fresh chain run over [prefix | first outputs] with no hand-over at
        # all must agree, past its start-up transient, with what the halo-primed nodes produced
        P, M = 512, 4096
        xb = torch.empty(P + M, dtype=torch.complex64, device=ctx.dev)
        c.synth_iq_dev(xb.data_ptr(), P + M, first_index - P, SEED, device=ctx.local_rank, stream=s)
        zb = torch.empty((P + M) // DEC_RATE, dtype=torch.complex64, device=ctx.dev)
        c.ChainNode(MIX_DPHASE, shard_mixer_phase(0.0, MIX_DPHASE, first_index - P), taps, DEC_RATE, False,
                    device=ctx.local_rank, mixer_after_fir=True).run_dev(xb.data_ptr(), P + M, zb.data_ptr(), s)
        torch.cuda.synchronize()
        err = max(err, float((zb[P // DEC_RATE:] - z[:M // DEC_RATE]).abs().max()))
        ok = err <= bound
    ctx.all_ok(ok, "chain vs fused chain / shard boundary differ by %g (bound %g)" % (err, bound))
    del chk, zc

    # Clock settling in front of the W warm-up steps (short batches only): a burst of launches runs its first ~2 ms at boost clocks, the
    # next ~20 ms up to 40 % slower, then settles (profiles/r05_config3_poly8_launch_order.txt, r03_config5_kernel_trace.txt).  The
    # buffer set-up and the consistency checks above idle the GPU long enough to restart that; K = 20 steps of 0.12 ms behind W = 5
    # would be timed inside the slow stretch (0.57-0.66 from run to run on one box).  The same step, untimed, for `settle_ms` first.
    settle_steps = 0
    if ctx.args.settle_ms > 0 and n <= (1 << 26):
        t_end = time.time() + ctx.args.settle_ms * 1e-3
        while time.time() < t_end:
            for _ in range(20):
                step()
            torch.cuda.synchronize()
            settle_steps += 20
    for _ in range(warmup):
        step()
    timer = c.KernelTimer(max(steps, 1), device=ctx.local_rank, stamps="both", stride=stride).attach(fir)
    elapsed = ctx.timed(step, steps, 0)
    kms, sms = kernel_times(timer)
    timer.close()
    ctx.collect(z, transfer)

    # ---- the same chain as ONE fused node (comms_chain_*: additional node, same results)
    chain = c.ChainNode(MIX_DPHASE, mix_phase, taps, DEC_RATE, False, device=ctx.local_rank, mixer_after_fir=True)
    zf = torch.empty_like(z)
    # (its own step count: 20 steps of 28 us would sit inside the chip's start-up clock transient; the stream leg keeps its few passes)
    fsteps = steps if n > (1 << 26) else max(steps, 200)
    fused_elapsed = ctx.timed(lambda: chain.run_dev(x.data_ptr(), n, zf.data_ptr(), s), fsteps, max(warmup, 20 if n <= (1 << 26) else 1))
    fused_elapsed *= float(steps) / fsteps   # (scaled to `steps` steps: the callers divide by that)
    kernel_ms = float(np.mean(kms)) if kms.size else float("nan")
    ceiling = copy_ceiling(torch, x, y)  # (behind every timed region)
    res = {"elapsed": elapsed, "fused_elapsed": fused_elapsed, "ranks": rank_report(ctx, kernel_ms), "ceiling": ceiling,
           "kernel_ms": kernel_ms, "launches_timed": int(kms.size), "kernel_ms_each": [round(float(v), 5) for v in kms],
           "in_stream": in_stream_fields(sms), "in_stream_each": [round(float(v), 5) for v in sms],
           "timer_stride": stride, "algo": fir.kernel_for(n), "fused": chain.fused, "fused_kernel": chain.kernel, "transfer": transfer,
           "settle_steps": settle_steps}
    del x, y, z, zf, fir, mixer, chain
    torch.cuda.empty_cache()
    return res


def host_vec_leg(ctx, n):
    """The drop-in path of an unchanged comms-rs graph (reported separately, never `value`): host memory in,
    host memory out -- run(&[Complex<f32>]) -> Vec<Complex<f32>> (src/filter/fir_node.rs:215-220).  The metric's chain over
    2^24 samples through the host-pointer entries, node by node (comms_fir_run -> comms_mixer_run -> comms_decimate_run)
    and as one comms_chain_run; PCIe both ways is inside the time.  Two forms of the caller: `fresh_out` allocates the
    output array in every call, as run() returning a fresh Vec does (a fresh 128-MiB allocation is page-faulted in while
    the copy lands in it); `reused_out` hands the same, touched, buffers to every call.  Long calls are pipelined in
    chunks by the library (csrc/common.hpp, Handle::run_host_units)."""
    import ctypes as C

    import comms_rs_amd as c
    from comms_rs_amd._lib import check

    taps = c.rrc_taps(N_TAPS, 8.0, 0.35)
    x = c.synth_iq(n, 0, SEED)
    lib = c.lib()
    res = {"samples": n, "unit": "Msamples/s"}

    def best(fn, reps=4):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return min(ts)

    fir = c.BatchFirNode(taps, device=ctx.local_rank)
    mixer = c.MixerNode(MIX_DPHASE, 0.0, device=ctx.local_rank)
    dec = c.DecimateNode(DEC_RATE, device=ctx.local_rank)
    chain = c.ChainNode(MIX_DPHASE, 0.0, taps, DEC_RATE, False, device=ctx.local_rank, mixer_after_fir=True)
    # fresh output per call (the wrappers' run(): np.empty per call, like a Vec returned by value)
    t_nodes_fresh = best(lambda: dec.run(mixer.run(fir.run(x))))
    t_chain_fresh = best(lambda: chain.run(x))
    # the same entries with caller-owned buffers reused over the calls
    y = np.empty(n, np.complex64)
    z = np.empty(n // DEC_RATE, np.complex64)
    y.fill(0)
    z.fill(0)
    m = C.c_size_t()
    vp = lambda a: a.ctypes.data_as(C.c_void_p)

    def nodes_reused():
        check(lib.comms_fir_run(fir._h, vp(x), n, vp(y)))
        check(lib.comms_mixer_run(mixer._h, vp(y), n, vp(y)))
        check(lib.comms_decimate_run(vp(y), n, 8, DEC_RATE, vp(z), C.byref(m), ctx.local_rank))

    t_nodes = best(nodes_reused)
    t_chain = best(lambda: check(lib.comms_chain_run(chain._h, vp(x), n, vp(z))))
    t_fir = best(lambda: check(lib.comms_fir_run(fir._h, vp(x), n, vp(y))))
    res.update({
        "node_by_node": {"value": round(n / t_nodes / 1e6, 1), "ms": round(t_nodes * 1e3, 3),
                         "fresh_out_value": round(n / t_nodes_fresh / 1e6, 1),
                         "pcie_bytes_per_sample": "8 + 8 (FIR) + 8 + 8 (mixer) + 8 + 1 (decimate): every node takes host memory"},
        "fused_chain": {"value": round(n / t_chain / 1e6, 1), "ms": round(t_chain * 1e3, 3),
                        "fresh_out_value": round(n / t_chain_fresh / 1e6, 1), "GBps_in": round(8.0 * n / t_chain / 1e9, 1),
                        "pcie_bytes_per_sample": "8 in + 1 out"},
        "fir_alone": {"value": round(n / t_fir / 1e6, 1), "ms": round(t_fir * 1e3, 3),
                      "GBps_each_way": round(8.0 * n / t_fir / 1e9, 1)},
        "pcie_note": "pageable host memory both ways; hipMemcpy on this box moves 56 GB/s one way and 46-48 GB/s each way "
                     "with both directions busy (profiles/r05_host_path.txt): the FIR node's 8 B in + 8 B out per sample "
                     "cannot pass ~5.9 Gsamples/s, the fused chain's 9 B ~6.2 Gsamples/s",
        "pipelined": "calls of 64 MiB and more whose input and output both carry bytes, in 16-MiB chunks: copy in + launch on the node's stream, copy out on a second "
                     "stream from a helper thread (csrc/common.hpp: Handle::run_host_units)"})
    return res


def run_config2(ctx):
    args, world = ctx.args, ctx.world
    n = 1 << args.n_log2 if args.n_log2 else N_SAMPLES
    # The 2^30-sample stream leg runs first: a few long passes (tens of ms of GPU work) also bring the clocks to
    # the state a continuously running pipeline sees, which the short headline region below (K steps of ~0.12 ms)
    # would otherwise spend partly in the ramp -- 20 steps straight after start-up measure 5-10 % slower than
    # 1000.  `--head-first` restores the old order for comparison.
    def stream_leg():
        total = 1 << args.stream_log2
        per = total // world
        s_steps = max(1, min(args.steps, 5))
        # every launch of this leg carries the timer's events: an 11-us gap is 0.3 % of a 3.5-ms launch
        st = fir_chain_pass(ctx, per, ctx.rank * per, s_steps, min(args.warmup, 2), args.algo, stride=1)
        st.update(total=total, per=per, steps=s_steps)
        return st

    # The host-pointer leg runs FIRST, on a process that has allocated nothing big yet: behind the 2^30-sample leg (16 GiB of
    # device buffers allocated and released again) the same pageable copies run at half the rate (FIR node 5.2 -> 2.8
    # Gsamples/s on one box, `profiles/r05_host_pipeline.txt`), which says something about that process state, not about
    # the path.  It touches the GPU for a few milliseconds: the legs below are not affected.
    host_vec = host_vec_leg(ctx, min(n, N_SAMPLES)) if (ctx.rank == 0 and not args.no_host_vec) else None
    stream = None
    if args.stream_log2 and not args.head_first:
        stream = stream_leg()
    head = fir_chain_pass(ctx, n, ctx.rank * n, args.steps, args.warmup, args.algo)
    if args.stream_log2 and args.head_first:
        stream = stream_leg()
    if ctx.rank != 0:
        return None
    kernel_ms, algo = head["kernel_ms"], head["algo"]
    total = float(world) * n * args.steps
    achieved = FIR_BYTES_PER_SAMPLE * n / (kernel_ms * 1e-3) / 1e9
    out = {
        "metric": "Msamples/s Complex<f32> through 255-tap FIR->mix->decimate chain",
        "value": round(total / head["elapsed"] / 1e6, 1),
        "unit": "Msamples/s",
        "n_gpus": world,
        "world_size_seen": head["ranks"]["world_size_seen"],
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(head["elapsed"] / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "BASELINE config 2: 255-tap BatchFirNode (rrc_taps(255,8,0.35)) -> mixer "
                               "(dphase 2pi*0.1) -> decimate-by-8 on 2^%d Complex<f32> IQ per GPU, "
                               "node by node, device-resident" % int(np.log2(n)),
                   "samples_per_gpu_per_step": n, "n_taps": N_TAPS, "dec_rate": DEC_RATE,
                   "fir_kernel": algo, "sharding": "contiguous stream shards, one-off RCCL halo",
                   "variant": args.variant, "backend": args.backend,
                   "leg_order": "headline first" if (args.head_first or not args.stream_log2) else
                                "2^%d-sample stream leg first (clock warm-up), then the headline" % args.stream_log2,
                   "settle": "%d untimed steps (%g ms) in front of the W warm-up steps: the chip's start-up clock transient" % (head["settle_steps"], args.settle_ms)},
        "roofline": {"bound": "hbm", "kernel": algo, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": pmc_traffic(algo, n),
                     "kernel_ms": round(kernel_ms, 5), "launches_timed": head["launches_timed"], "timer_stride": TIMER_STRIDE,
                     **head["in_stream"],
                     "frac_in_stream": (round(FIR_BYTES_PER_SAMPLE * n / (head["in_stream"]["kernel_ms_in_stream"] * 1e-3) / 1e9
                                              / HBM_PEAK_GBS, 4) if head["in_stream"]["kernel_ms_in_stream"] else None),
                     "algorithmic_bytes_per_launch": FIR_BYTES_PER_SAMPLE * n, "ceiling": head["ceiling"]},
        "ranks": head["ranks"],
        "box": box_info(ctx.torch, ctx.local_rank),
    }
    fused_ms = head["fused_elapsed"] / args.steps * 1e3
    fused_bytes = 9.0 * n   # 8 B read + 8/8 B written per input sample
    out["fused_chain"] = {"value": round(total / head["fused_elapsed"] / 1e6, 1), "unit": "Msamples/s",
                          "ms_per_step": round(fused_ms, 4), "steps": max(args.steps, 200), "fused": head["fused"],
                          "kernel": {"time": "fir_decim_kernel", "freq": "fir_os1024_kernel<.., MODE>", "poly": "fir_poly8_kernel",
                                     "unfused": "four kernels"}[head["fused_kernel"]],
                          "roofline": {"bound": "hbm", "algorithmic_bytes_per_launch": fused_bytes,
                                       "achieved": round(fused_bytes / (fused_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": round(fused_bytes / (fused_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                       "traffic": pmc_traffic({"poly": "fir_poly8_kernel"}.get(head["fused_kernel"], "fir_decim_kernel") + ":fused_chain", n),
                                       "clock": "step period of the timed loop (one launch per step, back to back)"},
                          "note": "same FIR->mixer->decimate chain as one comms_chain_* launch "
                                  "(8 B read + 1 B written per input sample); not the headline value"}
    if host_vec is not None:
        out["host_vec"] = host_vec
    if stream is not None:
        st = stream
        ach = FIR_BYTES_PER_SAMPLE * st["per"] / (st["kernel_ms"] * 1e-3) / 1e9
        out["stream_2p30"] = {
            "workload": "north-star stream: 2^%d samples cut into %d contiguous shards of %d samples per rank, the "
                        "same node-by-node chain, %d timed passes" % (int(np.log2(st["total"])), world, st["per"], st["steps"]),
            "scaling": "strong", "value": round(st["total"] * st["steps"] / st["elapsed"] / 1e6, 1), "unit": "Msamples/s",
            "ms_per_pass": round(st["elapsed"] / st["steps"] * 1e3, 4),
            "fused_chain_value": round(st["total"] * st["steps"] / st["fused_elapsed"] / 1e6, 1),
            "fir_kernel": st["algo"], "fir_kernel_ms": round(st["kernel_ms"], 5),
            "fir_kernel_ms_each": st["kernel_ms_each"], "timer_stride": st["timer_stride"],
            "fir_kernel_ms_in_stream": st["in_stream"]["kernel_ms_in_stream"], "fir_kernel_ms_in_stream_each": st["in_stream_each"],
            "fir_hbm_GBps": round(ach, 1), "fir_frac_of_peak": round(ach / HBM_PEAK_GBS, 4), "ceiling": st["ceiling"],
            "fir_kernel_ms_per_rank": st["ranks"]["kernel_ms_per_rank"],
            "note": "%.2f GiB of FIR input + output per GPU (HBM-resident once this is far past the 256 MiB "
                    "Infinity Cache)" % (16.0 * st["per"] / 2 ** 30)}
        if st["transfer"]:
            out["stream_2p30"]["transfer"] = st["transfer"]
    if head["transfer"]:
        out["transfer"] = head["transfer"]
    if args.n_log2:
        out["rehearsal"] = "per-GPU size overridden (--n-log2): not a BASELINE-config number"
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(n)
    return out


def run_config3(ctx):
    """mixer -> 127-tap LPF -> decimate-by-8 -> FM demod on 2^26 samples per rank, one fused launch.
    A shard is primed by the prefix that precedes it (FIR halo + the sample FM.prev comes from)."""
    import comms_rs_amd as c
    from comms_rs_amd.sharding import chain_prefix_len, halo_exchange, prime_chain, shard_mixer_phase

    args, torch, rank, world = ctx.args, ctx.torch, ctx.rank, ctx.world
    n = 1 << (args.n_log2 or 26)
    first = rank * n
    taps = lowpass_taps(C3_TAPS, 1.0 / 16)
    x = torch.empty(n, dtype=torch.complex64, device=ctx.dev)
    out = torch.empty(n // C3_RATE, dtype=torch.float32, device=ctx.dev)
    transfer = {}
    ctx.fill_shard(x, first if args.variant != "scatter" else 0, transfer)
    W = chain_prefix_len(C3_TAPS, C3_RATE, True)
    s = ctx.stream

    def make_chain(start):
        return c.ChainNode(C3_DPHASE, shard_mixer_phase(0.0, C3_DPHASE, start), taps, C3_RATE, True,
                           device=ctx.local_rank)

    chain = make_chain(first - W if (world > 1 and rank > 0) else first)
    if world > 1:
        tail = ctx.comm_view(x[n - W:].clone())
        halo = halo_exchange(ctx.dist, tail, rank, world)
        torch.cuda.synchronize()
        ok = True
        if rank > 0:
            pre = torch.view_as_complex(halo.contiguous()).to(ctx.dev)
            ok = np.array_equal(pre.cpu().numpy(), c.synth_iq(W, first - W, SEED))
            scratch = torch.empty(W // C3_RATE, dtype=torch.float32, device=ctx.dev)
            prime_chain(chain, pre.data_ptr(), W, scratch.data_ptr(), s)
            torch.cuda.synchronize()
        ctx.all_ok(ok, "prefix differs from the stream's samples before the shard")
    chain.run_dev(x.data_ptr(), n, out.data_ptr(), s)
    torch.cuda.synchronize()
    # self-check at the shard boundary (ranks > 0): a fresh chain over [long prefix | first outputs]
    ok, err = True, 0.0
    if rank > 0 and args.variant != "scatter":
        P, M = 1024, 8192
        xb = torch.empty(P + M, dtype=torch.complex64, device=ctx.dev)
        c.synth_iq_dev(xb.data_ptr(), P + M, first - P, SEED, device=ctx.local_rank, stream=s)
        ob = torch.empty((P + M) // C3_RATE, dtype=torch.float32, device=ctx.dev)
        make_chain(first - P).run_dev(xb.data_ptr(), P + M, ob.data_ptr(), s)
        torch.cuda.synchronize()
        d = (ob[P // C3_RATE:] - out[:M // C3_RATE]).abs()
        d = torch.minimum(d, (2 * np.pi - d).abs())
        # noise input: |y| can be tiny, where the angle is ill-conditioned -> judge the bulk, bound the rest
        err = float(d.median())
        ok = err <= 1e-5 and float((d > 1e-2).float().mean()) < 1e-3
    ctx.all_ok(ok, "config 3 shard boundary: median |d angle| %g" % err)

    for _ in range(args.warmup):
        chain.run_dev(x.data_ptr(), n, out.data_ptr(), s)
    timer = c.KernelTimer(max(args.steps, 1), device=ctx.local_rank, stamps="both", stride=TIMER_STRIDE).attach(chain)
    elapsed = ctx.timed(lambda: chain.run_dev(x.data_ptr(), n, out.data_ptr(), s), args.steps, 0)
    kms, sms = kernel_times(timer)
    timer.close()
    ctx.collect(out, transfer)
    kernel_ms = float(np.mean(kms))
    ranks = rank_report(ctx, kernel_ms)
    # (a plain copy moving about as many bytes as the chain does: half the input onto itself's other half, 8 n of its 8.5 n bytes)
    ceiling = copy_ceiling(torch, x[: n // 2], x[n // 2:]) if args.variant != "scatter" else None
    # ---- the literal example beside it (examples/fm_radio.rs:144-152 with its own 63 taps): RTL-SDR bytes -> 63-tap FIR
    # -> /5 -> FM demod as ONE launch (u8 read by the kernel's load stage), then Convert2 -> 63-tap FIR -> Convert3 -> /5
    # as the example wires them, device-resident
    t63 = fm_radio_taps()
    nl = n // 25 * 25
    n1, n2 = nl // 5, nl // 25
    blk = synth_u8(1 << 20)
    u8 = torch.from_numpy(np.ascontiguousarray(np.tile(blk, ((nl >> 20) + 1, 1))[:nl])).to(ctx.dev)
    front = c.ChainNode(0.0, 0.0, t63, 5, True, device=ctx.local_rank)
    front.set_input_format("u8")
    fir2, dec2 = c.BatchFirNode(t63, device=ctx.local_rank), c.DecimateNode(5, device=ctx.local_rank)
    la = torch.empty(n1, dtype=torch.float32, device=ctx.dev)
    lb, lc = torch.empty(n1, dtype=torch.complex64, device=ctx.dev), torch.empty(n1, dtype=torch.complex64, device=ctx.dev)
    ld, le = torch.empty(n1, dtype=torch.float32, device=ctx.dev), torch.empty(n2, dtype=torch.float32, device=ctx.dev)

    def literal():
        front.run_dev(u8.data_ptr(), nl, la.data_ptr(), s)
        c.real_to_c32_dev(la.data_ptr(), n1, lb.data_ptr(), ctx.local_rank, s)
        fir2.run_dev(lb.data_ptr(), n1, lc.data_ptr(), s)
        c.c32_re_dev(lc.data_ptr(), n1, ld.data_ptr(), ctx.local_rank, s)
        dec2.run_dev(ld.data_ptr(), n1, 4, le.data_ptr(), s)

    lit_steps = max(1, args.steps // 4)
    lit_elapsed = ctx.timed(literal, lit_steps, max(1, args.warmup // 4))
    lit_kernel = front.kernel
    del u8, la, lb, lc, ld, le
    if rank != 0:
        return None
    ach = C3_BYTES_PER_SAMPLE * n / (kernel_ms * 1e-3) / 1e9
    c3_kernel = {"time": "fir_decim_kernel", "poly": "fir_poly8_kernel"}.get(chain.kernel, chain.kernel)  # (after the runs: what they ran on)
    res = {"metric": "Msamples/s Complex<f32> through mixer->127-tap FIR->decimate-by-8->FM demod (BASELINE config 3)",
           "value": round(float(world) * n * args.steps / elapsed / 1e6, 1), "unit": "Msamples/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "BASELINE config 3: examples/fm_radio.rs chain, mixer (dphase 2pi*0.05) -> 127-tap "
                                  "windowed-sinc LPF -> decimate-by-8 -> FM demod, 2^%d samples per GPU, one fused launch"
                                  % int(np.log2(n)),
                      "samples_per_gpu_per_step": n, "kernel": chain.kernel, "prefix_samples": W,
                      "variant": args.variant, "backend": args.backend},
           "roofline": {"bound": "hbm", "kernel": c3_kernel,
                        "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                        "traffic": pmc_traffic(c3_kernel, n),
                        "kernel_ms": round(kernel_ms, 5), "launches_timed": int(kms.size), "timer_stride": TIMER_STRIDE,
                        **in_stream_fields(sms), "algorithmic_bytes_per_launch": C3_BYTES_PER_SAMPLE * n, "ceiling": ceiling}}
    res["literal_example"] = {"value": round(float(world) * nl * lit_steps / lit_elapsed / 1e6, 1), "unit": "Msamples/s",
                              "ms_per_step": round(lit_elapsed / lit_steps * 1e3, 4), "input_samples_per_gpu_per_step": nl,
                              "front_kernel": lit_kernel,
                              "what": "examples/fm_radio.rs as written, device-resident: u8 -> [63-tap FIR -> /5 -> FM demod] (one "
                                      "launch) -> Convert2 -> 63-tap FIR -> Convert3 -> /5; 2 B read per input sample"}
    res["world_size_seen"], res["ranks"] = ranks["world_size_seen"], ranks
    if transfer:
        res["transfer"] = transfer
    if world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline_c3()
    return res


def run_config5(ctx):
    """4097-tap overlap-save FIR, 2^27 samples per rank (the 2^30 stream over 8 GPUs), 4096-sample halo."""
    import comms_rs_amd as c
    from comms_rs_amd.sharding import halo_exchange, state_from_halo

    args, torch, rank, world = ctx.args, ctx.torch, ctx.rank, ctx.world
    n = 1 << (args.n_log2 or 27)
    first = rank * n
    taps = lowpass_taps(C5_TAPS, 1.0 / 64)
    x = torch.empty(n, dtype=torch.complex64, device=ctx.dev)
    y = torch.empty(n, dtype=torch.complex64, device=ctx.dev)
    transfer = {}
    ctx.fill_shard(x, first if args.variant != "scatter" else 0, transfer)
    fir = c.BatchFirNode(taps, device=ctx.local_rank)
    s = ctx.stream
    if world > 1:
        tail = ctx.comm_view(x[n - C5_TAPS:].clone())
        halo = halo_exchange(ctx.dist, tail, rank, world)
        torch.cuda.synchronize()
        ok = True
        if rank > 0:
            h = torch.view_as_complex(halo.contiguous()).cpu().numpy()
            ok = np.array_equal(h, c.synth_iq(C5_TAPS, first - C5_TAPS, SEED))
            fir.set_state(state_from_halo(h))
        ctx.all_ok(ok, "halo differs from the stream's samples before the shard")
    fir.run_dev(x.data_ptr(), n, y.data_ptr(), s)
    torch.cuda.synchronize()
    # self-check: the first outputs against the 4096-point kernel (an independent implementation) run
    # over [prefix | first outputs] with no hand-over
    P, M = (2 * 4096 if rank > 0 and args.variant != "scatter" else 0), 1 << 16
    xb = torch.empty(P + M, dtype=torch.complex64, device=ctx.dev)
    if P:
        c.synth_iq_dev(xb.data_ptr(), P + M, first - P, SEED, device=ctx.local_rank, stream=s)
    else:
        xb.copy_(x[:M])
    yb = torch.empty_like(xb)
    chk = c.BatchFirNode(taps, device=ctx.local_rank).set_algo(c.FIR_OS4096)
    if not P and rank > 0:
        chk.set_state(state_from_halo(h))
    chk.run_dev(xb.data_ptr(), P + M, yb.data_ptr(), s)
    torch.cuda.synchronize()
    err = float((yb[P:] - y[:M]).abs().max())
    bound = 2e-5 * float(np.sum(np.abs(taps)))
    ctx.all_ok(err <= bound, "config 5: os16k vs os4096 at the shard start differ by %g (bound %g)" % (err, bound))
    del xb, yb, chk

    for _ in range(args.warmup):
        fir.run_dev(x.data_ptr(), n, y.data_ptr(), s)
    timer = c.KernelTimer(max(args.steps, 1), device=ctx.local_rank, stamps="both", stride=TIMER_STRIDE).attach(fir)
    elapsed = ctx.timed(lambda: fir.run_dev(x.data_ptr(), n, y.data_ptr(), s), args.steps, 0)
    kms, sms = kernel_times(timer)
    timer.close()
    ctx.collect(y, transfer)
    kernel_ms = float(np.mean(kms))
    ranks = rank_report(ctx, kernel_ms)
    ceiling = copy_ceiling(torch, x, y)
    if rank != 0:
        return None
    ach = FIR_BYTES_PER_SAMPLE * n / (kernel_ms * 1e-3) / 1e9
    res = {"metric": "Msamples/s Complex<f32> through the 4097-tap overlap-save FIR (BASELINE config 5)",
           "value": round(float(world) * n * args.steps / elapsed / 1e6, 1), "unit": "Msamples/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "BASELINE config 5: 4097-tap windowed-sinc BatchFirNode, 2^%d samples per GPU (the "
                                  "2^30-sample stream over 8 GPUs), 4096-sample halo from the left neighbour"
                                  % int(np.log2(n)),
                      "samples_per_gpu_per_step": n, "n_taps": C5_TAPS, "fir_kernel": fir.kernel_for(n),
                      "variant": args.variant, "backend": args.backend},
           "roofline": {"bound": "hbm", "kernel": fir.kernel_for(n), "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(fir.kernel_for(n), n),
                        "kernel_ms": round(kernel_ms, 5), "launches_timed": int(kms.size), "timer_stride": TIMER_STRIDE,
                        **in_stream_fields(sms), "algorithmic_bytes_per_launch": FIR_BYTES_PER_SAMPLE * n, "ceiling": ceiling}}
    res["world_size_seen"], res["ranks"] = ranks["world_size_seen"], ranks
    if transfer:
        res["transfer"] = transfer
    if world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline_c5()
    return res


def run_config4(ctx):
    """2^20-point forward FFT, batch 4096 cut into 4096/N transforms per rank; no communication."""
    import comms_rs_amd as c
    from comms_rs_amd.sharding import shard_range

    args, torch, rank, world = ctx.args, ctx.torch, ctx.rank, ctx.world
    batch_total = (1 << args.n_log2) // FFT_N * world if args.n_log2 else FFT_BATCH
    b0, b1 = shard_range(batch_total, world, rank)
    nb = b1 - b0
    n = nb * FFT_N
    x = torch.empty(n, dtype=torch.complex64, device=ctx.dev)
    y = torch.empty(n, dtype=torch.complex64, device=ctx.dev)
    transfer = {}
    ctx.fill_shard(x, b0 * FFT_N if args.variant != "scatter" else 0, transfer)
    fwd = c.FFTBatchNode(FFT_N, False, device=ctx.local_rank)
    inv = c.FFTBatchNode(FFT_N, True, device=ctx.local_rank)
    s = ctx.stream
    fwd.run_dev(x.data_ptr(), n, y.data_ptr(), s)
    # self-check on the first and last transform of the shard: Parseval + inverse round trip
    ok = True
    for t in (0, nb - 1):
        xt, yt = x[t * FFT_N:(t + 1) * FFT_N], y[t * FFT_N:(t + 1) * FFT_N]
        back = torch.empty_like(xt)
        inv.run_dev(yt.data_ptr(), FFT_N, back.data_ptr(), s)
        torch.cuda.synchronize()
        ex, ey = float((xt.abs() ** 2).sum(dtype=torch.float64)), float((yt.abs() ** 2).sum(dtype=torch.float64))
        rt = float((back / FFT_N - xt).abs().max())
        ok = ok and abs(ey / (FFT_N * ex) - 1.0) < 1e-5 and rt < 2e-5
    ctx.all_ok(ok, "config 4: Parseval / round trip")
    for _ in range(args.warmup):
        fwd.run_dev(x.data_ptr(), n, y.data_ptr(), s)
    # (every call bracketed: an event pair's ~11 us are nothing against a 27-ms call)
    timer = c.KernelTimer(max(args.steps, 1), device=ctx.local_rank, stamps="both").attach(fwd)
    elapsed = ctx.timed(lambda: fwd.run_dev(x.data_ptr(), n, y.data_ptr(), s), args.steps, 0)
    kms, sms = kernel_times(timer)
    timer.close()
    ctx.collect(y, transfer)
    kernel_ms = float(np.mean(kms))
    ranks = rank_report(ctx, kernel_ms)
    if rank != 0:
        return None
    ach = FFT_BYTES_PER_POINT * n / (kernel_ms * 1e-3) / 1e9
    # the two-pass ceiling, measured in this run: a 2^20-point transform does not fit a workgroup, so every point crosses
    # HBM twice each way; the plain copy of one launch group's footprint (what ONE pass would cost if it did nothing else)
    grp = min(n, (2048 << 20) // 8)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for a, b in ev:
        a.record()
        y[:grp].copy_(x[:grp])
        b.record()
    torch.cuda.synchronize()
    copy_ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
    copy_gbs = 16.0 * grp / (copy_ms * 1e-3) / 1e9
    ceiling = {"passes": 2, "copy_GBps": round(copy_gbs, 1), "copy_points": grp,
               "frac_at_copy_rate": round(copy_gbs / 2 / HBM_PEAK_GBS, 4),
               "note": "a plain device copy of %d points (16 B/point), timed here; two passes at that rate = the most this "
                       "two-pass transform can show against the 16 B/point roofline" % grp}
    res = {"metric": "Mpoints/s Complex<f32> through the 2^20-point FFT node, batch 4096 (BASELINE config 4)",
           "value": round(float(batch_total) * FFT_N * args.steps / elapsed / 1e6, 1), "unit": "Mpoints/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "BASELINE config 4: FFTBatchNode(2^20, forward), %d transforms cut into %d per GPU, "
                                  "out of place, no communication" % (batch_total, nb),
                      "transforms_per_gpu": nb, "fft_size": FFT_N, "variant": args.variant, "backend": args.backend},
           "roofline": {"bound": "hbm", "kernel": "fft1024x16_kernel (two passes; the timer brackets both)",
                        "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                        "traffic": pmc_traffic("fft1024x16_kernel", n), "kernel_ms": round(kernel_ms, 5),
                        "launches_timed": int(kms.size), "timer_stride": 1, **in_stream_fields(sms),
                        "algorithmic_bytes_per_launch": FFT_BYTES_PER_POINT * n, "ceiling": ceiling}}
    res["world_size_seen"], res["ranks"] = ranks["world_size_seen"], ranks
    if transfer:
        res["transfer"] = transfer
    if world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline_c4()
    return res


def run_config1(ctx):
    """BASELINE config 1 on the GPU: the transmit chain PRBS7 -> BPSK -> PulseNode(rrc_taps(63, 4, 0.25), 4) -> MixerNode
    on 2^18 symbols = 2^20 samples per step, as ONE launch (pulse_poly_kernel<4,real,MIX>); the two-node form and the
    literal example (32 taps, no mixer, i16 wire format out) beside it.  Every rank runs its own symbol block."""
    import comms_rs_amd as c

    args, torch, rank, world = ctx.args, ctx.torch, ctx.rank, ctx.world
    nsym = 1 << ((args.n_log2 or 20) - 2)
    n = 4 * nsym
    # the symbol source is not on the hot path (restated on the host): PRBS7 bits, LFSR poly 0xC0 state 0x01
    bits = np.empty(nsym, np.uint8)
    st = 0x01
    for i in range(nsym):                                 # prns.rs:64-71: out = MSB, shift left, OR in parity(state & mask)
        fb = bin(st & 0xC0).count("1") & 1
        bits[i] = st >> 7
        st = ((st << 1) | fb) & 0xFF
    sym_h = (bits.astype(np.float32) * 2.0 - 1.0).astype(np.complex64)
    sym = torch.from_numpy(sym_h).to(ctx.dev)
    out = torch.empty(n, dtype=torch.complex64, device=ctx.dev)
    taps = c.rrc_taps(63, 4.0, 0.25)
    s = ctx.stream
    fused = c.PulseNode(taps, 4, device=ctx.local_rank).set_mixer(MIX_DPHASE)
    # self-check with product code only: one launch against the two reference nodes in series
    pn, mx = c.PulseNode(taps, 4, device=ctx.local_rank), c.MixerNode(MIX_DPHASE, device=ctx.local_rank)
    two = torch.empty_like(out)
    fused.run_dev(sym.data_ptr(), nsym, out.data_ptr(), s)
    pn.run_dev(sym.data_ptr(), nsym, two.data_ptr(), s)
    mx.run_dev(two.data_ptr(), n, two.data_ptr(), s)
    torch.cuda.synchronize()
    err, bound = float((out - two).abs().max()), 2e-5 * float(np.sum(np.abs(taps)))
    ctx.all_ok(err <= bound, "config 1: fused transmit chain vs PulseNode -> MixerNode differ by %g (bound %g)" % (err, bound))

    # `value`: the plain step loop (no timer attached: the pulse node's timer records two events around its launch,
    # a few us that a 10-us step would show); the kernel's duration comes from a second loop of the same launches
    elapsed = ctx.timed(lambda: fused.run_dev(sym.data_ptr(), nsym, out.data_ptr(), s), args.steps, args.warmup)
    # in-stream kernel time: the same loop again with the kernel stamping its own begin / end (a 5-us launch notices even
    # that: `stamped_ms_per_step` beside `ms_per_step`)
    stamper = c.KernelTimer(max(args.steps + args.warmup, 1), device=ctx.local_rank, stamps=True).attach(fused)
    stamped_elapsed = ctx.timed(lambda: fused.run_dev(sym.data_ptr(), nsym, out.data_ptr(), s), args.steps, args.warmup)
    sms = stamper.read_stamps_ms()[args.warmup:]
    sms = sms[sms > 0]
    stamper.close()
    timer = c.KernelTimer(max(args.steps, 1), device=ctx.local_rank).attach(fused)
    for _ in range(args.steps):
        fused.run_dev(sym.data_ptr(), nsym, out.data_ptr(), s)
    torch.cuda.synchronize()
    kms = timer.read_ms()
    timer.close()

    def two_nodes():
        pn.run_dev(sym.data_ptr(), nsym, two.data_ptr(), s)
        mx.run_dev(two.data_ptr(), n, two.data_ptr(), s)

    two_elapsed = ctx.timed(two_nodes, args.steps, max(args.warmup, 1))
    lit = c.PulseNode(c.rrc_taps(32, 4.0, 0.25), 4, device=ctx.local_rank).set_output_format("i16", 8192.0)
    out16 = torch.empty((n, 2), dtype=torch.int16, device=ctx.dev)
    lit_elapsed = ctx.timed(lambda: lit.run_dev(sym.data_ptr(), nsym, out16.data_ptr(), s), args.steps, max(args.warmup, 1))
    kernel_ms = float(np.mean(kms))
    ranks = rank_report(ctx, kernel_ms)
    if rank != 0:
        return None
    bytes_per_out = 8.0 / 4 + 8.0   # SURVEY 8d, pulse (sps = 4): 2 B read + 8 B written per output sample
    ach = bytes_per_out * n / (kernel_ms * 1e-3) / 1e9
    res = {"metric": "Msamples/s Complex<f32> out of PRBS->BPSK->63-tap RRC pulse shaping->mixer (BASELINE config 1)",
           "value": round(float(world) * n * args.steps / elapsed / 1e6, 1), "unit": "Msamples/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 5),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "BASELINE config 1 (examples/single_thread_bpsk.rs made deterministic): 2^%d PRBS7/BPSK "
                                  "symbols -> PulseNode(rrc_taps(63,4,0.25), 4) -> MixerNode(2pi*0.1) = 2^%d output samples "
                                  "per GPU and step, one launch" % (int(np.log2(nsym)), int(np.log2(n))),
                      "output_samples_per_gpu_per_step": n, "n_taps": 63, "sam_per_sym": 4, "kernel": "pulse_poly_kernel<4,real,MIX>"},
           "roofline": {"bound": "hbm", "kernel": "pulse_poly_kernel", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": pmc_traffic("pulse_poly_kernel", n),
                        "kernel_ms": round(kernel_ms, 5), "launches_timed": int(kms.size), "timer_stride": 1,
                        **in_stream_fields(sms), "stamped_ms_per_step": round(stamped_elapsed / args.steps * 1e3, 5),
                        "algorithmic_bytes_per_launch": bytes_per_out * n,
                        "note": "2^20 outputs are 10.5 MB: at this size the launch is latency-bound (the same kernel at "
                                "2^24 outputs: DESIGN.md section 4); the kernel's own begin / end timestamps, every launch of a second loop"},
           "two_nodes": {"value": round(float(world) * n * args.steps / two_elapsed / 1e6, 1), "unit": "Msamples/s",
                         "ms_per_step": round(two_elapsed / args.steps * 1e3, 5), "what": "PulseNode -> MixerNode, two launches"},
           "literal_example": {"value": round(float(world) * n * args.steps / lit_elapsed / 1e6, 1), "unit": "Msamples/s",
                               "ms_per_step": round(lit_elapsed / args.steps * 1e3, 5),
                               "what": "single_thread_bpsk.rs as written: 32-tap RRC x4, no mixer, x8192 -> i16 file "
                                       "samples written by the kernel's store stage, one launch"}}
    res["world_size_seen"], res["ranks"] = ranks["world_size_seen"], ranks
    if world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline_c1()
    return res


def free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n, argv):
    """`python bench.py --gpus N` with N > 1 and no RANK in the environment: spawn the N rank processes
    (this file again, one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as torchrun would), relay
    rank 0's JSON line and return the worst exit code.  Child processes only -- nothing is exec'ed -- and this
    parent imports no torch and makes no HIP call, before or after."""
    import signal
    import subprocess

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    env["MASTER_PORT"] = str(free_port())
    env["WORLD_SIZE"] = env["LOCAL_WORLD_SIZE"] = str(n)
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        # rank 0's stdout is the JSON line; the other ranks print nothing there
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc, left, stopped = 0, set(range(n)), set()
    try:
        while left:
            for r in sorted(left):
                code = procs[r].poll()
                if code is None:
                    continue
                left.discard(r)
                if code != 0 and r not in stopped:
                    rc = rc or (code if code > 0 else 1)
                    sys.stderr.write("bench.py: rank %d exited with %d; stopping the other ranks\n" % (r, code))
                    for q in left - stopped:  # the exact PIDs started above
                        procs[q].send_signal(signal.SIGTERM)
                        stopped.add(q)
            time.sleep(0.05)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
            pr.wait()
    return rc


def launch_check(ctx_args):
    """--launch-check: rendezvous only (gloo, CPU): every rank joins the group, the ranks sum their rank
    numbers, rank 0 prints what the group saw.  Runs without a GPU: the test of the launcher itself."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    import torch
    import torch.distributed as dist

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    assert world == ctx_args.gpus, "--gpus %d but WORLD_SIZE=%d" % (ctx_args.gpus, world)
    seen, total = 1, 0
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank)], dtype=torch.float64)
        dist.all_reduce(t)
        seen, total = dist.get_world_size(), int(t.item())
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "world_size_seen": seen, "rank_sum": total,
                          "local_rank": int(os.environ.get("LOCAL_RANK", "0"))}), flush=True)


def dry_comm(args):
    """--dry-comm: the communication of a multi-GPU run and nothing else -- the rendezvous of the backend (nccl = RCCL
    over xGMI unless --backend gloo), one halo hand-over per neighbour pair at each config's message size, the all_ok /
    max_over_ranks / over_ranks collectives of the timed loop, and out.  Seconds on a first 8-GPU lease instead of a
    bench run that finds a bring-up problem after minutes; every rank checks the halo it got against the stream."""
    import comms_rs_amd as c
    from comms_rs_amd.sharding import halo_exchange

    t_start = time.perf_counter()
    ctx = Ctx(args)
    t_init = time.perf_counter() - t_start
    torch, rank, world = ctx.torch, ctx.rank, ctx.world
    sizes = {"config2_fir_halo": N_TAPS, "config3_chain_prefix": 136, "config5_fir_halo": 4097}
    per = 1 << 16   # a small shard per rank: the halo is its last samples
    x = torch.empty(per, dtype=torch.complex64, device=ctx.dev)
    c.synth_iq_dev(x.data_ptr(), per, rank * per, SEED, device=ctx.local_rank, stream=ctx.stream)
    torch.cuda.synchronize()
    report, ok = {}, True
    for name, h in sizes.items():
        t0 = time.perf_counter()
        if world > 1:
            halo = halo_exchange(ctx.dist, ctx.comm_view(x[per - h:].clone()), rank, world)
            torch.cuda.synchronize()
            if rank > 0:
                got = torch.view_as_complex(halo.contiguous()).cpu().numpy()
                ok = ok and np.array_equal(got, c.synth_iq(h, rank * per - h, SEED))
        report[name] = {"samples": h, "bytes_per_neighbour_pair": 8 * h, "ms": round((time.perf_counter() - t0) * 1e3, 3)}
    ctx.all_ok(ok, "a halo differs from the stream's samples before the shard")
    worst = ctx.max_over_ranks(float(rank))
    seen = ctx.over_ranks(float(rank))
    ctx.barrier()
    if rank == 0:
        print(json.dumps({"dry_comm": True, "backend": args.backend if world > 1 else None, "n_gpus": world,
                          "world_size_seen": ctx.world_size_seen(), "init_s": round(t_init, 3), "halos": report,
                          "max_over_ranks": worst, "over_ranks": seen, "total_s": round(time.perf_counter() - t_start, 3)}), flush=True)
    if world > 1:
        ctx.dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)   # 0.12 s of timed region at config 2: long enough for an outside
    ap.add_argument("--settle-ms", type=float, default=40.0)  # untimed steps before the warm-up (clock transient; 0 = none)
    ap.add_argument("--warmup", type=int, default=50)    # utilisation sampler to see; the whole default run stays ~10 s
    ap.add_argument("--config", type=int, choices=[1, 2, 3, 4, 5], default=2)
    ap.add_argument("--variant", choices=["inplace", "scatter"], default="inplace")
    ap.add_argument("--stream-log2", type=int, default=30,
                    help="config 2: log2 of the whole stream of the strong-scaling extra (0 = skip it)")
    ap.add_argument("--n-log2", type=int, default=0, help="rehearsal: override log2 of the per-GPU size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-vec", action="store_true", help="skip the host-pointer (Vec in, Vec out) leg of the default config")
    ap.add_argument("--head-first", action="store_true",
                    help="config 2: time the headline chain before the 2^30-sample stream leg (default: after it)")
    ap.add_argument("--algo", choices=["auto", "direct", "os1024", "os4096"], default="auto")
    # rehearsal aid: "gloo" runs the N>1 logic with CPU-side messages, ranks sharing the visible GPUs
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl")
    ap.add_argument("--dry-comm", action="store_true",
                    help="rendezvous of the chosen backend, one halo hand-over per config's message size, the collectives of "
                         "the timed loop, and exit: a seconds-long first use of an N-GPU node")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous of the N ranks only (gloo on the CPU), no GPU work: checks the launcher")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.launch_check:
        return launch_check(args)
    if args.dry_comm:
        return dry_comm(args)
    if args.config == 1 and args.steps == 1000 and args.warmup == 50:
        args.steps, args.warmup = 200, 20
    elif args.config in (3, 5) and args.steps == 1000 and args.warmup == 50:
        # steps of 0.11 / 0.6 ms: a warm-up as long as the chip's start-up clock transient (about 25 ms: the launches of the
        # second to twentieth millisecond of a burst run up to 40 % slower, profiles/r03_config5_kernel_trace.txt,
        # r05_config3_poly8_launch_order.txt), short enough for a default run
        args.steps, args.warmup = 200, (200 if args.config == 3 else 40)
    elif args.config != 2 and args.steps == 1000 and args.warmup == 50:
        args.steps, args.warmup = 20, 3   # config 4's steps are 31 ms each

    ctx = Ctx(args)
    out = {1: run_config1, 2: run_config2, 3: run_config3, 4: run_config4, 5: run_config5}[args.config](ctx)
    if ctx.rank == 0:
        print(json.dumps(out), flush=True)
    if ctx.world > 1:
        ctx.dist.destroy_process_group()


if __name__ == "__main__":
    main()
