#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X.

Metric : Msamples/s of Complex<f32> through the 255-tap FIR -> mixer -> decimate
         chain (BASELINE.json `metric`), whole job over all ranks.
Step   : one pass of the chain over one resident batch of 2^24 synthetic IQ
         samples per GPU (BASELINE config 2), node by node through the C ABI
         (comms_fir_run_dev -> comms_mixer_run_dev -> comms_decimate_run_dev),
         device-resident in and out, FIR/mixer state carried across steps.
N > 1  : the long stream is cut into N contiguous shards, one rank per GPU; the
         only cross-rank step is the one-off 254-sample halo hand-over to the
         right-hand neighbour over RCCL (setup, untimed).  The data path has no
         collective -> weak scaling (per-GPU work fixed).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
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


def pmc_traffic(kernel):
    """HBM bytes per FIR launch from the committed rocprofv3 PMC passes (profiles/), or None."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
        if d.get("kernel") == kernel and d.get("n_samples") == N_SAMPLES:
            return d.get("hbm_bytes_per_launch")
    except Exception:
        pass
    return None


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

        threads = max(1, min(os.cpu_count() or 1, 64))
        chunk = n // threads

        def run_chunk(i, norotate):
            st = oracle.default_state(taps)
            seg = x[i * chunk:(i + 1) * chunk]
            oracle.decimate(oracle.Mixer(0.0, MIX_DPHASE).mix(oracle.batch_fir(seg, taps, st, norotate=norotate)), DEC_RATE)

        for key, norot in (("all_cores", False), ("all_cores_norotate", True)):
            with ThreadPoolExecutor(threads) as ex:
                list(ex.map(lambda i: run_chunk(i, norot), range(threads)))  # warm-up: threads, first-touch pages
                t0 = time.perf_counter()
                list(ex.map(lambda i: run_chunk(i, norot), range(threads)))
                dt = time.perf_counter() - t0
            out[key] = {"value": round(chunk * threads / dt / 1e6, 2), "unit": "Msamples/s", "cores": threads}
        t0 = time.perf_counter()
        run_chunk(0, True)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--algo", choices=["auto", "direct", "os1024", "os4096"], default="auto")
    # rehearsal aid: "gloo" runs the N>1 logic with CPU-side messages, ranks sharing the visible GPUs
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import comms_rs_amd as c
    from comms_rs_amd.sharding import halo_exchange, shard_mixer_phase, state_from_halo

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)
    ndev = torch.cuda.device_count()
    assert ndev >= 1, "no MI355X visible (no CPU fallback)"
    if args.backend == "gloo":
        local_rank = local_rank % ndev
    assert local_rank < ndev, "LOCAL_RANK %d but only %d GPU(s) visible" % (local_rank, ndev)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":  # RCCL over xGMI
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", local_rank)
    comm_dev = dev if args.backend == "nccl" else torch.device("cpu")

    n = N_SAMPLES
    taps = c.rrc_taps(N_TAPS, 8.0, 0.35)
    # rank r owns stream samples [r*n, (r+1)*n): generated in place on its GPU
    x = torch.empty(n, dtype=torch.complex64, device=dev)
    y = torch.empty(n, dtype=torch.complex64, device=dev)
    z = torch.empty(n // DEC_RATE, dtype=torch.complex64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    c.synth_iq_dev(x.data_ptr(), n, rank * n, SEED, device=local_rank, stream=stream)

    fir = c.BatchFirNode(taps, device=local_rank)
    if args.algo != "auto":
        fir.set_algo({"direct": c.FIR_DIRECT, "os1024": c.FIR_OS1024, "os4096": c.FIR_OS4096}[args.algo])
    # every rank continues the un-sharded oscillator: closed-form phase of its first sample
    mix_phase = shard_mixer_phase(0.0, MIX_DPHASE, rank * n)
    mixer = c.MixerNode(MIX_DPHASE, mix_phase, device=local_rank)
    dec = c.DecimateNode(DEC_RATE, device=local_rank)

    # ---- one-off halo hand-over to the right-hand neighbour (RCCL send/recv)
    if world > 1:

        tail = torch.view_as_real(x[n - N_TAPS:].clone()).to(comm_dev)
        halo = halo_exchange(dist, tail, rank, world)
        torch.cuda.synchronize()
        if rank > 0:
            h = torch.view_as_complex(halo).cpu().numpy()
            assert np.array_equal(h, c.synth_iq(N_TAPS, rank * n - N_TAPS, SEED)), "halo mismatch"
            fir.set_state(state_from_halo(h))

    def step():
        fir.run_dev(x.data_ptr(), n, y.data_ptr(), stream)
        mixer.run_dev(y.data_ptr(), n, y.data_ptr(), stream)
        dec.run_dev(y.data_ptr(), n, 8, z.data_ptr(), stream)

    # ---- self-check of the first step (fail loudly): the node-by-node chain (FFT overlap-save
    # FIR, then mixer, then decimate) against the fused chain node, an independent time-domain
    # kernel, on the same shard.  (Parity with the reference's arithmetic is the job of tests/
    # and smoke(); nothing on this path touches the CPU checker.)
    step()
    chk = c.ChainNode(MIX_DPHASE, mix_phase, taps, DEC_RATE, False, device=local_rank, mixer_after_fir=True)
    if world > 1 and rank > 0:
        chk.set_fir_state(state_from_halo(h))
    zc = torch.empty_like(z)
    chk.run_dev(x.data_ptr(), n, zc.data_ptr(), stream)
    torch.cuda.synchronize()
    err = float((z - zc).abs().max())
    assert err <= 2e-5 * float(np.sum(np.abs(taps))), "self-check failed: chain vs fused chain differ by %g" % err
    del chk, zc

    for _ in range(args.warmup):
        step()
    timer = c.KernelTimer(max(args.steps, 1), device=local_rank).attach(fir)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kms = timer.read_ms()
    timer.close()

    # ---- the same chain as ONE fused node (comms_chain_*: additional node, same results)
    chain = c.ChainNode(MIX_DPHASE, mix_phase, taps, DEC_RATE, False, device=local_rank, mixer_after_fir=True)
    zf = torch.empty_like(z)
    for _ in range(max(args.warmup, 1)):
        chain.run_dev(x.data_ptr(), n, zf.data_ptr(), stream)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    tf0 = time.perf_counter()
    for _ in range(args.steps):
        chain.run_dev(x.data_ptr(), n, zf.data_ptr(), stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    fused_elapsed = time.perf_counter() - tf0
    if world > 1:
        t = torch.tensor([fused_elapsed], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        fused_elapsed = float(t.item())
    kernel_ms = float(np.mean(kms)) if kms.size else float("nan")
    algo = fir.kernel_for(n)

    if rank == 0:
        total = float(world) * n * args.steps
        achieved = FIR_BYTES_PER_SAMPLE * n / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "Msamples/s Complex<f32> through 255-tap FIR->mix->decimate chain",
            "value": round(total / elapsed / 1e6, 1),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE config 2: 255-tap BatchFirNode (rrc_taps(255,8,0.35)) -> mixer "
                                   "(dphase 2pi*0.1) -> decimate-by-8 on 2^24 Complex<f32> IQ per GPU, "
                                   "node by node, device-resident",
                       "samples_per_gpu_per_step": n, "n_taps": N_TAPS, "dec_rate": DEC_RATE,
                       "fir_kernel": algo, "sharding": "contiguous stream shards, one-off RCCL halo"},
            "roofline": {"bound": "hbm", "kernel": algo, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": pmc_traffic(algo),
                         "kernel_ms": round(kernel_ms, 5), "launches_timed": int(kms.size),
                         "algorithmic_bytes_per_launch": FIR_BYTES_PER_SAMPLE * n},
        }
        out["fused_chain"] = {"value": round(total / fused_elapsed / 1e6, 1), "unit": "Msamples/s",
                              "ms_per_step": round(fused_elapsed / args.steps * 1e3, 4), "fused": chain.fused,
                              "kernel": {"time": "fir_decim_kernel", "freq": "fir_os1024_kernel<.., MODE>",
                                         "unfused": "four kernels"}[chain.kernel],
                              "note": "same FIR->mixer->decimate chain as one comms_chain_* launch "
                                      "(8 B read + 1 B written per input sample); not the headline value"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
