"""GPU tests of the multi-GPU path's building blocks, driven through the PRODUCT nodes (C ABI):
stream shards that continue their left neighbour (FIR halo, closed-form mixer phase, FM.prev via
a primed prefix), the checkpoint hooks (FIR state, oscillator phase, FM.prev getters / setters),
the one-stream-at-a-time rule of stateful handles, and a 2-rank gloo process group whose ranks
share the one GPU of the box (the N>1 logic of bench.py with the real kernels).  Run with -m gpu.

Reference behaviour being preserved: state persists across `run` calls -- src/filter/fir_node.rs:193-220,
src/mixer.rs:79-82, src/modulation/analog.rs:9,31,43-47; SURVEY.md section 8e for the partitioning.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
from test_gpu_parity import circ, fir_close, fm_stream, lowpass_taps, rand_c

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def c():
    import comms_rs_amd as c

    assert c.device_count() >= 1, "no MI355X visible: the HIP path cannot be tested (no CPU fallback)"
    return c


def oracle_chain(x, taps, dphase, phase0, rate, fm, after):
    """The reference nodes in series on the whole stream (the un-sharded truth)."""
    mx, st = oracle.Mixer(phase0, dphase), oracle.default_state(taps)
    if after:
        y = mx.mix(oracle.batch_fir(x, taps, st, norotate=True))
    else:
        y = oracle.batch_fir(mx.mix(x), taps, st, norotate=True)
    y = oracle.decimate(y, rate)
    return (oracle.FM().demod(y), y) if fm else (y, y)


def fm_close(got, want, y_dec, taps, lo=0):
    """Angles on the circle; where |y| is small the angle is ill-conditioned (error = FIR error / |y|)."""
    mag = np.minimum(np.abs(y_dec), np.abs(np.concatenate([[0.0], y_dec[:-1]])))
    d = circ(got.astype(np.float64) - want) * mag
    assert np.max(d[lo:], initial=0.0) <= 4e-5 * np.sum(np.abs(taps)), np.max(d[lo:])


# ------------------------------------------------------------------ shards through the product nodes
@pytest.mark.parametrize("rate", [8, 20, 100])   # per-rate kernel; any-rate kernel, LDS-staged and direct (variant "time")
@pytest.mark.parametrize("fm", [False, True])
@pytest.mark.parametrize("after", [False, True])
@pytest.mark.parametrize("variant", ["time", "freq", "unfused"])
def test_chain_shards_primed_by_prefix_continue_the_stream(c, variant, after, fm, rate):
    """Four contiguous shards of one stream, each on its own chain node: rank r > 0 primes its node
    with the chain_prefix_len raw samples before its shard (sharding.prime_chain: FIR history,
    oscillator phase AND FM.prev in one go).  The concatenated outputs equal the reference nodes in
    series on the whole stream."""
    import torch
    from comms_rs_amd.sharding import chain_prefix_len, prime_chain, shard_mixer_phase, shard_range

    n_taps, world = 127, 4
    total = rate * (4096 if rate == 8 else 1024) * world
    x = fm_stream(total) if fm else rand_c(np.random.default_rng(11), total)
    taps = lowpass_taps(n_taps, 1 / (2 * rate))
    dphase, phase0 = 2 * np.pi * 0.05, 0.3
    want, y_dec = oracle_chain(x, taps, dphase, phase0, rate, fm, after)
    W = chain_prefix_len(n_taps, rate, fm)
    assert W % rate == 0 and W >= n_taps - 1 + (rate if fm else 0)
    kw = dict(mixer_after_fir=after, unfused=variant == "unfused", kernel="auto" if variant == "unfused" else variant)
    xd = torch.from_numpy(x).cuda()
    s = torch.cuda.current_stream().cuda_stream
    outs = []
    for r in range(world):
        a, b = shard_range(total, world, r)
        start = a - W if r else a
        node = c.ChainNode(dphase, shard_mixer_phase(phase0, dphase, start), taps, rate, fm, **kw)
        o = torch.empty((b - a) // rate, dtype=torch.float32 if fm else torch.complex64, device="cuda:0")
        if r:
            scratch = torch.empty(W // rate, dtype=o.dtype, device="cuda:0")
            prime_chain(node, xd.data_ptr() + 8 * start, W, scratch.data_ptr(), s)
        node.run_dev(xd.data_ptr() + 8 * a, b - a, o.data_ptr(), s)
        torch.cuda.synchronize()
        outs.append(o.cpu().numpy())
    got = np.concatenate(outs)
    if fm:
        fm_close(got, want, y_dec, taps, lo=32)
    else:
        fir_close(got, want, taps, x)


def test_node_by_node_shards_with_fm_prev_setter(c):
    """The four reference nodes as four product nodes per shard (no fusion): FIR state from the halo
    (set_state), closed-form mixer phase, and FM.prev handed over explicitly with
    comms_fmdemod_set_prev -- here taken from the left shard's demodulator input, as a checkpointing
    host would.  Equal to the un-sharded product nodes and to the oracle within tolerance."""
    from comms_rs_amd.sharding import shard_mixer_phase, shard_range, state_from_halo

    rate, n_taps, world = 8, 127, 3
    total = rate * 3000 * world
    x = fm_stream(total)
    taps = lowpass_taps(n_taps, 1 / 16)
    dphase = 2 * np.pi * 0.05
    want, y_dec = oracle_chain(x, taps, dphase, 0.0, rate, True, False)
    whole = c.FMDemodNode().run(c.DecimateNode(rate).run(c.BatchFirNode(taps).run(c.MixerNode(dphase).run(x))))
    outs, prev = [], None
    for r in range(world):
        a, b = shard_range(total, world, r)
        mixer = c.MixerNode(dphase, shard_mixer_phase(0.0, dphase, a))
        fir, fmn = c.BatchFirNode(taps), c.FMDemodNode()
        if r:
            # the FIR node sits behind the mixer: its history holds MIXED samples
            halo_mixed = c.MixerNode(dphase, shard_mixer_phase(0.0, dphase, a - n_taps)).run(x[a - n_taps:a])
            fir.set_state(state_from_halo(halo_mixed))
            fmn.prev = prev
            assert fmn.prev == prev
        y = c.DecimateNode(rate).run(fir.run(mixer.run(x[a:b])))
        outs.append(fmn.run(y))
        prev = fmn.prev
        assert prev == y[-1]          # FM.prev after a batch = its last input sample (analog.rs:31)
    got = np.concatenate(outs)
    # (not bit for bit: a shard's mixer evaluates the same closed-form phase from another lane's rotor)
    fm_close(got, whole, y_dec, taps)
    assert np.mean(got == whole) > 0.99
    fm_close(got, want, y_dec, taps, lo=32)


# ------------------------------------------------------------------ checkpoint hooks
@pytest.mark.parametrize("rate", [8, 20, 100])
@pytest.mark.parametrize("fm", [False, True])
@pytest.mark.parametrize("variant", ["time", "freq", "unfused"])
def test_chain_checkpoint_and_restore(c, variant, fm, rate):
    """Run half a stream, read the chain's whole cross-call state (FIR history, oscillator phase,
    FM.prev), put it into a brand-new chain and run the second half: same as one chain run in two calls."""
    n_taps = 127
    n, cut = rate * 2500, rate * 1100
    x = fm_stream(n)
    taps = lowpass_taps(n_taps, 1 / (2 * rate))
    dphase, phase0 = 2 * np.pi * 0.05, 1.25
    kw = dict(unfused=variant == "unfused", kernel="auto" if variant == "unfused" else variant)
    one = c.ChainNode(dphase, phase0, taps, rate, fm, **kw)
    ref = np.concatenate([one.run(x[:cut]), one.run(x[cut:])])
    first = c.ChainNode(dphase, phase0, taps, rate, fm, **kw)
    out_a = first.run(x[:cut])
    state, phase = first.fir_state(n_taps), first.phase
    assert np.array_equal(state, x[:cut][::-1][:n_taps])     # raw input samples, newest first
    assert abs(phase - oracle_phase(phase0, dphase, cut)) < 1e-9
    second = c.ChainNode(dphase, 0.0, taps, rate, fm, **kw)
    second.set_fir_state(state)
    second.phase = phase
    if fm:
        second.fm_prev = first.fm_prev
        assert second.fm_prev == first.fm_prev
    else:
        with pytest.raises(c.CommsError):
            first.fm_prev
    got = np.concatenate([out_a, second.run(x[cut:])])
    if fm:
        assert np.max(circ(got.astype(np.float64) - ref)) <= 1e-6
    else:
        fir_close(got, ref, taps, x)
        assert np.mean(got == ref) > 0.99          # the phase read-back drops 11 of 64 fractional bits, no more


def oracle_phase(phase0, dphase, n):
    mx = oracle.Mixer(phase0, dphase)
    mx.mix(np.zeros(n, np.complex64))
    return mx.phase.value


def test_fmdemod_prev_roundtrip_and_mixer_set_phase(c):
    rng = np.random.default_rng(3)
    x = rand_c(rng, 5000)
    a = c.FMDemodNode()
    assert a.prev == 0
    whole = c.FMDemodNode().run(x)
    first = a.run(x[:1234])
    assert a.prev == x[1233]
    b = c.FMDemodNode()
    b.prev = a.prev
    assert np.array_equal(np.concatenate([first, b.run(x[1234:])]), whole)
    m1 = c.MixerNode(0.123, 0.5)
    ref = m1.run(x)
    m2 = c.MixerNode(0.123)
    m2.phase = 0.5
    assert np.array_equal(m2.run(x), ref)
    m2.phase = 0.5 + 4 * 2 * np.pi                # any finite angle, reduced like Mixer::new + wrap
    assert np.max(np.abs(m2.run(x) - ref)) <= 1e-6
    with pytest.raises(c.CommsError):
        m2.phase = float("nan")


# ------------------------------------------------------------------ streams
@pytest.mark.parametrize("algo", [1, 3])
def test_fir_state_is_safe_across_streams(c, algo):
    """the C ABI takes any stream.  Launch on a non-blocking side stream and read the state
    back at once; then alternate two non-blocking streams call by call.  The handle follows one stream
    at a time (it drains the old stream when the caller switches), so the results equal the oracle's."""
    import torch

    rng = np.random.default_rng(21)
    n_taps, n = 255, 1 << 20
    taps = rand_c(rng, n_taps)
    x = rand_c(rng, 4 * n)
    xd = torch.from_numpy(x).cuda()
    yd = torch.empty_like(xd)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()   # hipStreamNonBlocking
    node = c.BatchFirNode(taps).set_algo(algo)
    node.run_dev(xd.data_ptr(), n, yd.data_ptr(), s1.cuda_stream)
    st = node.state(n_taps)                               # no explicit synchronisation by the caller
    assert np.array_equal(st, x[:n][::-1][:n_taps])
    for i, s in enumerate((s2, s1, s2), start=1):
        node.run_dev(xd.data_ptr() + 8 * i * n, n, yd.data_ptr() + 8 * i * n, s.cuda_stream)
    st = node.state(n_taps)
    assert np.array_equal(st, x[::-1][:n_taps])
    torch.cuda.synchronize()
    got = yd.cpu().numpy()
    for a in (0, n - 1500, 2 * n - 1500, 3 * n - 1500, 4 * n - 3000):
        lo = max(0, a - (n_taps - 1))
        want = oracle.batch_fir(x[lo:a + 3000], taps, oracle.default_state(taps), norotate=True)[a - lo:]
        fir_close(got[a:a + 3000], want, taps, x)


def test_in_place_misuse_is_rejected(c):
    """outputs smaller than inputs invite in-place use; every such entry refuses overlap."""
    import torch

    x = torch.zeros(1 << 16, dtype=torch.complex64, device="cuda:0")
    taps = lowpass_taps(63, 1 / 16)
    for kw in (dict(kernel="time"), dict(kernel="freq"), dict(unfused=True)):
        with pytest.raises(c.CommsError):
            c.ChainNode(0.1, 0.0, taps, 8, False, **kw).run_dev(x.data_ptr(), x.numel(), x.data_ptr(), 0)
    with pytest.raises(c.CommsError):
        c.DecimateNode(8).run_dev(x.data_ptr(), x.numel(), 8, x.data_ptr(), 0)
    with pytest.raises(c.CommsError):                     # partial overlap is not "in place"
        c.FFTBatchNode(1024, False).run_dev(x.data_ptr(), 4096, x.data_ptr() + 8 * 512, 0)
    c.FFTBatchNode(1024, False).run_dev(x.data_ptr(), 4096, x.data_ptr(), 0)
    c.DecimateNode(1).run_dev(x.data_ptr(), x.numel(), 8, x.data_ptr(), 0)   # rate 1 = copy: same buffer is fine
    torch.cuda.synchronize()


# ------------------------------------------------------------------ two ranks, one GPU, gloo
_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
import comms_rs_amd as c
from comms_rs_amd.sharding import (chain_prefix_len, gather_shards, halo_exchange, prime_chain, scatter_shards,
                                   shard_mixer_phase, shard_range, state_from_halo)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
rate, n_taps, total = 8, 127, 8 * 40000
k = np.arange(n_taps) - (n_taps - 1) / 2
taps = (2 / 16 * np.sinc(2 / 16 * k) * np.hamming(n_taps)).astype(np.float32).astype(np.complex64)
dphase = 2 * np.pi * 0.05
a, b = shard_range(total, world, rank)
n = b - a
W = chain_prefix_len(n_taps, rate, True)
s = torch.cuda.current_stream().cuda_stream

# ---- scatter: the root holds the stream, every rank receives its shard (CPU tensors under gloo)
full = None
if rank == 0:
    idx = np.arange(total, dtype=np.float64)
    stream = np.exp(1j * (-2 * np.pi * 0.05 * idx + 8.0 * np.cos(2 * np.pi * idx / 4096))).astype(np.complex64)
    full = torch.view_as_real(torch.from_numpy(stream))
shard = torch.empty(n, 2)
scatter_shards(dist, full, shard, rank, world)
x = torch.view_as_complex(shard).cuda()

# ---- hand-over: the last W raw samples to the right-hand neighbour
halo = halo_exchange(dist, torch.view_as_real(x[n - W:].cpu()).contiguous(), rank, world)

# ---- product nodes: fused FM chain primed by the prefix; FIR node with the halo as state
start = a - W if rank else a
chain = c.ChainNode(dphase, shard_mixer_phase(0.0, dphase, start), taps, rate, True)
fm_out = torch.empty(n // rate, dtype=torch.float32, device="cuda:0")
fir = c.BatchFirNode(taps)
if rank:
    pre = torch.view_as_complex(halo.contiguous()).cuda()
    scratch = torch.empty(W // rate, dtype=torch.float32, device="cuda:0")
    prime_chain(chain, pre.data_ptr(), W, scratch.data_ptr(), s)
    fir.set_state(state_from_halo(pre[W - n_taps:].cpu().numpy()))
chain.run_dev(x.data_ptr(), n, fm_out.data_ptr(), s)
y = torch.empty_like(x)
fir.run_dev(x.data_ptr(), n, y.data_ptr(), s)
torch.cuda.synchronize()

# ---- gather both outputs on the root and compare with ONE product node over the whole stream
fm_full = torch.empty(total // rate) if rank == 0 else None
gather_shards(dist, fm_out.cpu(), fm_full, rank, world)
y_full = torch.empty(total, 2) if rank == 0 else None
gather_shards(dist, torch.view_as_real(y.cpu()).contiguous(), y_full, rank, world)
ok = 1.0
if rank == 0:
    whole_fm = c.ChainNode(dphase, 0.0, taps, rate, True).run(stream)
    whole_y = c.BatchFirNode(taps).run(stream)
    d = np.abs(fm_full.numpy().astype(np.float64) - whole_fm)
    d = np.minimum(d, 2 * np.pi - d)
    e = np.max(np.abs(torch.view_as_complex(y_full).numpy() - whole_y))
    print("fm max|d| %.3e (settled), fir max|d| %.3e" % (d[32:].max(), e), flush=True)
    if not (d[32:].max() <= 2e-5 and e <= 1e-5 * np.sum(np.abs(taps))):
        ok = 0.0
t = torch.tensor([ok])
dist.all_reduce(t, op=dist.ReduceOp.MIN)
dist.destroy_process_group()
sys.exit(0 if t.item() == 1.0 else 3)
"""


def test_two_ranks_share_the_gpu_gloo_product_nodes(tmp_path):
    """world_size 2 over gloo, both ranks on the box's one GPU: scatter -> halo hand-over -> product
    nodes (fused FM chain primed by the prefix, FIR with the halo as state) -> gather, against one
    product node over the whole stream.  This is bench.py's N>1 logic with the real kernels; RCCL
    itself needs one GPU per rank and runs in the driver's multi-GPU bench."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", WORLD_SIZE="2",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r))) for r in range(2)]
    rcs = [p.wait(timeout=600) for p in procs]
    assert rcs == [0, 0], rcs


def test_bench_dry_comm_two_ranks():
    """`bench.py --gpus 2 --dry-comm`: the rendezvous, one halo hand-over at every config's message size, the collectives
    of the timed loop -- here with gloo and both ranks on the box's one GPU (the nccl form is the driver's first
    multi-GPU step; with one GPU only its one-rank form can run: second call)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry-comm"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["dry_comm"] and line["world_size_seen"] == 2 and line["over_ranks"] == [0.0, 1.0] and line["max_over_ranks"] == 1.0
    assert line["halos"]["config2_fir_halo"]["bytes_per_neighbour_pair"] == 255 * 8
    assert line["halos"]["config3_chain_prefix"]["bytes_per_neighbour_pair"] == 136 * 8
    assert line["halos"]["config5_fir_halo"]["bytes_per_neighbour_pair"] == 4097 * 8
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dry-comm"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert one.returncode == 0 and json.loads(one.stdout.strip().splitlines()[-1])["world_size_seen"] == 1
