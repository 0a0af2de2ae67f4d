"""CPU-side tests (no GPU): the C ABI library builds, loads and exports every
symbol include/comms_hip.h declares; product entry points fail loudly without a
device; host tap design matches the reference goldens; the C++ graph runtime;
the stream-sharding logic under a 2-process gloo group."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def c():
    import __graft_entry__ as g

    g.build()
    import comms_rs_amd as c

    return c


def header_symbols():
    text = open(os.path.join(ROOT, "include", "comms_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(comms_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(c):
    import ctypes

    lib = ctypes.CDLL(c.LIB_PATH)
    names = header_symbols()
    assert len(names) >= 60
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # the ctypes prototype table covers the same set
    from comms_rs_amd import _lib

    assert sorted(_lib.all_symbols()) == names


def test_no_cpu_fallback_and_error_reporting(c):
    # in this container there is no GPU: every create must fail with COMMS_ERR_DEVICE (2),
    # never compute on the host.  On a GPU box this test is skipped.
    if c.device_count() > 0:
        pytest.skip("GPU present")
    for make in (lambda: c.BatchFirNode(np.ones(4, np.complex64)), lambda: c.MixerNode(0.1),
                 lambda: c.FMDemodNode(), lambda: c.FFTBatchNode(16, False), lambda: c.PulseNode(np.ones(4), 2),
                 lambda: c.ChainNode(0.1, 0.0, np.ones(4), 2, True), lambda: c.DeviceBuf(16),
                 lambda: c.DecimateNode(2).run(np.arange(6, dtype=np.int32))):
        with pytest.raises(c.CommsError) as e:
            make()
        assert e.value.code == 2
        assert "no CPU fallback" in str(e.value) or "HIP" in str(e.value)


def test_product_package_never_touches_the_oracle():
    # the oracle is test infrastructure: nothing under comms_rs_amd/ may import, link or open it
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "comms_rs_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"\boracle\b", txt):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_host_tap_design_matches_reference_goldens(c, kats):
    # src/util/math.rs:359-488 through comms_{rrc,rc,gaussian,rect}_taps (host code, no GPU needed)
    for key, fn in (("rrc_taps", c.rrc_taps), ("rc_taps", c.rc_taps), ("gaussian_taps", c.gaussian_taps)):
        k = kats[key]
        got = fn(k["n_taps"], k["sam_per_sym"], k.get("beta", k.get("alpha")))
        assert np.all(got.imag == 0)
        assert np.max(np.abs(got.real - np.array(k["expected_re"]))) < k["tol_abs"]
    assert np.array_equal(c.rect_taps(12), np.ones(12, np.complex64))
    for bad in (-0.1, 1.1):
        with pytest.raises(c.CommsError) as e:
            c.rrc_taps(8, 4.0, bad)
        assert e.value.code == 1  # MathError::InvalidRolloffError -> COMMS_ERR_ARG
    # product tap design == oracle tap design, bit for bit, incl. both special branches
    import oracle

    for n, sps, beta in [(63, 4.0, 0.25), (255, 8.0, 0.35), (32, 4.0, 0.25), (33, 3.18, 0.234), (9, 2.0, 0.0), (9, 2.0, 1.0)]:
        assert np.array_equal(c.rrc_taps(n, sps, beta), oracle.rrc_taps(n, sps, beta), equal_nan=True)
        # beta == 0 makes the reference's rc_taps centre tap sinc(1/0) = NaN (math.rs:161,183):
        # reproduced, not "fixed"
        assert np.array_equal(c.rc_taps(n, sps, beta), oracle.rc_taps(n, sps, beta), equal_nan=True)


def test_synthetic_generator_is_counter_based(c):
    a = c.synth_iq(1000, 0)
    b = np.concatenate([c.synth_iq(300, 0), c.synth_iq(700, 300)])
    assert np.array_equal(a, b)
    assert a.real.min() >= -1 and a.real.max() < 1 and abs(a.real.mean()) < 0.1
    assert not np.array_equal(a, c.synth_iq(1000, 0, seed=1))


def test_out_len_helpers(c):
    assert c.DecimateNode(3).out_len(8) == 3 and c.DecimateNode(0).out_len(8) == 8
    assert c.DecimateNode(100).out_len(6) == 1 and c.DecimateNode(2).out_len(0) == 0
    assert c.UpsampleNode(4).out_len(4) == 16 and c.UpsampleNode(1).out_len(4) == 4


def test_cpp_graph_runtime():
    # comms_rs_amd/host: Node / NodeReceiver / NodeSender / connect / start / Graph semantics
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "comms_rs_amd", "host"), "-s"], timeout=600)
    out = subprocess.run([os.path.join(ROOT, "comms_rs_amd", "lib", "test_graph")], capture_output=True,
                         text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "all passed" in out.stdout


@pytest.mark.parametrize("san", ["asan", "tsan"])
def test_cpp_graph_runtime_under_sanitizers(san):
    """AddressSanitizer + UBSan, and ThreadSanitizer, over the host graph runtime's own tests (channels,
    DeriveNode incl. drained blocks, feedback, thread pool, Graph).  CPU build only."""
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "comms_rs_amd", "host"), san], capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "all passed" in out.stdout


def test_u8_conversion_shortcut_matches_the_divide_for_all_bytes():
    """csrc/fir_handle.hpp InU8::cvt replaces (x - 127.5) / 127.5 (examples/fm_radio.rs:85-86) by
    q = a * fl(1/127.5); q + fma(-q, 127.5, a) * fl(1/127.5).  Exhaustive over the 256 byte values,
    with the fma steps evaluated exactly (f64 holds the 24 x 24 bit products) and rounded once."""
    x = np.arange(256, dtype=np.float32)
    want = (x - np.float32(127.5)) / np.float32(127.5)
    kinv = np.float32(1.0) / np.float32(127.5)
    a = x - np.float32(127.5)
    q = a * kinv
    r = (a.astype(np.float64) - q.astype(np.float64) * 127.5).astype(np.float32)
    got = (r.astype(np.float64) * np.float64(kinv) + q.astype(np.float64)).astype(np.float32)
    assert np.array_equal(got, want)
    import oracle

    assert np.array_equal(oracle.iq_u8_to_c32(np.stack([x, x], 1).astype(np.uint8)).real, want)


def test_shard_ranges():
    from comms_rs_amd.sharding import shard_range

    for total, world in [(1 << 30, 8), (1000, 3), (7, 8), (0, 2)]:
        edges = [shard_range(total, world, r) for r in range(world)]
        assert edges[0][0] == 0 and edges[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(edges[:-1], edges[1:]))
        sizes = [b - a for a, b in edges]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
import oracle
from comms_rs_amd import synth_iq
from comms_rs_amd.sharding import (chain_prefix_len, gather_shards, halo_exchange, scatter_shards, shard_range,
                                   state_from_halo, shard_mixer_phase)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
total, n_taps, rate = 51200, 255, 8                      # 51200 / world is a multiple of the rate for world = 2, 4, 8
taps = oracle.rrc_taps(n_taps, 8.0, 0.35)
a, b = shard_range(total, world, rank)
x = synth_iq(b - a, a)                                   # shard generated in place
tail = torch.view_as_real(torch.from_numpy(x[-n_taps:].copy()))
halo = halo_exchange(dist, tail, rank, world)            # the only communication on the path
state = oracle.default_state(taps)
if rank > 0:
    state = state_from_halo(torch.view_as_complex(halo).numpy())
y = oracle.batch_fir(x, taps, state, norotate=True)      # checker stands in for the GPU node here
full = oracle.batch_fir(synth_iq(total, 0), taps, oracle.default_state(taps), norotate=True)
# mixer after the FIR: the shard's oscillator starts at the closed-form phase of sample a
dphase = 2 * np.pi * 0.1
ym = oracle.Mixer(shard_mixer_phase(0.3, dphase, a), dphase).mix(y)
fullm = oracle.Mixer(0.3, dphase).mix(full)
same_mix = np.max(np.abs(ym - fullm[a:b])) <= 1e-6 * np.max(np.abs(fullm))
ok = np.array_equal(y, full[a:b]) and same_mix

# ---- secondary variant: scatter from the root, FM chain primed by a prefix, gather on the root
W = chain_prefix_len(n_taps, rate, True)
assert W % rate == 0 and W >= n_taps - 1 + rate and chain_prefix_len(n_taps, rate, False) == 256
whole = synth_iq(total, 0, 5)
shard = torch.empty(b - a, 2)
scatter_shards(dist, torch.view_as_real(torch.from_numpy(whole)) if rank == 0 else None, shard, rank, world)
xs = torch.view_as_complex(shard).numpy()
ok = ok and np.array_equal(xs, whole[a:b])
pre = halo_exchange(dist, torch.view_as_real(torch.from_numpy(xs[-W:].copy())), rank, world)
def chain(start):   # MixerNode -> BatchFirNode -> DecimateNode -> FMDemodNode, reference arithmetic
    mx, st, fm = oracle.Mixer(shard_mixer_phase(0.0, dphase, start), dphase), oracle.default_state(taps), oracle.FM()
    return lambda v: fm.demod(oracle.decimate(oracle.batch_fir(mx.mix(v), taps, st, norotate=True), rate))
node = chain(a - W if rank else a)
if rank:
    node(torch.view_as_complex(pre).numpy())             # prime: outputs thrown away
out = node(xs)
res = torch.empty(total // rate) if rank == 0 else None
gather_shards(dist, torch.from_numpy(out), res, rank, world)
if rank == 0:
    ref = chain(0)(whole)
    d = np.abs(res.numpy().astype(np.float64) - ref); d = np.minimum(d, 2 * np.pi - d)
    y_dec = oracle.decimate(oracle.batch_fir(oracle.Mixer(0.0, dphase).mix(whole), taps, oracle.default_state(taps), norotate=True), rate)
    mag = np.minimum(np.abs(y_dec), np.abs(np.concatenate([[0.0], y_dec[:-1]])))
    ok = ok and np.max(d * mag) <= 1e-5                  # the primed FIR restarts its rounding: not bit-equal
ok = torch.tensor([1.0 if ok else 0.0])
dist.all_reduce(ok, op=dist.ReduceOp.MIN)
dist.destroy_process_group()
sys.exit(0 if ok.item() == 1.0 else 3)
"""


def _run_gloo_world(tmp_path, world, port):
    import __graft_entry__ as g

    g.build()
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world),
               COMMS_NO_TORCH_PRELOAD="1", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r))) for r in range(world)]
    rcs = [p.wait(timeout=300) for p in procs]
    assert rcs == [0] * world, rcs


def test_sharded_stream_equals_unsharded_gloo_world2(tmp_path):
    """Two CPU ranks (gloo): contiguous shards + one halo hand-over reproduce the
    un-sharded filter output exactly; scatter -> prefix-primed FM chain -> gather reproduces the
    un-sharded chain -- the N>1 logic of bench.py and sharding.py, minus the GPU (the same logic
    with the product nodes: tests/test_gpu_sharding.py)."""
    _run_gloo_world(tmp_path, 2, 29517)


@pytest.mark.parametrize("world,port", [(4, 29518), (8, 29519)])
def test_sharded_stream_equals_unsharded_gloo_world4_and_8(tmp_path, world, port):
    """The same with four and eight ranks -- the driver's scaling run goes to N = 8: interior ranks both send
    and receive a halo, the root scatters to / gathers from seven peers."""
    _run_gloo_world(tmp_path, world, port)


def test_rust_shim_declares_every_header_symbol():
    """The Rust shim is untested source (no rustc here); at least its `extern "C"` block is
    generated from the header and must be in step with it."""
    import subprocess

    assert subprocess.call([sys.executable, os.path.join(ROOT, "scripts", "gen_rust_ffi.py"), "--check"]) == 0, \
        "comms_rs_amd/rust_shim/src/ffi.rs is stale: run scripts/gen_rust_ffi.py"


# Constructors of the reference nodes on the hot path: struct, type-parameter bound as the shim
# narrows it, arguments of `new`, and the reference's declaration (file:line in the reference checkout).
RUST_NODE_CONTRACT = [
    ("FirNode", "FirSample", ["taps: Vec<Complex<T>>", "state: Option<Vec<Complex<T>>>"], "src/filter/fir_node.rs:45,89"),
    ("BatchFirNode", "FirSample", ["taps: Vec<Complex<T>>", "state: Option<Vec<Complex<T>>>"], "src/filter/fir_node.rs:148,193"),
    ("FFTBatchNode", "FloatSample", ["fft_size: usize", "ifft: bool"], "src/fft/fft_node.rs:28,65"),
    ("FFTSampleNode", "FloatSample", ["fft_size: usize", "ifft: bool"], "src/fft/fft_node.rs:104,142"),
    ("MixerNode", "MixerSample", ["dphase: f64", "phase: Option<f64>"], "src/mixer.rs:93,128"),
    ("PulseNode", "FirSample", ["taps: Vec<Complex<T>>", "sam_per_sym: usize"], "src/pulse.rs:38,71"),
    ("DecimateNode", "Copy + Send", ["dec_rate: usize"], "src/util/resample_node.rs:10,23"),
    ("UpsampleNode", "Copy + Send + Zero", ["ups_rate: usize"], "src/util/resample_node.rs:74,87"),
    ("FMDemodNode", "FloatSample", [], "src/modulation/analog_node.rs:20,43"),
]


def test_rust_shim_keeps_the_reference_constructors():
    """Every reference node of the hot path has a same-named generic struct in the shim with public `input` /
    `output` fields and a `pub fn new` of the reference's arity and argument types, so that a graph switches by
    its `use` lines (examples/fm_radio.rs:146-153 declares `BatchFirNode<f32>`, `DecimateNode<Complex<f32>>` and
    `DecimateNode<f32>`).  Text check only: there is no rustc here."""
    import re

    src = open(os.path.join(ROOT, "comms_rs_amd", "rust_shim", "src", "lib.rs")).read()
    for name, bound, args, where in RUST_NODE_CONTRACT:
        m = re.search(r"pub struct %s<T>\s*where\s*T: ([^,{]+),\s*\{(.*?)\n\}" % name, src, flags=re.S)
        assert m, "%s<T> (%s) is not declared as a generic struct" % (name, where)
        assert m.group(1).strip() == bound, (name, m.group(1))
        body = m.group(2)
        assert re.search(r"pub input: NodeReceiver<", body) and re.search(r"pub output: NodeSender<", body), name
        im = re.search(r"impl<T> %s<T>\s*where\s*T: [^{]+\{(.*?)\n\}" % name, src, flags=re.S)
        assert im, "no `impl<T> %s<T>`" % name
        nm = re.search(r"pub fn new\(([^)]*)\) -> Self", im.group(1))
        assert nm, "%s::new is missing" % name
        got = [a.strip() for a in nm.group(1).split(",") if a.strip()]
        assert got == args, (name, got, args, where)
        assert re.search(r"pub fn run\(&mut self", im.group(1)), "%s::run is missing" % name
    # the aggregate node keeps its attribute (node_derive/src/lib.rs:139-151) and the per-sample message type
    assert re.search(r"#\[derive\(Node\)\]\s*#\[aggregate\]\s*#\[pass_by_ref\]\s*pub struct FFTSampleNode<T>", src)
    assert "-> Result<Option<Vec<Complex<T>>>, NodeError>" in src
    # the sealed sample traits are implemented for what the C ABI has
    for t, tys in (("FirSample", ("f32", "i16")), ("MixerSample", ("f32", "f64")), ("FloatSample", ("f32",))):
        for ty in tys:
            assert "impl %s for %s" % (t, ty) in src, (t, ty)
    # every extern function the shim calls is declared in the generated ffi.rs
    ffi = open(os.path.join(ROOT, "comms_rs_amd", "rust_shim", "src", "ffi.rs")).read()
    declared = set(re.findall(r"pub fn (comms_[a-z0-9_]+)\(", ffi))
    used = set(re.findall(r"\b(comms_[a-z0-9_]+)\(", src))
    assert used <= declared, sorted(used - declared)


def test_library_exports_exactly_the_header():
    """The product .so exports every comms_* symbol of include/comms_hip.h and nothing else: probes,
    stamps and kernel selectors live in the diagnostic build only (`make -C comms_rs_amd/csrc diag`)."""
    import re
    import subprocess

    import comms_rs_amd as c

    hdr = open(os.path.join(ROOT, "include", "comms_hip.h")).read()
    declared = set(re.findall(r"\b(comms_[a-z0-9_]+)\s*\(", hdr))
    out = subprocess.check_output(["nm", "-D", "--defined-only", c.LIB_PATH], text=True)
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.split()[-1].startswith("comms_")}
    assert exported == declared, (sorted(exported - declared), sorted(declared - exported))


def test_header_is_plain_c():
    """include/comms_hip.h is the drop-in boundary: it must compile as C99 (and C++17) on its own."""
    import subprocess

    hdr = os.path.join(ROOT, "include", "comms_hip.h")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr])
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c++", hdr])


def _bench(*argv, timeout=300):
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True,
                          text=True, timeout=timeout)


@pytest.mark.parametrize("world", [2, 4])
def test_bench_launches_its_own_ranks(world):
    """`python bench.py --gpus N` with no RANK in the environment spawns the N rank processes itself (children,
    no exec) and relays rank 0's line: here the rendezvous-only mode, which needs no GPU.  The group must have
    seen N ranks (the round-2 command died with `--gpus 2 but WORLD_SIZE=1`)."""
    import json

    r = _bench("--gpus", str(world), "--launch-check")
    assert r.returncode == 0, r.stderr
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1, r.stdout
    d = json.loads(line[0])
    assert d["n_gpus"] == world and d["world_size_seen"] == world and d["rank_sum"] == world * (world - 1) // 2


def test_bench_rank_failure_is_the_exit_code():
    """Without a GPU every rank of the real benchmark fails loudly (no CPU fallback); the launching parent
    must report that -- not the old WORLD_SIZE assertion -- and exit non-zero without leaving ranks behind."""
    r = _bench("--gpus", "2", "--backend", "gloo", "--n-log2", "20", "--stream-log2", "0")
    import comms_rs_amd as c

    if c.device_count() >= 1:
        pytest.skip("a GPU is present: this is the CPU-box behaviour")
    assert r.returncode != 0
    assert "WORLD_SIZE=1" not in r.stderr
    assert "no MI355X visible" in r.stderr or "COMMS_ERR_DEVICE" in r.stderr, r.stderr[-2000:]


def test_buf_alloc_rejects_absurd_sizes_before_touching_the_pool():
    """sizes beyond 2^40 bytes (e.g. an underflowed size_t) are COMMS_ERR_ARG -- checked before any
    device call, so this holds on the CPU box too (it used to walk past the size-class table / never return)."""
    import comms_rs_amd as c

    for bad in ((1 << 40) + 1, (1 << 48), (1 << 64) - 1):
        with pytest.raises(c.CommsError) as e:
            c.DeviceBuf(bad)
        assert e.value.code == c.COMMS_ERR_ARG


def test_shard_helpers_of_the_c_abi_match_their_definition():
    """comms_shard_range / comms_state_from_halo / comms_shard_mixer_phase / comms_chain_prefix_len (host arithmetic in
    the C ABI, bound by sharding.py and host/comms/nodes.hpp) against their definitions restated here."""
    from comms_rs_amd import sharding as sh

    for total, world in ((10, 4), (1 << 30, 8), (7, 8), (0, 3), (4096, 1)):
        cuts = [sh.shard_range(total, world, r) for r in range(world)]
        base, rem = divmod(total, world)
        assert cuts[0][0] == 0 and cuts[-1][1] == total
        for r, (a, b) in enumerate(cuts):
            assert b - a == base + (1 if r < rem else 0) and (r == 0 or a == cuts[r - 1][1])
    with pytest.raises(ValueError):
        sh.shard_range(10, 2, 2)
    h = (np.arange(7) + 1j * np.arange(7)[::-1]).astype(np.complex64)
    assert np.array_equal(sh.state_from_halo(h), h[::-1])
    two_pi = np.longdouble(2.0) * np.longdouble(np.pi)
    for p0, dp, i in ((0.0, 2 * np.pi * 0.1, 1 << 27), (0.3, 0.123, -136), (1.0, 5.9, 7 * (1 << 27)), (0.0, 0.0, 5)):
        want = np.fmod(np.longdouble(p0) + np.longdouble(i) * np.longdouble(dp), two_pi)
        want = float(want + two_pi if want < 0 else want)
        got = sh.shard_mixer_phase(p0, dp, i)
        assert 0.0 <= got < 2 * np.pi + 1e-12 and abs(got - want) < 1e-12
    for taps, rate, fm in ((127, 8, True), (255, 8, False), (63, 5, True), (1, 8, False), (4097, 1, False), (2, 0, True)):
        r = max(rate, 1)
        need = (taps - 1) + (r if fm else 0)
        assert sh.chain_prefix_len(taps, rate, fm) == -(-need // r) * r


def test_no_environment_switch_reaches_a_product_launch_path():
    """Kernel selectors and sweep knobs exist in the diagnostic build only (csrc/common.hpp: diag_knob is a getenv under
    COMMS_DIAG and a compile-time default otherwise).  Every other getenv in csrc/ must be on this allow-list of documented
    runtime LIMITS (resources, not kernel choices or results): a new one fails this test until it is listed here and in
    DESIGN.md section 1."""
    import re

    allowed = {("runtime.hip", "COMMS_ZERO_COPY_BYTES"), ("runtime.hip", "COMMS_BUF_POOL_MB"),
               ("runtime.hip", "COMMS_HOST_PIPE_BYTES"), ("runtime.hip", "COMMS_HOST_CHUNK_BYTES")}
    csrc = os.path.join(ROOT, "comms_rs_amd", "csrc")
    found = set()
    for name in sorted(os.listdir(csrc)):
        if not name.endswith((".hip", ".hpp", ".cpp", ".h")):
            continue
        text = open(os.path.join(csrc, name)).read()
        # strip what only the diagnostic build compiles
        depth, keep, stack = 0, [], []
        for line in text.split("\n"):
            st = line.strip()
            if st.startswith("#if"):
                stack.append(st.startswith("#ifdef COMMS_DIAG") or st.startswith("#if defined(COMMS_DIAG)"))
            elif st.startswith("#else") and stack:
                stack[-1] = False if stack[-1] else stack[-1]
            elif st.startswith("#endif") and stack:
                stack.pop()
            if not any(stack):
                keep.append(line)
        for m in re.finditer(r'getenv\(\s*("([A-Z0-9_]+)"|[a-z_]+)\s*\)', "\n".join(l for l in keep if not l.strip().startswith("//"))):
            found.add((name, m.group(2) or m.group(1)))
    assert found <= allowed, "getenv outside COMMS_DIAG: %s" % sorted(found - allowed)
    assert allowed <= found, "allow-list entries that no longer exist: %s" % sorted(allowed - found)
    # and the results-changing debug switch of round 4 is gone from every build's launch path
    assert 'getenv("COMMS_DECIM_DEBUG_NOMAC")' not in open(os.path.join(csrc, "fir_decim.hip")).read()


def test_polyphase_chain_kernel_lane_model():
    """scripts/proto_poly8.py: the lane-accurate numpy model of fir_poly8_kernel (index maps, exchange layouts, lane swaps, both
    output phases) against the direct sum y[8j] = sum_k h[k] x[8j - k] -- the description the kernel was written from."""
    import subprocess
    import sys

    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "scripts", "proto_poly8.py")], text=True, timeout=300)
    assert out.strip().endswith("OK"), out
