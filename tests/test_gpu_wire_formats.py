"""Raw-IQ input read by the first kernel (src/io/raw_iq.rs:16,50-51,
examples/fm_radio.rs:82-90, examples/single_thread_bpsk.rs:43): a node told that its input is i16 / u8
converts in its load stage.  The conversion must be iqformat.hip's (= the oracle's) bit for bit, so
"node on raw input" == "node on converted input" exactly, and == the oracle chain within the node's own
tolerance.  Run with -m gpu."""
import os

import numpy as np
import pytest

import oracle
from test_gpu_parity import circ, fir_close, lowpass_taps, rand_c

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c():
    import comms_rs_amd as c

    assert c.device_count() >= 1, "no MI355X visible: the HIP path cannot be tested (no CPU fallback)"
    return c


def raw_stream(rng, n, fmt):
    """An FM-like stream quantised to the wire format, plus every byte value at least once."""
    idx = np.arange(n, dtype=np.float64)
    z = np.exp(1j * (-2 * np.pi * 0.05 * idx + 8.0 * np.cos(2 * np.pi * idx / 4096)))
    if fmt == "u8":
        raw = np.stack([np.clip(np.rint(z.real * 100 + 127.5), 0, 255), np.clip(np.rint(z.imag * 100 + 127.5), 0, 255)], 1).astype(np.uint8)
        k = min(n, 256)
        raw[:k, 0] = np.arange(k)
        raw[:k, 1] = np.arange(k)[::-1]
        return raw, oracle.iq_u8_to_c32(raw)
    raw = np.stack([np.rint(z.real * 8192), np.rint(z.imag * 8192)], 1).astype(np.int16)
    raw[:4] = [[-32768, 32767], [0, -1], [1, 0], [12345, -12345]]
    return raw, oracle.iq_i16_to_c32(raw, 1.0 / 8192)


def test_u8_conversion_formula_is_exact_for_every_byte(c):
    """InU8::cvt (multiply by 1/127.5 + one fma residual step) against the correctly rounded divide of
    iqformat.hip, the oracle and numpy, for all 256 values, through a kernel that uses it (the direct FIR
    with the single tap 1)."""
    raw = np.stack([np.arange(256), np.arange(256)[::-1]], 1).astype(np.uint8)
    want = ((raw.astype(np.float32) - np.float32(127.5)) / np.float32(127.5))
    assert np.array_equal(oracle.iq_u8_to_c32(raw).view(np.float32).reshape(-1, 2), want)
    assert np.array_equal(c.iq_u8_to_c32(raw).view(np.float32).reshape(-1, 2), want)
    got = c.BatchFirNode(np.array([1 + 0j], np.complex64)).set_algo(c.FIR_DIRECT).set_input_format("u8").run(raw)
    assert np.array_equal(got.view(np.float32).reshape(-1, 2), want)


@pytest.mark.parametrize("fmt", ["i16", "u8"])
@pytest.mark.parametrize("algo,n", [("direct", 5000), ("os1024", 30000), ("dyn", 4200 * 768 + 100), ("os4096", 20000),
                                    ("os16k", 100000)])
def test_fir_node_reads_raw_iq(c, fmt, algo, n):
    """BatchFirNode on raw samples == BatchFirNode on the converted samples, bit for bit (same kernel, same
    arithmetic after the load stage), across two calls (the history is kept converted)."""
    rng = np.random.default_rng(5)
    n_taps = {"os4096": 600, "os16k": 3000}.get(algo, 255)
    taps = lowpass_taps(n_taps, 1 / 16)
    raw, x = raw_stream(rng, n, fmt)
    al = {"direct": c.FIR_DIRECT, "os1024": c.FIR_OS1024_FIXED, "dyn": c.FIR_OS1024, "os4096": c.FIR_OS4096, "os16k": c.FIR_OS16K}[algo]
    scale = 1.0 / 8192
    a = c.BatchFirNode(taps).set_algo(al).set_input_format(fmt, scale)
    b = c.BatchFirNode(taps).set_algo(al)
    if algo == "dyn":
        assert a.kernel_for(n) == "fir_os1024_dyn_kernel"
    cut = n // 2 if algo != "dyn" else n   # (the ticketed kernel needs >= 4096 segments per call)
    got = np.concatenate([a.run(raw[:cut]), a.run(raw[cut:])]) if cut < n else a.run(raw)
    ref = np.concatenate([b.run(x[:cut]), b.run(x[cut:])]) if cut < n else b.run(x)
    assert np.array_equal(got.view(np.float32), ref.view(np.float32))
    assert np.array_equal(a.state(n_taps), x[::-1][:n_taps])
    want = oracle.batch_fir(x[:4000], taps, oracle.default_state(taps), norotate=True)
    fir_close(got[:4000], want, taps, x)


@pytest.mark.parametrize("fmt", ["i16", "u8"])
@pytest.mark.parametrize("variant", ["time", "freq", "unfused"])
def test_fm_radio_chain_from_raw_iq(c, variant, fmt):
    """examples/fm_radio.rs:82-90,144-152: RTL-SDR bytes -> convert -> 63-tap FIR -> decimate by 5 -> FM demod,
    as ONE chain node fed the raw bytes (the chain's mixer at dphase 0 = identity).  Equal to the same chain fed
    the converted samples -- bit for bit on the time-domain kernel, which converts in its load stage -- and to the
    oracle's iq_u8_to_c32 -> batch_fir -> decimate -> FM::demod."""
    rng = np.random.default_rng(8)
    n = 5 * 40000
    t = rng.uniform(-0.03, 0.03, 31).astype(np.float32)
    taps = np.concatenate([t, [np.float32(0.18)], t[::-1]]).astype(np.complex64)
    raw, x = raw_stream(rng, n, fmt)
    kw = dict(unfused=variant == "unfused", kernel="auto" if variant == "unfused" else variant)
    a = c.ChainNode(0.0, 0.0, taps, 5, True, **kw).set_input_format(fmt, 1.0 / 8192)
    b = c.ChainNode(0.0, 0.0, taps, 5, True, **kw)
    cuts = [0, 5 * 1000, 5 * 1001, n]
    got = np.concatenate([a.run(raw[p:q]) for p, q in zip(cuts[:-1], cuts[1:])])
    ref = np.concatenate([b.run(x[p:q]) for p, q in zip(cuts[:-1], cuts[1:])])
    assert np.array_equal(got, ref)
    y = oracle.decimate(oracle.batch_fir(x, taps, oracle.default_state(taps), norotate=True), 5)
    want = oracle.FM().demod(y)
    mag = np.minimum(np.abs(y), np.abs(np.concatenate([[0.0], y[:-1]])))
    assert np.max((circ(got.astype(np.float64) - want) * mag)[16:]) <= 4e-5 * np.sum(np.abs(taps))


@pytest.mark.parametrize("fmt", ["i16", "u8"])
@pytest.mark.parametrize("rate,n_taps,kernel", [(7, 63, "time"), (14, 127, "time"), (20, 127, "time_any"), (33, 255, "time_any"),
                                                 (100, 63, "time_any"), (1000, 255, "time_any")])
def test_chains_at_the_rates_of_round_3_from_raw_iq(c, rate, n_taps, kernel, fmt):
    """Raw i16 / u8 samples into the chain kernels added in round 3 -- the per-rate kernel at 7 and 14, the any-rate kernel in
    its LDS-staged (20, 33) and direct (100, 1000) forms -- with the mixer in front of the FIR and FM demod behind the
    decimator: bit for bit what the same chain makes of the converted samples (the conversion sits in the load stage),
    in ragged calls; and the oracle's convert -> mix -> batch_fir -> decimate -> FM::demod."""
    rng = np.random.default_rng(rate + n_taps)
    n = rate * 2600
    k = np.arange(n_taps) - (n_taps - 1) / 2
    taps = (np.sinc(k / (1.2 * rate)) / (1.2 * rate) * np.hamming(n_taps)).astype(np.float32).astype(np.complex64)
    raw, x = raw_stream(rng, n, fmt)
    dphase, phase = 0.05, 0.3
    a = c.ChainNode(dphase, phase, taps, rate, True, kernel="time").set_input_format(fmt, 1.0 / 8192)
    b = c.ChainNode(dphase, phase, taps, rate, True, kernel="time")
    assert a.kernel == kernel and b.kernel == kernel
    cuts = [0, rate * 1, rate * 700, rate * 701, n]
    got = np.concatenate([a.run(raw[p:q]) for p, q in zip(cuts[:-1], cuts[1:])])
    ref = np.concatenate([b.run(x[p:q]) for p, q in zip(cuts[:-1], cuts[1:])])
    assert np.array_equal(got, ref)
    y = oracle.decimate(oracle.batch_fir(oracle.Mixer(phase, dphase).mix(x), taps, oracle.default_state(taps), norotate=True), rate)
    want = oracle.FM().demod(y)
    mag = np.minimum(np.abs(y), np.abs(np.concatenate([[0.0], y[:-1]])))
    skip = n_taps // rate + 2
    assert np.max((circ(got.astype(np.float64) - want) * mag)[skip:]) <= 4e-5 * np.sum(np.abs(taps))


def test_metric_chain_from_i16_full_size(c):
    """The BASELINE metric's chain (255-tap FIR -> mixer -> decimate by 8) on 2^24 i16 samples resident in HBM
    (64 MiB instead of 128): fused load-convert vs. conversion kernel + chain, bit for bit."""
    import torch

    n = 1 << 24
    taps = c.rrc_taps(255, 8.0, 0.35)
    x = torch.empty(n, dtype=torch.complex64, device="cuda:0")
    c.synth_iq_dev(x.data_ptr(), n, 0, 12)
    raw = torch.empty(n, 2, dtype=torch.int16, device="cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    c.lib().comms_iq_c32_to_i16_dev(x.data_ptr(), n, 8192.0, raw.data_ptr(), 0, s)
    c.lib().comms_iq_i16_to_c32_dev(raw.data_ptr(), n, 1.0 / 8192, x.data_ptr(), 0, s)   # x = the converted stream
    a = c.ChainNode(0.6, 0.1, taps, 8, False, mixer_after_fir=True).set_input_format("i16", 1.0 / 8192)
    b = c.ChainNode(0.6, 0.1, taps, 8, False, mixer_after_fir=True)
    assert a.kernel == "time"
    za = torch.empty(n // 8, dtype=torch.complex64, device="cuda:0")
    zb = torch.empty_like(za)
    a.run_dev(raw.data_ptr(), n, za.data_ptr(), s)
    b.run_dev(x.data_ptr(), n, zb.data_ptr(), s)
    torch.cuda.synchronize()
    assert torch.equal(torch.view_as_real(za), torch.view_as_real(zb))
    assert float(za.abs().max()) > 0.1


@pytest.mark.parametrize("n_taps,sps,mix", [(63, 4, True), (63, 4, False), (32, 4, False), (17, 3, True), (100, 7, True),
                                             (31, 2, True), (51, 5, True), (48, 6, False), (127, 8, True), (101, 10, False),
                                             (64, 12, True), (255, 16, True), (100, 20, False), (64, 32, True)])
def test_transmit_chain_writes_the_i16_wire_format(c, n_taps, sps, mix):
    """examples/single_thread_bpsk.rs:29-44: symbols -> pulse shaping [-> mixer] -> `(8192.0 * x) as i16` -> IQOutput.
    The pulse node's store stage writes the i16 pairs itself (polyphase kernel at sps 3 / 4, generic kernel at 7):
    bit-identical to comms_iq_c32_to_i16 of the same node's Complex<f32> output, across two calls, and within one
    LSB of the oracle chain (the f32 outputs differ by rounding, the truncation can then differ by one)."""
    rng = np.random.default_rng(n_taps + sps)
    taps = oracle.rrc_taps(n_taps, float(sps), 0.25)
    sym = ((2 * rng.integers(0, 2, 20000) - 1) + 1j * (2 * rng.integers(0, 2, 20000) - 1)).astype(np.complex64)
    a, b = c.PulseNode(taps, sps), c.PulseNode(taps, sps)
    if mix:
        a.set_mixer(2 * np.pi * 0.1, 0.3)
        b.set_mixer(2 * np.pi * 0.1, 0.3)
    a.set_output_format("i16", 8192.0)
    got = np.concatenate([a.run(sym[:7777]), a.run(sym[7777:])])
    ref = np.concatenate([b.run(sym[:7777]), b.run(sym[7777:])])
    assert got.dtype == np.int16 and got.shape == (sym.size * sps, 2)
    assert np.array_equal(got, c.iq_c32_to_i16(ref, 8192.0))
    y = oracle.batch_fir(oracle.upsample(sym, sps), taps, oracle.default_state(taps), norotate=True)
    if mix:
        y = oracle.Mixer(0.3, 2 * np.pi * 0.1).mix(y)
    want = oracle.iq_c32_to_i16(y, 8192.0)
    assert np.max(np.abs(got.astype(np.int32) - want.astype(np.int32))) <= 1
    # saturation and the default again
    a.set_output_format("i16", 1e9)
    sat = a.run(sym[:16])
    assert np.mean(np.isin(sat, [-32768, 32767, 0])) > 0.9   # (a few outputs of the long filters are ~1e-9: not saturated)
    a.set_output_format("c32")
    assert a.run(sym[:16]).dtype == np.complex64


def test_raw_iq_i16_reference_vector_bit_exact(c):
    """The byte stream of the reference's own raw-IQ tests (src/io/raw_iq.rs:239-300: Complex<i16>(2i, 2i+1), i = 0..99,
    native-endian; tests/golden/reference_kats.json "raw_iq_i16"): comms_iq_i16_to_c32 yields exactly those samples, the FIR
    node's i16 load stage reads them as such, and comms_iq_c32_to_i16 writes the same bytes back."""
    import json

    k = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_kats.json")))["raw_iq_i16"]
    raw = np.ascontiguousarray(np.frombuffer(bytes.fromhex(k["bytes_hex_le"]), "<i2").reshape(-1, 2)).astype(np.int16)
    want = np.array([complex(a, b) for a, b in k["expected"]], np.complex64)
    got = c.iq_i16_to_c32(raw)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(got.view(np.uint32), oracle.iq_i16_to_c32(raw).view(np.uint32))
    assert np.array_equal(c.iq_c32_to_i16(got, 1.0), raw)
    one = c.BatchFirNode(np.array([1 + 0j], np.complex64)).set_algo(c.FIR_DIRECT).set_input_format("i16", 1.0).run(raw)
    assert np.array_equal(one.view(np.uint32), want.view(np.uint32))
