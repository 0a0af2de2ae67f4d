"""Pins the CPU oracle (oracle/oracle.cpp) against every golden vector the
reference's own tests hold for the hot path.  CPU only."""
import numpy as np
import pytest

import oracle


def cx(a, dtype):
    a = np.asarray(a, dtype=np.float64)
    return (a[:, 0] + 1j * a[:, 1]).astype(dtype)


def test_fir_i16_sample_by_sample(kats):
    # src/filter/fir_node.rs:235-338 (FirNode, one sample per call)
    k = kats["fir_i16"]
    taps = np.array(k["taps"], np.int16)
    state = np.zeros_like(taps)  # `None` state -> zeros of taps.len()
    got = [oracle.fir(np.array(x, np.int16), taps, state) for x in k["input"]]
    assert np.array_equal(np.array(got)[:9], np.array(k["expected"], np.int16))


def test_fir_i16_batch_non_vacuous(kats):
    # src/filter/fir_node.rs:342-449 is vacuous in the reference (its assert
    # never fires); run the same data through batch_fir two samples at a time,
    # exercising state carry-over across run() calls.
    k = kats["fir_i16"]
    taps = np.array(k["taps"], np.int16)
    state = np.zeros_like(taps)
    x = np.array(k["input"], np.int16)
    got = np.concatenate([oracle.batch_fir(x[i:i + 2], taps, state) for i in range(0, 10, 2)])
    assert np.array_equal(got[:9], np.array(k["expected"], np.int16))


def test_fir_f32_matches_i16_golden_and_norotate(kats):
    k = kats["fir_i16"]
    taps = cx(k["taps"], np.complex64)
    x = cx(k["input"], np.complex64)
    want = cx(k["expected"], np.complex64)
    for norot in (False, True):
        st = oracle.default_state(taps)
        got = oracle.batch_fir(x, taps, st, norotate=norot)
        assert np.array_equal(got[:9], want)


def test_norotate_variant_is_bit_identical():
    rng = np.random.default_rng(1)
    for n_taps, n in [(1, 7), (5, 64), (63, 1000), (255, 2048)]:
        taps = (rng.standard_normal(n_taps) + 1j * rng.standard_normal(n_taps)).astype(np.complex64)
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        s0 = (rng.standard_normal(n_taps) + 1j * rng.standard_normal(n_taps)).astype(np.complex64)
        s1, s2 = s0.copy(), s0.copy()
        a = np.concatenate([oracle.batch_fir(x[:n // 3], taps, s1), oracle.batch_fir(x[n // 3:], taps, s1)])
        b = np.concatenate([oracle.batch_fir(x[:n // 2], taps, s2, norotate=True),
                            oracle.batch_fir(x[n // 2:], taps, s2, norotate=True)])
        assert a.tobytes() == b.tobytes()
        assert s1.tobytes() == s2.tobytes()


def test_fir_short_state_truncates_taps():
    # fir.rs:53 zips taps with state: a user state shorter than the taps uses
    # only the first state.len() taps; a longer one behaves like taps.len().
    rng = np.random.default_rng(11)
    taps = (rng.standard_normal(9) + 1j * rng.standard_normal(9)).astype(np.complex64)
    x = (rng.standard_normal(50) + 1j * rng.standard_normal(50)).astype(np.complex64)
    short = np.zeros(4, np.complex64)
    a = oracle.batch_fir(x, taps, short)
    b = oracle.batch_fir(x, taps[:4].copy(), np.zeros(4, np.complex64))
    assert a.tobytes() == b.tobytes()
    long = np.zeros(20, np.complex64)
    c = oracle.batch_fir(x, taps, long)
    d = oracle.batch_fir(x, taps, np.zeros(9, np.complex64))
    assert c.tobytes() == d.tobytes()


def test_fir_independent_definition_f64():
    # cross-check against numpy's convolution in f64 (independent definition)
    rng = np.random.default_rng(2)
    taps = (rng.standard_normal(31) + 1j * rng.standard_normal(31)).astype(np.complex128)
    x = (rng.standard_normal(500) + 1j * rng.standard_normal(500)).astype(np.complex128)
    st = oracle.default_state(taps)
    got = oracle.batch_fir(x, taps, st)
    want = np.convolve(x, taps)[:500]
    assert np.max(np.abs(got - want)) < 1e-12


def test_fft_forward_golden(kats):
    # src/fft/fft_node.rs:180-263
    k = kats["fft_fwd_10"]
    got = oracle.fft(cx(k["input"], np.complex64), inverse=False)
    assert np.max(np.abs(got - cx(k["expected"], np.complex128))) < k["tol_abs"]


@pytest.mark.parametrize("n", [1, 2, 3, 7, 10, 16, 60, 128, 1000, 1024])
def test_fft_matches_numpy_both_directions(n):
    # IFFT is unpinned by the reference (no test): cross-check the restatement
    # against an independent definition (numpy, unnormalised inverse = ifft*N).
    rng = np.random.default_rng(n)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex128)
    f = oracle.fft(x, inverse=False)
    b = oracle.fft(x, inverse=True)
    scale = np.linalg.norm(x) * np.sqrt(n)
    assert np.max(np.abs(f - np.fft.fft(x))) < 1e-13 * scale
    assert np.max(np.abs(b - np.fft.ifft(x) * n)) < 1e-13 * scale


@pytest.mark.parametrize("key", ["mixer_phase0", "mixer_phase0p1"])
def test_mixer_golden(kats, key):
    # src/mixer.rs:160-246, :250-336 ; MixerNode::new(dphase, phase) -> Mixer::new(phase, dphase)
    k = kats[key]
    m = oracle.Mixer(k["phase"], k["dphase"])
    got = m.mix(cx(k["input"], np.complex128))
    want = cx(k["expected"], np.complex128)
    assert np.max(np.abs(got.real - want.real)) < k["tol_abs"]
    assert np.max(np.abs(got.imag - want.imag)) < k["tol_abs"]


def test_mixer_state_persists_and_dphase_wraps():
    x = (np.arange(40) + 1j).astype(np.complex64)
    a = oracle.Mixer(0.3, 0.7).mix(x)
    m = oracle.Mixer(0.3, 0.7 + 4 * np.pi)  # Mixer::new wraps dphase into [0, 2pi)
    b = np.concatenate([m.mix(x[:13]), m.mix(x[13:])])
    assert np.allclose(a, b, rtol=0, atol=1e-4)
    c = oracle.Mixer(0.3, 0.7 - 2 * np.pi).mix(x)
    assert np.allclose(a, c, rtol=0, atol=1e-4)
    # independent closed form
    want = x.astype(np.complex128) * np.exp(1j * (0.3 + 0.7 * np.arange(40)))
    assert np.max(np.abs(a - want)) < 1e-4


def test_pulse_golden(kats):
    # src/pulse.rs:105-209
    k = kats["pulse_rect_i16"]
    taps = oracle.rect_taps(k["n_taps"], np.int16)
    state = np.zeros_like(taps)
    sym = np.array(k["symbols"], np.int16)
    # one symbol per run() call, as PulseNode does
    got = np.concatenate([oracle.pulse(sym[i:i + 1], taps, k["sam_per_sym"], state) for i in range(len(sym))])
    assert np.array_equal(got, np.array(k["expected"], np.int16))


def test_pulse_equals_upsample_then_fir():
    rng = np.random.default_rng(3)
    taps = oracle.rrc_taps(63, 4.0, 0.25)
    sym = (rng.integers(0, 2, 200) * 2 - 1).astype(np.complex64)
    a = oracle.pulse(sym, taps, 4, oracle.default_state(taps))
    b = oracle.batch_fir(oracle.upsample(sym, 4), taps, oracle.default_state(taps))
    assert a.tobytes() == b.tobytes()


@pytest.mark.parametrize("key,fn", [("rrc_taps", "rrc_taps"), ("rc_taps", "rc_taps"), ("gaussian_taps", "gaussian_taps")])
def test_tap_design_golden(kats, key, fn):
    # src/util/math.rs:359-488
    k = kats[key]
    shape = k.get("beta", k.get("alpha"))
    got = getattr(oracle, fn)(k["n_taps"], k["sam_per_sym"], shape, dtype=np.complex128)
    want = np.array(k["expected_re"])
    assert np.all(got.imag == 0)
    assert np.max(np.abs(got.real - want)) < k["tol_abs"]


def test_rrc_special_branches_and_errors():
    # 63 taps at sps=4, beta=0.25 hits t=0 and t=+-1/(4 beta)=+-1 exactly
    t = oracle.rrc_taps(63, 4.0, 0.25, dtype=np.complex128).real
    assert np.allclose(t, t[::-1], atol=1e-15)
    assert abs(t[31] - (1 + 0.25 * (4 / np.pi - 1))) < 1e-15
    assert np.all(np.isfinite(t))
    for bad in (-0.1, 1.1):
        with pytest.raises(oracle.InvalidRolloffError):
            oracle.rrc_taps(8, 4.0, bad)
        with pytest.raises(oracle.InvalidRolloffError):
            oracle.rc_taps(8, 4.0, bad)


def test_rect_and_sinc(kats):
    assert np.array_equal(oracle.rect_taps(12), np.ones(12, np.complex64))
    for x, want in kats["sinc"]["points"]:
        assert abs(oracle.sinc(x) - want) < kats["sinc"]["tol_abs"]


def test_decimate_upsample_golden(kats):
    # src/util/resample_node.rs:139-175 + doctests
    for c in kats["decimate"]["cases"]:
        assert oracle.decimate(np.array(c["input"], np.int32), c["rate"]).tolist() == c["expected"]
    for c in kats["upsample"]["cases"]:
        assert oracle.upsample(np.array(c["input"], np.int32), c["rate"]).tolist() == c["expected"]
    # any Copy type: complex64 and bytes
    z = (np.arange(10) + 2j).astype(np.complex64)
    assert np.array_equal(oracle.decimate(z, 4), z[::4])
    assert np.array_equal(oracle.upsample(z, 2)[::2], z)
    assert oracle.decimate(np.zeros(0, np.float32), 3).size == 0


def test_fm_demod_properties():
    # FM demod is UNPINNED by the reference (no test).  Cross-check against the
    # independent definition angle(x[n] * conj(x[n-1])) and pin the signed-zero
    # first-sample edge (analog.rs:27-28,45): prev = 0+0i, x0 in quadrant III -> pi.
    rng = np.random.default_rng(4)
    x = (rng.standard_normal(1000) + 1j * rng.standard_normal(1000)).astype(np.complex64)
    fm = oracle.FM()
    got = np.concatenate([fm.demod(x[:333]), fm.demod(x[333:])])
    x64 = x.astype(np.complex128)
    want = np.angle(x64[1:] * np.conj(x64[:-1]))
    d = np.abs(got[1:] - want)
    d = np.minimum(d, 2 * np.pi - d)
    assert np.max(d) < 1e-5
    assert oracle.FM().demod(np.array([-1 - 1j], np.complex64))[0] == np.float32(np.pi)
    assert oracle.FM().demod(np.array([1 + 1j], np.complex64))[0] == 0.0


def test_prbs7_golden(kats):
    # src/prns.rs:191-221
    k = kats["prbs7"]
    bits, _ = oracle.prns_u8(k["poly_mask"], k["state"], 128)
    assert bits.tolist() == k["expected"]
    assert oracle.prns_u8(0xC0, 0xFF, 1)[0][0] == 1
    # maximal-length PRBS8 (prns.rs:146-189, golden "prns8"): 255 distinct states, then the first again
    k8 = kats["prns8"]
    seen, st = set(), k8["state"]
    for _ in range(k8["distinct_states"]):
        assert st not in seen
        seen.add(st)
        _, st = oracle.prns_u8(k8["poly_mask"], st, 1)
    assert st == k8["state"] and len(seen) == 255


def test_raw_iq_i16_golden(kats):
    # src/io/raw_iq.rs:239-300: the byte stream the reference's IQInput tests read, and what its samples are; the
    # Complex<f32> the hot path then sees is cast_complex of them (util/math.rs:20-28), exactly
    k = kats["raw_iq_i16"]
    raw = np.frombuffer(bytes.fromhex(k["bytes_hex_le"]), "<i2").reshape(-1, 2)
    assert raw.shape == (k["n"], 2) and raw.tolist() == k["expected"]
    got = oracle.iq_i16_to_c32(np.ascontiguousarray(raw).astype(np.int16))
    want = np.array([complex(a, b) for a, b in k["expected"]], np.complex64)
    assert np.array_equal(got, want)
    assert np.array_equal(oracle.iq_c32_to_i16(got, 1.0), raw)   # and back: what IQOutput would write (raw_iq.rs:303-372)


def test_iq_wire_formats():
    # raw_iq.rs:16,50-51 / single_thread_bpsk.rs:40-44 / fm_radio.rs:82-90
    i16 = np.array([[3, -4], [32767, -32768], [0, 1]], np.int16)
    assert np.array_equal(oracle.iq_i16_to_c32(i16), np.array([3 - 4j, 32767 - 32768j, 1j], np.complex64))
    x = np.array([0.5 - 0.25j, 100 - 100j, np.nan + 0.9999j, -0.00013 + 0.00013j], np.complex64)
    assert oracle.iq_c32_to_i16(x, 8192.0).tolist() == [[4096, -2048], [32767, -32768], [0, 8191], [-1, 1]]
    u8 = np.array([[0, 255], [127, 128]], np.uint8)
    got = oracle.iq_u8_to_c32(u8)
    assert got[0] == np.complex64(-1 + 1j) and abs(got[1].real + 0.5 / 127.5) < 1e-7
