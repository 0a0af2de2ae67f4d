"""GPU parity tests of the Complex<f64> FFT and FM-demodulation nodes (csrc/fft_f64.hip): FFTBatchNode<f64> /
FFTSampleNode<f64> (src/fft/fft_node.rs:24,65-83,99,142-167; BatchFFT::run_fft computes in f64, src/fft/mod.rs:73-96) and
FMDemodNode<f64> (src/modulation/analog.rs:22-35), against the oracle's f64 restatements and numpy's f64 FFT.

Tolerance: ||d||2 / ||X||2 <= 1e-12 (f64 rounding through log2 N layers, the same measure as the f32 nodes' 1e-5); FM angles
<= 1e-12 rad where the samples are not ~0."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c():
    import comms_rs_amd as c

    assert c.device_count() >= 1, "no MI355X visible: the HIP path cannot be tested (no CPU fallback)"
    return c


def rand_z(rng, n):
    return rng.standard_normal(n) + 1j * rng.standard_normal(n)


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def test_fft_f64_reference_golden(c, kats):
    """The reference's own FFT vector (src/fft/fft_node.rs:230-241: N = 10) run through the f64 node."""
    g = kats["fft_fwd_10"]
    x = np.asarray(g["input"], np.float64)
    x = x[:, 0] + 1j * x[:, 1]
    want = np.asarray(g["expected"], np.float64)
    want = want[:, 0] + 1j * want[:, 1]
    got = c.FFTBatchNodeF64(x.size, False).run(x)
    assert np.max(np.abs(got - want)) < g["tol_abs"]
    assert rel(got, np.fft.fft(x)) <= 1e-13


# one pass through LDS (<= 4096), four-step (to 2^24), the DFT sum (other lengths <= 4096), Bluestein (above)
@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("n,batch", [(1, 5), (2, 7), (4, 3), (8, 1), (64, 33), (256, 17), (1024, 5), (4096, 3), (8192, 2), (65536, 3),
                                     (1 << 20, 1), (3, 4), (10, 9), (12, 1), (100, 6), (1000, 2), (4095, 1), (4097, 2), (5000, 1),
                                     (10007, 2), (100000, 1)])
def test_fft_f64_against_numpy_and_oracle(c, n, batch, inverse):
    rng = np.random.default_rng(n + batch)
    x = rand_z(rng, n * batch)
    node = c.FFTBatchNodeF64(n, inverse)
    got = node.run(x).reshape(batch, n)
    xr = x.reshape(batch, n)
    want = np.fft.ifft(xr, axis=1) * n if inverse else np.fft.fft(xr, axis=1)
    assert rel(got, want) <= 1e-12
    if n <= 4096:  # the oracle's restatement of rustfft's transform (O(N^2) beyond its small sizes)
        o = oracle.fft(xr[0].astype(np.complex128), inverse=inverse)
        assert rel(got[0], o) <= 1e-12


def test_fft_f64_2p24_and_round_trip(c):
    rng = np.random.default_rng(24)
    n = 1 << 24
    x = rand_z(rng, n)
    X = c.FFTBatchNodeF64(n, False).run(x)
    assert rel(X, np.fft.fft(x)) <= 1e-12
    back = c.FFTBatchNodeF64(n, True).run(X) / n
    assert rel(back, x) <= 1e-12


def test_fft_f64_errors_and_sample_node(c):
    node = c.FFTBatchNodeF64(16, False)
    with pytest.raises(c.CommsError):
        node.run(np.zeros(15, np.complex128))  # rustfft panics on a wrong length (src/fft/mod.rs:88)
    with pytest.raises(c.CommsError):
        c.FFTBatchNodeF64(0, False)
    with pytest.raises(c.CommsError):
        c.FFTBatchNodeF64((1 << 23) + 1, False)  # its chirp transform would need 2^25 points
    rng = np.random.default_rng(1)
    x = rand_z(rng, 24)
    s = c.FFTSampleNodeF64(8, False)
    outs = [s.run(v) for v in x]
    assert [o is not None for o in outs] == [(i % 8) == 7 for i in range(24)]
    for k in range(3):
        assert rel(outs[8 * k + 7], np.fft.fft(x[8 * k:8 * k + 8])) <= 1e-13


def test_fft_f64_in_place_on_the_device(c):
    import torch

    rng = np.random.default_rng(3)
    for n, batch in [(1024, 8), (1 << 16, 2), (5000, 2)]:
        x = rand_z(rng, n * batch)
        d = torch.from_numpy(x.copy()).to("cuda:0")
        node = c.FFTBatchNodeF64(n, False)
        node.run_dev(d.data_ptr(), n * batch, d.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert rel(d.cpu().numpy().reshape(batch, n), np.fft.fft(x.reshape(batch, n), axis=1)) <= 1e-12
    d = torch.zeros(100, dtype=torch.complex128, device="cuda:0")
    with pytest.raises(c.CommsError):  # the DFT sum reads its inputs while other workgroups write outputs
        c.FFTBatchNodeF64(100, False).run_dev(d.data_ptr(), 100, d.data_ptr(), 0)


def test_fmdemod_f64_against_oracle(c):
    rng = np.random.default_rng(9)
    n = 50000
    t = np.arange(n)
    x = np.exp(1j * (0.3 * t + 2.0 * np.sin(t / 300.0))) * (1 + 0.05 * rng.standard_normal(n))
    node, ofm = c.FMDemodNodeF64(), oracle.FM(np.complex128)
    for a, b in [(0, 1), (1, 2), (2, 1000), (1000, 1001), (1001, n)]:
        got, want = node.run(x[a:b]), ofm.demod(x[a:b])
        assert got.dtype == np.float64
        d = np.abs((got - want + np.pi) % (2 * np.pi) - np.pi)
        assert np.max(d) <= 1e-12
    assert node.prev == x[-1]
    node.prev = 1j  # the 1-sample halo of a sharded stream
    assert abs(node.run(np.array([1.0 + 0j]))[0] - np.angle(1.0 * np.conj(1j))) <= 1e-15
