"""Lane-accurate numpy model of fir_poly8_kernel (fir_poly8.hip): FIR -> keep every 8th output, computed as eight
polyphase branches in the frequency domain (128-point transforms), one wave per 1024-sample segment; and the second output
phase y[8j + 4] (rate 4) from the same forward transforms through the branch spectra of h[8m - c + 4].

CPU only; no GPU, no library.  Every array has a leading axis of 64 lanes; LDS exchanges are gathers on flat
arrays with the kernel's strides; the lane swaps follow the semantics probed on the device
(profiles/r01_probe_valu.txt).  Checks the result against the direct sum  y[8j] = sum_k h[k] x[8j - k].
"""
import numpy as np

rng = np.random.default_rng(1)
T = 255
h = (rng.standard_normal(T) + 1j * rng.standard_normal(T)).astype(np.complex128)
NSEG = 5
n = 768 * NSEG
x = (rng.standard_normal(n) + 1j * rng.standard_normal(n))
hist = (rng.standard_normal(256) + 1j * rng.standard_normal(256))  # x[-256 .. -1]
xs = np.concatenate([hist, x, np.zeros(1024)])  # xs[i + 256] = x[i]

# ---- reference: y[8j] and y[8j + 4]
ref = np.zeros((2, n // 8), complex)
for ph in range(2):
    for j in range(n // 8):
        acc = 0
        for k in range(T):
            acc += h[k] * xs[8 * j + 4 * ph - k + 256]
        ref[ph, j] = acc

# ---- tables
lane = np.arange(64)
W = lambda N, e: np.exp(-2j * np.pi * (np.asarray(e) % N) / N)
# g_c[m] = h[8m - c + 4 ph], m = 0..32 ; G_c[k] = sum_m g_c[m] W128^{mk} / 128   (ph = 0: y[8j]; ph = 1: y[8j + 4])
GG = np.zeros((2, 8, 128), complex)
for ph in range(2):
    for c in range(8):
        g = np.zeros(128, complex)
        for m in range(33):
            k = 8 * m - c + 4 * ph
            if 0 <= k < T:
                g[m] = h[k]
        GG[ph, c] = np.fft.fft(g) / 128.0
S1, S2, S3 = 66, 34, 17


def swap32(a, b):  # a' = [a.lo | b.lo], b' = [a.hi | b.hi]
    a2 = np.concatenate([a[:32], b[:32]])
    b2 = np.concatenate([a[32:], b[32:]])
    return a2, b2


def swap16(a, b):  # a' = [a.r0, b.r0, a.r2, b.r2], b' = [a.r1, b.r1, a.r3, b.r3]
    r = lambda v, i: v[16 * i:16 * i + 16]
    a2 = np.concatenate([r(a, 0), r(b, 0), r(a, 2), r(b, 2)])
    b2 = np.concatenate([r(a, 1), r(b, 1), r(a, 3), r(b, 3)])
    return a2, b2


out = np.zeros((2, n // 8), complex)
for s in range(NSEG):
    base = 768 * s - 256
    # rows: v[a][l] = x[base + 64 a + l]
    v = np.array([xs[base + 64 * a + lane + 256] for a in range(16)])  # [16][64]
    c = lane & 7
    d = lane >> 3
    # step 1: DFT16 over a
    A = np.array([sum(v[a] * W(16, a * k1) for a in range(16)) for k1 in range(16)])
    # step 2: twiddle W128^{d k1}
    B = np.array([A[k1] * W(128, d * k1) for k1 in range(16)])
    # exchange 1
    lds = np.zeros(16 * S1, complex)
    for k1 in range(16):
        lds[k1 * S1 + lane] = B[k1]
    k1l = lane & 15
    cg = lane >> 4
    xx = np.zeros((2, 8, 64), complex)
    for ci in range(2):
        for dd in range(8):
            xx[ci, dd] = lds[k1l * S1 + (cg + 4 * ci) + 8 * dd]
    # step 3: DFT8 over d
    V = np.zeros((2, 8, 64), complex)
    for ci in range(2):
        for k2 in range(8):
            V[ci, k2] = sum(xx[ci, dd] * W(8, dd * k2) for dd in range(8))
    for ph in range(2):  # the forward transforms V serve both output phases
        # step 4: spectrum MAC
        P = np.zeros((8, 64), complex)
        for k2 in range(8):
            for ci in range(2):
                P[k2] += GG[ph, cg + 4 * ci, k1l + 16 * k2] * V[ci, k2]
        # exchange 2 + reduction over cg
        lds = np.zeros(16 * S2, complex)
        for k2 in range(8):
            lds[k1l * S2 + 4 * k2 + cg] = P[k2]
        j = lane >> 4
        Za = sum(lds[k1l * S2 + 4 * j + g] for g in range(4))
        Zb = sum(lds[k1l * S2 + 4 * (j + 4) + g] for g in range(4))
        # step 5: inverse DFT8 over k2, distributed (in-lane 2 x lane bits 4, 5)
        iW = lambda N, e: np.conj(W(N, e))
        a_ = Za + Zb
        b_ = (Za - Zb) * iW(8, j)
        a_, b_ = swap32(a_, b_)
        s_ = a_ + b_
        d_ = a_ - b_
        j0 = (lane >> 4) & 1
        j1 = lane >> 5
        d_ = d_ * np.where(j0 == 1, 1j, 1.0)
        s_, d_ = swap16(s_, d_)
        u = s_ + d_
        w = s_ - d_
        q1a = 2 * j0 + j1
        # step 6: twiddle omega128^{q1 k1}
        U0 = u * iW(128, q1a * k1l)
        U1 = w * iW(128, (q1a + 4) * k1l)
        # exchange 3
        lds = np.zeros(8 * S3, complex)
        lds[q1a * S3 + k1l] = U0
        lds[(q1a + 4) * S3 + k1l] = U1
        mu = lane & 31
        q1 = mu & 7
        k1a = mu >> 3
        y = np.array([lds[q1 * S3 + k1a + 4 * k1b] for k1b in range(4)])
        # step 7: inverse DFT4 over k1b, twiddle omega16^{q2a k1a}
        M = np.array([sum(y[k1b] * iW(4, q2a * k1b) for k1b in range(4)) * iW(16, q2a * k1a) for q2a in range(4)])
        # exchange 4: [q1][q2a][k1a]
        lds = np.zeros(128, complex)
        for q2a in range(4):
            lds[(q1 * 4 + q2a) * 4 + k1a] = M[q2a]
        q2a_l = mu >> 3
        m = np.array([lds[(q1 * 4 + q2a_l) * 4 + ka] for ka in range(4)])
        z = np.array([sum(m[ka] * iW(4, q2b * ka) for ka in range(4)) for q2b in range(4)])
        for q2b in range(1, 4):
            for l in range(32):
                jo = 96 * s + l + 32 * (q2b - 1)
                if jo < n // 8:
                    out[ph, jo] = z[q2b, l]

err = np.max(np.abs(out - ref)) / np.max(np.abs(ref))
print("max rel err", err)
assert err < 1e-12
print("OK")
