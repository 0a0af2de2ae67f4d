// fft.hip -- FFT / IFFT node: unnormalised DFT of Complex<f32> blocks on gfx950.
//
// Replaces FFTBatchNode::new(fft_size, ifft) + BatchFFT::run_fft (reference
// src/fft/fft_node.rs:65-74, src/fft/mod.rs:73-96; arithmetic by rustfft 2.1.0
// in f64).  Here the transform runs in f32 with twiddles generated in f64 on
// the host and rounded once.
//
// Kernels, by length N:
//   * 2 ... 16384 (powers of two): fft_rx1024_kernel, ONE pass.  A 16-wave workgroup owns 16 rows
//     of 1024 points; every wave runs a barrier-free radix-16 register transform on its row
//     (1024 points, 16 x 64, 4 x 256, or 16 consecutive points per lane for 2 ... 16), 2048 ... 16384
//     add a radix-RAD butterfly over the rows on the way into LDS, 32 / 128 / 512 a radix-2 one;
//     loads are one tile ahead, stores fully coalesced.
//   * 2^15 ... 2^20 (four-step, N = N1 * 1024): pass 1 = fft_cols_kernel (N1 = 32 ... 512: the same
//     wave forms on columns, x W_N^{n2*k1}) or fft1024x16_kernel (N1 = 1024, config 4);
//     pass 2 = fft1024x16_kernel on the rows with the transposed store.
//   * 2^21 ... 2^24: fft1024x16_kernel on 1024-point columns (stride N/1024, four-step twiddle), fft_rx1024_kernel on
//     the N/1024-point rows in place, fft_transpose_kernel for the output order (three launches, 48 B/point).
//   * the ragged tail of a batch (and the COMMS_FFT_NO_* fallbacks): fft_tile_kernel -- radix-4 Stockham passes in LDS on
//     tiles of C sub-transforms of length L (C*L <= 16384 points), one pass or four-step.
//   * other lengths: exact-index O(N^2) DFT with f64 accumulation (N <= 64) or Bluestein's
//     chirp-z on the power-of-two kernels (N > 64).
#include <cmath>
#include <cstdlib>
#include <vector>

#include "common.hpp"
#include "fft_radix.hpp"
#include "fir_handle.hpp"  // make_rsrc: buffer-addressed rows

namespace comms {

constexpr int FT_MAX_POINTS = 16384;  // per tile
constexpr int FT_PTS = 16;            // points per lane per pass

struct FftTileParams {
    int L, logL, C, logC;
    int in_c_fast, out_c_fast;       // which tile index runs along the lanes for global I/O
    size_t in_cs, in_ls;             // element (c,l): in_off(tile) + c*in_cs + l*in_ls
    size_t out_cs, out_ks;           // element (c,k): out_off(tile) + c*out_cs + k*out_ks
    size_t tiles_per_xform;          // tiles inside one length-N transform
    size_t tile_step_in, tile_step_out;
    size_t N;                        // distance between transforms of the batch
    size_t n_tiles;                  // total (batch * tiles_per_xform)
    const cf* twL;               // W_L^m, m < L (forward sign)
    int apply_tw;                    // four-step pass 1: times W_N^{(tile*C + c) * k}
    const cf* tw_lo;             // W_N^e, e < 4096
    const cf* tw_hi;             // W_N^{4096 e}
    KStamp ks;                   // in-kernel begin / end stamps of a stamps timer (both passes of a call stamp the same
                                 // slots: first pass in to second pass out), or null
};

// The generic tile kernel keeps plain C++ complex arithmetic (the compiler schedules
// it freely across its guarded, partially unrolled passes); the hand-packed butterflies
// of fft_radix.hpp are for the fully unrolled radix-16 kernels.
__device__ __forceinline__ cf g_mul(cf a, cf b) {
    return cf{__builtin_fmaf(-a.y, b.y, a.x * b.x), __builtin_fmaf(a.y, b.x, a.x * b.y)};
}
__device__ __forceinline__ cf g_mulc(cf a, cf b) {
    return cf{__builtin_fmaf(a.y, b.y, a.x * b.x), __builtin_fmaf(a.y, b.x, -(a.x * b.y))};
}
template <int DIR>
__device__ __forceinline__ cf tw_apply(cf x, cf w) {
    return DIR < 0 ? g_mul(x, w) : g_mulc(x, w);
}
template <int DIR>
__device__ __forceinline__ void g_radix4(cf& a, cf& b, cf& c, cf& d) {
    const cf t0 = a + c, t1 = a - c, t2 = b + d, u = b - d;
    const cf t3 = DIR < 0 ? cf{u.y, -u.x} : cf{-u.y, u.x};
    a = t0 + t2;
    c = t0 - t2;
    b = t1 + t3;
    d = t1 - t3;
}

template <int DIR>
__global__ __launch_bounds__(1024) void fft_tile_kernel(const cf* in, cf* out,
                                                        FftTileParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* tile = reinterpret_cast<cf*>(smem);  // [c][l], row stride L
    const int T = blockDim.x;
    const int tid = threadIdx.x;
    const int L = p.L, C = p.C;
    const int npts = L * C;

    for (size_t tix = blockIdx.x; tix < p.n_tiles; tix += gridDim.x) {
        const size_t b = tix / p.tiles_per_xform;
        const size_t tl = tix - b * p.tiles_per_xform;
        const cf* src = in + b * p.N + tl * p.tile_step_in;
        cf* dst = out + b * p.N + tl * p.tile_step_out;

        // ---- load tile
        for (int i = tid; i < npts; i += T) {
            int c, l;
            if (p.in_c_fast) {
                c = i & (C - 1);
                l = i >> p.logC;
            } else {
                l = i & (L - 1);
                c = i >> p.logL;
            }
            tile[c * L + l] = src[c * p.in_cs + l * p.in_ls];
        }
        __syncthreads();

        // ---- Stockham passes, in place (read all -> barrier -> write all)
        int Ns = 1;
        int logNs = 0;
        for (; (Ns << 2) <= L; Ns <<= 2, logNs += 2) {
            cf v[FT_PTS];
            const int nb = npts >> 2;  // radix-4 butterflies in the tile
            const int q = L >> 2;
#pragma unroll
            for (int u = 0; u < FT_PTS / 4; ++u) {
                const int w = tid + u * T;
                if (w < nb) {
                    const int c = w / q, j = w - c * q;
                    const cf* row = tile + c * L;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[4 * u + r] = row[j + r * q];
                }
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < FT_PTS / 4; ++u) {
                const int w = tid + u * T;
                if (w < nb) {
                    const int c = w / q, j = w - c * q;
                    const int k = j & (Ns - 1);
                    const int estep = k * (L >> (logNs + 2));  // k * L/(4 Ns)
                    if (k) {
                        v[4 * u + 1] = tw_apply<DIR>(v[4 * u + 1], p.twL[estep]);
                        v[4 * u + 2] = tw_apply<DIR>(v[4 * u + 2], p.twL[2 * estep]);
                        v[4 * u + 3] = tw_apply<DIR>(v[4 * u + 3], p.twL[3 * estep]);
                    }
                    g_radix4<DIR>(v[4 * u], v[4 * u + 1], v[4 * u + 2], v[4 * u + 3]);
                    const int j0 = ((j >> logNs) << (logNs + 2)) + k;
                    cf* row = tile + c * L;
#pragma unroll
                    for (int r = 0; r < 4; ++r) row[j0 + r * Ns] = v[4 * u + r];
                }
            }
            __syncthreads();
        }
        if (Ns < L) {  // one radix-2 pass left (Ns == L/2)
            cf v[FT_PTS];
            const int nb = npts >> 1;
            const int q = L >> 1;
#pragma unroll
            for (int u = 0; u < FT_PTS / 2; ++u) {
                const int w = tid + u * T;
                if (w < nb) {
                    const int c = w / q, j = w - c * q;
                    v[2 * u] = tile[c * L + j];
                    v[2 * u + 1] = tile[c * L + j + q];
                }
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < FT_PTS / 2; ++u) {
                const int w = tid + u * T;
                if (w < nb) {
                    const int c = w / q, j = w - c * q;
                    // Ns == q here, so k == j and the twiddle is W_L^j
                    cf a = v[2 * u];
                    cf bb = j ? tw_apply<DIR>(v[2 * u + 1], p.twL[j]) : v[2 * u + 1];
                    tile[c * L + j] = a + bb;
                    tile[c * L + j + q] = a - bb;
                }
            }
            __syncthreads();
        }

        // ---- store tile (optionally times the four-step twiddle)
        for (int i = tid; i < npts; i += T) {
            int c, k;
            if (p.out_c_fast) {
                c = i & (C - 1);
                k = i >> p.logC;
            } else {
                k = i & (L - 1);
                c = i >> p.logL;
            }
            cf x = tile[c * L + k];
            if (p.apply_tw) {
                const size_t e = (tl * static_cast<size_t>(C) + c) * static_cast<size_t>(k);  // < N
                const cf w = g_mul(p.tw_hi[e >> 12], p.tw_lo[e & 4095]);
                x = tw_apply<DIR>(x, w);
            }
            dst[c * p.out_cs + k * p.out_ks] = x;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- 16 x 1024-point tile kernel
// The fast path for L = 1024, C = 16 (a 2^20-point transform is two passes of it; a
// batch of 1024-point transforms is one).  16 waves per workgroup; the tile is loaded
// coalesced into 16 LDS buffers (one sub-transform each), every wave then runs ITS
// 1024-point transform on its own buffer with the radix-16 register core of
// fft_radix.hpp (n = 64a + 4b + c; R16 over a, xW1024, exchange, R16 over b, xW64,
// exchange, 4 x R4 over c) with no workgroup barrier, writes the spectrum back in
// natural order, and the tile is stored coalesced (transposed / twiddled as the pass
// needs).  LDS: 16 x 1090 complex + W1024 / W64 tables = 148 KiB.
constexpr int FW_BUF = 1090;  // per-wave buffer stride (elements): 2180 dwords = 4 (mod 64 banks)
constexpr int FW_S1 = 66;
constexpr int FW_P = 272;

__device__ __forceinline__ void fw_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int DIR, int NW, bool TWS = false>  // TWS: the four-step store twiddle in its factored form (pass 1 of N >= 2^20)
__global__ __launch_bounds__(64 * NW, NW == 16 ? 4 : 4) void fft1024x16_kernel(const cf* in, cf* out, FftTileParams p,
                                                             const cf* __restrict__ tw1g,
                                                             const cf* __restrict__ tw2g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* tw1 = reinterpret_cast<cf*>(smem);  // [16][64]  W1024^{lane*k0}
    cf* tw2 = tw1 + 1024;                   // [16][4]   W64^{c*k1}
    cf* bufs = tw2 + 64;                    // [16][FW_BUF]
    cf* twu = bufs + NW * FW_BUF;           // [16][16]  W_N^{64 C u} of the tile's columns (four-step store twiddle)
    const int tid = threadIdx.x;
    const int l = tid & 63, wave = tid >> 6;
    const int q0 = l & 15, q1 = l >> 4;
    cf* buf = bufs + wave * FW_BUF;
    kstamp_begin(p.ks);
    for (int i = tid; i < 1024; i += 64 * NW) tw1[i] = tw1g[i];
    if (tid < 64) tw2[tid] = tw2g[tid];

    // A tile's 16 elements per lane: global element u of this lane and where it goes in LDS.
    constexpr int LOG = NW == 16 ? 4 : 3;
    // Column tiles (in_c_fast): which (column, row) of the tile a thread loads and puts into LDS.  A wave-instruction
    // covers four rows x sixteen adjacent columns either way (whole 128-byte pieces); WITHIN it, sixteen consecutive
    // lanes used to hold the sixteen columns of one row -- sixteen ds_write_b64 to buffers 2180 dwords apart, banks
    // 4 c mod 32: columns c and c + 8 on the same pair of banks, two-way conflicts on every tile-in write (pass 1 of the
    // two-pass sizes: 18 % of its LDS cycles were conflict cycles against 10 % in pass 2, profiles/r04_pmc_config4.txt).
    // Now a group of sixteen lanes holds eight columns x two rows: banks 4 c + 2 r, all thirty-two pairs distinct.
    const int lc = NW == 16 ? ((l & 7) | ((l >> 1) & 8)) : (tid & (NW - 1));
    const int lr = NW == 16 ? (4 * wave + (((l >> 3) & 1) | ((l >> 4) & 2))) : (tid >> LOG);
    auto tile_src = [&](size_t tix) {
        const size_t b = tix / p.tiles_per_xform;
        return in + b * p.N + (tix - b * p.tiles_per_xform) * p.tile_step_in;
    };
    // element offsets inside one transform fit 32 bits (N <= 2^24): uniform 64-bit base in
    // SGPRs + one 32-bit VGPR offset per load keeps the 16 in-flight loads cheap in registers
    const unsigned in_cs = static_cast<unsigned>(p.in_cs), in_ls = static_cast<unsigned>(p.in_ls);
    auto fetch = [&](const cf* src, cf (&r)[16]) {
        if (p.in_c_fast) {  // lane holds (column lc, rows lr + 64u)
            const unsigned o0 = lc * in_cs + lr * in_ls, step = 64u * in_ls;
#pragma unroll
            for (int u = 0; u < 16; ++u) r[u] = src[o0 + u * step];
        } else if (NW == 16) {  // element i = tid + 1024u -> (c = u, r = tid)
            const unsigned o0 = tid * in_ls;
#pragma unroll
            for (int u = 0; u < 16; ++u) r[u] = src[o0 + u * in_cs];
        } else {  // element i = tid + 512u -> (c = u/2, r = tid + 512 (u&1))
            const unsigned o0 = tid * in_ls, o1 = o0 + 512u * in_ls;
#pragma unroll
            for (int u = 0; u < 16; ++u) r[u] = src[((u & 1) ? o1 : o0) + (u >> 1) * in_cs];
        }
    };
    // software pipeline: the next tile's elements are requested into VGPRs before this
    // tile's transform and land in LDS at the top of the next iteration -- with one
    // workgroup per CU nothing else would cover their HBM latency
    // (two 8-wave workgroups per CU already cover each other, and there the extra 32
    // VGPRs spill -- so only the 16-wave form prefetches)
    constexpr bool PREFETCH = NW == 16;
    cf pre[16];
    if (PREFETCH && blockIdx.x < p.n_tiles) fetch(tile_src(blockIdx.x), pre);

    for (size_t tix = blockIdx.x; tix < p.n_tiles; tix += gridDim.x) {
        const size_t b = tix / p.tiles_per_xform;
        const size_t tl = tix - b * p.tiles_per_xform;
        cf* dst = out + b * p.N + tl * p.tile_step_out;
        __syncthreads();  // previous tile fully stored (and the tables are in place)
        // Four-step twiddle of the store stage, W_N^{C k} with C = col0 + c and k = (tid >> 4) + 64 u: factored into
        // W_N^{C (tid >> 4)} (one value per thread) and W_N^{64 C u} (one per column and u: 256 per tile, in LDS), both
        // fetched here, a whole transform ahead of their use.  (Looked up per stored element -- two dependent gathers
        // and a product each, waited for one at a time -- the stores of pass 1 were a chain of sixteen L2 round trips.)
        cf tw_a = cf{1.f, 0.f};
        constexpr bool tw_split = TWS;
        if (tw_split) {
            const unsigned col0 = static_cast<unsigned>(tl) * NW;
            const unsigned ea = (col0 + (tid & 15)) * static_cast<unsigned>(tid >> 4);
            tw_a = g_mul(p.tw_hi[ea >> 12], p.tw_lo[ea & 4095]);
            if (tid < 256) {
                const unsigned eb = (col0 + (tid & 15)) * 64u * static_cast<unsigned>(tid >> 4);
                twu[tid] = g_mul(p.tw_hi[eb >> 12], p.tw_lo[eb & 4095]);
            }
        }
        if (!PREFETCH) fetch(tile_src(tix), pre);
        if (p.in_c_fast) {
            cf* d = bufs + lc * FW_BUF + lr;
#pragma unroll
            for (int u = 0; u < 16; ++u) d[u * 64] = pre[u];
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int i = tid + 64 * NW * u;
                bufs[(i >> 10) * FW_BUF + (i & 1023)] = pre[u];
            }
        }
        __syncthreads();
        if (PREFETCH && tix + gridDim.x < p.n_tiles) fetch(tile_src(tix + gridDim.x), pre);
        // ---- this wave's 1024-point transform, in its own buffer
        cf v[16];
#pragma unroll
        for (int a = 0; a < 16; ++a) v[a] = buf[64 * a + l];
        fw_wave_sync();
        radix16<DIR>(v);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            cf x = v[R16_POS(k)];
            if (k) x = tw_mul<DIR>(x, tw1[k * 64 + l]);
            buf[k * FW_S1 + l] = x;
        }
        fw_wave_sync();
#pragma unroll
        for (int bb = 0; bb < 16; ++bb) v[bb] = buf[q0 * FW_S1 + 4 * bb + q1];
        fw_wave_sync();
        radix16<DIR>(v);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            cf x = v[R16_POS(k)];
            if (k) x = tw_mul<DIR>(x, tw2[k * 4 + q1]);
            buf[q1 * FW_P + 17 * q0 + k] = x;
        }
        fw_wave_sync();
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) v[4 * j + c] = buf[c * FW_P + 17 * (q1 + 4 * j) + q0];
        fw_wave_sync();
        // R4 over c -> k2; X[k], k = (q1 + 4j) + 16 q0 + 256 k2, goes to position k + (k >> 4)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            radix4<DIR>(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) buf[q1 + 4 * j + 17 * q0 + 272 * k2] = v[4 * j + k2];
        }
        __syncthreads();
        // ---- coalesced tile store (optionally times the four-step twiddle); 32-bit offsets
        {
            const unsigned out_cs = static_cast<unsigned>(p.out_cs), out_ks = static_cast<unsigned>(p.out_ks);
            const unsigned col0 = static_cast<unsigned>(tl) * NW;
#pragma unroll 8
            for (int u = 0; u < 16; ++u) {
                unsigned c, k;
                if (p.out_c_fast) {
                    c = tid & (NW - 1);
                    k = (tid >> LOG) + 64 * u;
                } else {
                    const unsigned i = tid + 64 * NW * u;
                    c = i >> 10;
                    k = i & 1023;
                }
                cf x = bufs[c * FW_BUF + k + (k >> 4)];
                if (tw_split) {
                    x = tw_apply<DIR>(x, g_mul(tw_a, twu[u * 16 + c]));
                } else if (p.apply_tw) {
                    const unsigned e = (col0 + c) * k;  // < N <= 2^24
                    x = tw_apply<DIR>(x, g_mul(p.tw_hi[e >> 12], p.tw_lo[e & 4095]));
                }
                dst[c * out_cs + k * out_ks] = x;
            }
        }
    }
    kstamp_end(p.ks);
}

#ifdef COMMS_DIAG  // (diagnostic build only: a measured-no-faster trial, kept for the comparison it records)
// ---------------------------------------------------------------- the same tile, two LDS trips shorter (round 4 trial)
// fft1024x16_kernel<DIR, 16> with (1) the tile loaded straight into the transform's register layout -- rows: wave w
// reads transform w, point 64 a + lane into register a (512-B runs); columns: thread (c, t) = (tid & 15, tid >> 4)
// already holds rows t + 64 u of column c, which IS register u of lane t of column c's transform -- so the first
// radix-16 runs on the registers the loads land in and the tile's first write-to-LDS / read-back is gone (for columns
// the exchange that follows is then written across waves: one barrier where the tile-in barrier was; for rows it
// stays wave-local and the tile needs two barriers instead of three); (2) the last radix-4 across lanes
// (radix4_lanes: v_permlane32/16_swap) instead of through LDS.  Per thread and tile 32 + 32 LDS accesses instead of
// 64 + 64, 122-126 VGPRs and no spill (128 and 1-3 there).  Same tables, same buffers, same store stage, same
// results (all FFT parity tests pass on it).  And no faster: config 4 27.78 against 27.37 ms per step, N = 2^17 ... 2^22
// 0-2 % slower in two alternating pairs of runs (profiles/r04_fft_tile_direct.txt) -- these passes do not wait for
// LDS; what they wait for is the memory side.  Kept for that measurement, off by default
// (COMMS_FFT_TILE_DIRECT=1 selects it).
template <int DIR, bool TWS = false>
__global__ __launch_bounds__(1024, 4) void fft1024x16d_kernel(const cf* in, cf* out, FftTileParams p,
                                                               const cf* __restrict__ tw1g,
                                                               const cf* __restrict__ tw2g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = 16;
    cf* tw1 = reinterpret_cast<cf*>(smem);  // [16][64]  W1024^{lane*k0}
    cf* tw2 = tw1 + 1024;                   // [16][4]   W64^{c*k1}
    cf* bufs = tw2 + 64;                    // [16][FW_BUF]
    cf* twu = bufs + NW * FW_BUF;           // [16][16]  W_N^{64 C u} of the tile's columns (four-step store twiddle)
    const int tid = threadIdx.x;
    const int l = tid & 63, wave = tid >> 6;
    const int q0 = l & 15, q1 = l >> 4;
    cf* buf = bufs + wave * FW_BUF;
    kstamp_begin(p.ks);
    tw1[tid] = tw1g[tid];
    if (tid < 64) tw2[tid] = tw2g[tid];
    auto tile_src = [&](size_t tix) {
        const size_t b = tix / p.tiles_per_xform;
        return in + b * p.N + (tix - b * p.tiles_per_xform) * p.tile_step_in;
    };
    const unsigned in_cs = static_cast<unsigned>(p.in_cs), in_ls = static_cast<unsigned>(p.in_ls);
    const bool cols = p.in_c_fast != 0;
    // register u <- point 64 u + t of transform c: (c, t) = (tid & 15, tid >> 4) along columns, (wave, lane) along rows
    const unsigned o0 = cols ? (tid & 15) * in_cs + (tid >> 4) * in_ls : wave * in_cs + l * in_ls;
    const unsigned ostep = 64u * in_ls;
    auto fetch = [&](const cf* src, cf (&r)[16]) {
#pragma unroll
        for (int u = 0; u < 16; ++u) r[u] = src[o0 + u * ostep];
    };
    cf pre[16];
    if (blockIdx.x < p.n_tiles) fetch(tile_src(blockIdx.x), pre);
    // where this thread's exchange-1 values go: its transform's buffer, column t of the [k][66] image
    cf* x1 = cols ? bufs + (tid & 15) * FW_BUF + (tid >> 4) : buf + l;
    const cf* t1 = tw1 + (cols ? (tid >> 4) : l);

    for (size_t tix = blockIdx.x; tix < p.n_tiles; tix += gridDim.x) {
        const size_t b = tix / p.tiles_per_xform;
        const size_t tl = tix - b * p.tiles_per_xform;
        cf* dst = out + b * p.N + tl * p.tile_step_out;
        cf v[16];
#pragma unroll
        for (int a = 0; a < 16; ++a) v[a] = pre[a];
        // (unconditional: the last tile asks for itself again, so that `pre` is redefined on every path)
        fetch(tile_src(tix + gridDim.x < p.n_tiles ? tix + gridDim.x : tix), pre);
        __syncthreads();  // previous tile fully stored (and the tables are in place)
        cf tw_a = cf{1.f, 0.f};
        if (TWS) {  // four-step twiddle of the store stage, factored (see fft1024x16_kernel)
            const unsigned col0 = static_cast<unsigned>(tl) * NW;
            const unsigned ea = (col0 + (tid & 15)) * static_cast<unsigned>(tid >> 4);
            tw_a = g_mul(p.tw_hi[ea >> 12], p.tw_lo[ea & 4095]);
            if (tid < 256) {
                const unsigned eb = (col0 + (tid & 15)) * 64u * static_cast<unsigned>(tid >> 4);
                twu[tid] = g_mul(p.tw_hi[eb >> 12], p.tw_lo[eb & 4095]);
            }
        }
        radix16<DIR>(v);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            cf x = v[R16_POS(k)];
            if (k) x = tw_mul<DIR>(x, t1[k * 64]);
            x1[k * FW_S1] = x;
        }
        if (cols) __syncthreads();  // written by the threads that loaded them, read by the transform's own wave
        else fw_wave_sync();
#pragma unroll
        for (int bb = 0; bb < 16; ++bb) v[bb] = buf[q0 * FW_S1 + 4 * bb + q1];
        fw_wave_sync();
        radix16<DIR>(v);
        cf r[16], z[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            r[k] = v[R16_POS(k)];
            if (k) r[k] = tw_mul<DIR>(r[k], tw2[k * 4 + q1]);
        }
        radix4_lanes<DIR>(r, z);
        // z[4 t + m] of lane q0 + 16 g + 32 h is X[k], k = q0 + 16 (2 m + 8 g + h) + 256 t, at position k + (k >> 4)
        {
            cf* o = buf + q0 + 17 * (8 * (q1 & 1) + (q1 >> 1));
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int m = 0; m < 4; ++m) o[34 * m + 272 * t] = z[4 * t + m];
        }
        __syncthreads();
        // ---- coalesced tile store (optionally times the four-step twiddle); 32-bit offsets
        {
            const unsigned out_cs = static_cast<unsigned>(p.out_cs), out_ks = static_cast<unsigned>(p.out_ks);
            const unsigned col0 = static_cast<unsigned>(tl) * NW;
#pragma unroll 8
            for (int u = 0; u < 16; ++u) {
                unsigned c, k;
                if (p.out_c_fast) {
                    c = tid & (NW - 1);
                    k = (tid >> 4) + 64 * u;
                } else {
                    const unsigned i = tid + 64 * NW * u;
                    c = i >> 10;
                    k = i & 1023;
                }
                cf x = bufs[c * FW_BUF + k + (k >> 4)];
                if (TWS) {
                    x = tw_apply<DIR>(x, g_mul(tw_a, twu[u * 16 + c]));
                } else if (p.apply_tw) {
                    const unsigned e = (col0 + c) * k;  // < N <= 2^24
                    x = tw_apply<DIR>(x, g_mul(p.tw_hi[e >> 12], p.tw_lo[e & 4095]));
                }
                dst[c * out_cs + k * out_ks] = x;
            }
        }
    }
    kstamp_end(p.ks);
}
#endif  // COMMS_DIAG

// ---------------------------------------------------------------- N = RAD * 1024 in ONE pass (RAD = 1, 2, 4, 8, 16)
// n = 1024 n1 + n2, k = k1 + RAD k2:
//   X[k1 + RAD k2] = sum_{n2} W_1024^{n2 k2} * W_N^{n2 k1} * ( sum_{n1} x[1024 n1 + n2] W_RAD^{n1 k1} )
// A 16-wave workgroup owns 16 rows of 1024 points = 16/RAD transforms.  Lane n2 loads the RAD
// rows of a transform (coalesced), does the radix-RAD butterfly over n1 in registers, applies
// W_N^{n2 k1} (two small LDS tables: W_N^{64 wave k1} * W_N^{lane k1}) and drops row k1 into the
// LDS buffer of the wave that will transform it; every wave then runs the same barrier-free
// 1024-point transform as fft1024x16_kernel, and the tile goes back fully coalesced, because
// element i = k1 + RAD k2 (+ N j) of the tile is simply output i.  One HBM read and one write
// per point, where the tile kernel needs two passes above 4096 points.
template <int DIR>
__device__ __forceinline__ void radix2(cf& a, cf& b) {
    const cf t = cadd(a, b);
    b = csub(a, b);
    a = t;
}
// 8-point DFT, natural order in and out
template <int DIR>
__device__ __forceinline__ void radix8(cf (&v)[8]) {
    constexpr float R2 = 0.70710678118654752440f;
    radix4<DIR>(v[0], v[2], v[4], v[6]);  // E[0..3] in v[0], v[2], v[4], v[6]
    radix4<DIR>(v[1], v[3], v[5], v[7]);  // O[0..3] in v[1], v[3], v[5], v[7]
    const cf o1 = tw_mul<DIR>(v[3], cf{R2, -R2});
    const cf o3 = tw_mul<DIR>(v[7], cf{-R2, -R2});
    const cf e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1], o2 = v[5];
    v[0] = cadd(e0, o0);
    v[4] = csub(e0, o0);
    v[1] = cadd(e1, o1);
    v[5] = csub(e1, o1);
    v[2] = cadd_di<DIR>(e2, o2);  // W8^2 = -+i
    v[6] = csub_di<DIR>(e2, o2);
    v[3] = cadd(e3, o3);
    v[7] = csub(e3, o3);
}

// RAD = 0 stands for N = 64: the rows are 16 contiguous 64-point transforms each, which is the
// 1024-point wave transform with its first radix-16 stage (and twiddle) left out.
// RAD = 256 stands for N = 256: a row is 4 transforms, lane (t, b) does radix-16 over a
// (m = 16a + b), times W256^{b*ka}, one exchange, radix-16 over b: f = ka + 16 kb.
typedef float cf2v __attribute__((ext_vector_type(4)));  // two packed complex values (16 B)
// Bluestein riding on the single-pass kernel (padded length M = the kernel's N, a power of two):
//   mode 1 (forward transform of the pair): load x[b*n + m] * chirp[m] for m < n, zeros up to M;
//          store times bspec[m] (the spectrum of the wrapped conjugate chirp, 1/M folded in);
//   mode 2 (inverse transform): plain load; store only m < n, times chirp[m], compacted to n.
struct BluArgs {
    int mode;
    unsigned n, logM;
    const cf* chirp;
    const cf* bspec;
    // mode 3 (N = 1024 * the kernel's row length R * 1024, i.e. 2^21 ... 2^24): the COLUMN pass of the four-step split.  A tile
    // is XPT = 16 / R adjacent columns of the [R * 1024][1024] matrix, gathered in pieces of XPT * 8 B (the R tiles that
    // share 128-byte lines on one XCD in the same step); its spectra leave times W_N^{column * k} in runs of 16 XPT points,
    // laid out [k / 16][column][k % 16] so that the row pass reads whole contiguous tiles (16 adjacent k, all 1024 columns)
    const cf* tw_lo = nullptr;  // W_N^e, e < 4096
    const cf* tw_hi = nullptr;  // W_N^{4096 e}
};
// RAD = 128 / 512 stand for N = 2 x 64 / 2 x 256: a radix-2 butterfly over the two halves of a
// transform (x W_N^{n2}) on the way into LDS, then the 64- / 256-point form; lanes load both
// halves themselves (eight row pairs each), and the interleave k = k1 + 2 k2 is undone by the
// coalesced tile store.
constexpr int F256_T = 272;  // 256 + 16: the four transforms of a row start 32 banks apart
template <int RAD>
struct RxGeom {
    // RAD < 0 stands for the tiny lengths N = -RAD = 2 ... 32: a lane holds 16 consecutive points
    // (18-element rows in LDS) and transforms them in registers (C16N = 2, 4, 8 or 16 points at a
    // time; 32 = a radix-2 front stage + 16)
    static constexpr bool C64 = RAD == 0 || RAD == 128, C256 = RAD == 256 || RAD == 512, C16 = RAD < 0;
    static constexpr bool PRE2 = RAD == 128 || RAD == 512 || RAD == -32;
    static constexpr int C16N = !C16 ? 0 : (RAD == -32 ? 16 : -RAD);
    static constexpr int R = (C64 || C256 || C16) ? 1 : RAD;
    static constexpr int BUF = (C256 || C16) ? 1160 : 1088 + 32 / R;  // per-wave buffer stride: rows k1 land 32/RAD slots apart -> conflict-free tile reads
    static constexpr size_t LDS = (1024 + 64 + R * 16 + R * 64 + 16 * BUF) * sizeof(float2);
};

// The wave-level part shared by the row kernel (fft_rx1024_kernel) and the column kernel
// (fft_cols_kernel): the wave's 1024-point buffer holds one 1024-point transform (exchange
// layout of fft1024x16_kernel), 16 blocks of 64 points (C64: the same without the first radix-16
// stage; block a at a*66) or 4 transforms of 256 points (C256: at t*272); spectra are left in
// place in natural order (k + (k >> 4); a + 17 f; t*272 + f).
template <int DIR, bool C64, bool C256, int C16N = 0>
__device__ __forceinline__ void rx_wave_core(cf* buf, const cf* tw1, const cf* tw2, int l, int q0, int q1) {
    cf v[16];
    if constexpr (C16N != 0) {  // 16 consecutive points per lane: 16/C16N transforms, all in registers
        cf2v* r = reinterpret_cast<cf2v*>(buf + l * 18);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const cf2v x = r[j];
            v[2 * j] = cf{x.x, x.y};
            v[2 * j + 1] = cf{x.z, x.w};
        }
        if constexpr (C16N == 16) {
            radix16<DIR>(v);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const cf a = v[R16_POS(2 * j)], b = v[R16_POS(2 * j + 1)];
                r[j] = cf2v{a.x, a.y, b.x, b.y};
            }
        } else {
            if constexpr (C16N == 8) {
                cf lo[8], hi[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    lo[j] = v[j];
                    hi[j] = v[8 + j];
                }
                radix8<DIR>(lo);
                radix8<DIR>(hi);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    v[j] = lo[j];
                    v[8 + j] = hi[j];
                }
            }
            if constexpr (C16N == 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) radix4<DIR>(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
            }
            if constexpr (C16N == 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) radix2<DIR>(v[2 * j], v[2 * j + 1]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = cf2v{v[2 * j].x, v[2 * j].y, v[2 * j + 1].x, v[2 * j + 1].y};
        }
        return;
    }
    if constexpr (C256) {
        // this wave's row: four 256-point transforms, lane (t, b) = (q1, q0)
#pragma unroll
        for (int a = 0; a < 16; ++a) v[a] = buf[q1 * F256_T + 16 * a + q0];
        fw_wave_sync();
        radix16<DIR>(v);
#pragma unroll
        for (int k = 0; k < 16; ++k) {  // -> exchange row (t, ka), column b; 18-element rows
            cf x = v[R16_POS(k)];
            if (k) x = tw_mul<DIR>(x, tw1[k * 16 + q0]);
            buf[(q1 * 16 + k) * 18 + q0] = x;
        }
        fw_wave_sync();
        {  // lane (t, ka): its 16 consecutive b values
            const cf2v* r = reinterpret_cast<const cf2v*>(buf + l * 18);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const cf2v x = r[j];
                v[2 * j] = cf{x.x, x.y};
                v[2 * j + 1] = cf{x.z, x.w};
            }
        }
        fw_wave_sync();
        radix16<DIR>(v);
#pragma unroll
        for (int k = 0; k < 16; ++k) buf[q1 * F256_T + q0 + 16 * k] = v[R16_POS(k)];  // f = ka + 16 kb
    }
    if constexpr (!C64 && !C256) {
#pragma unroll
        for (int a = 0; a < 16; ++a) v[a] = buf[64 * a + l];
        fw_wave_sync();
        radix16<DIR>(v);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            cf x = v[R16_POS(k)];
            if (k) x = tw_mul<DIR>(x, tw1[k * 64 + l]);
            buf[k * FW_S1 + l] = x;
        }
        fw_wave_sync();
    }
    if constexpr (!C256) {
#pragma unroll
    for (int bb = 0; bb < 16; ++bb) v[bb] = buf[q0 * FW_S1 + 4 * bb + q1];
    fw_wave_sync();
    radix16<DIR>(v);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        cf x = v[R16_POS(k)];
        if (k) x = tw_mul<DIR>(x, tw2[k * 4 + q1]);
        buf[q1 * FW_P + 17 * q0 + k] = x;
    }
    fw_wave_sync();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) v[4 * j + c] = buf[c * FW_P + 17 * (q1 + 4 * j) + q0];
    fw_wave_sync();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        radix4<DIR>(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) buf[q1 + 4 * j + 17 * q0 + 272 * k2] = v[4 * j + k2];
    }
    }
}

// Buffer stride of the column form (BluArgs mode 3).  Its lanes run over the tile's columns first: stage 1 writes column j's row k
// to buffer j R + k, and the store reads 16 consecutive k of one column (rows k % R, 16 / R consecutive positions each).  An odd
// stride keeps the former apart, = 9 / 9 / 5 / 1 (mod 32 slots of 8 B) for R = 2 / 4 / 8 / 16 the latter.
constexpr int rx_cols_buf(int r) { return r == 8 ? 1093 : r == 16 ? 1089 : 1097; }
constexpr size_t rx_cols_lds(int r) { return static_cast<size_t>(1024 + 64 + r * 16 + r * 64 + 16 * rx_cols_buf(r)) * sizeof(float2); }

// PART: the launch covers a partly filled tile (whole transforms only): zeros in, nothing out past
// the end.  Full tiles run the unguarded instantiation.
template <int DIR, int RAD, int BLU = 0, bool PART = false>
__global__ __launch_bounds__(1024, 4) void fft_rx1024_kernel(const cf* in, cf* out, size_t n_tiles, size_t first_tile, size_t n_points,
                                                             const cf* __restrict__ tw1g, const cf* __restrict__ tw2g,
                                                             const cf* __restrict__ twag, const cf* __restrict__ twbg,
                                                             const BluArgs blu) {
    constexpr int R = RxGeom<RAD>::R, N = R * 1024, XPT = 16 / R, BUF = BLU == 3 ? rx_cols_buf(RxGeom<RAD>::R) : RxGeom<RAD>::BUF;
    constexpr bool C64 = RxGeom<RAD>::C64, C256 = RxGeom<RAD>::C256, C16 = RxGeom<RAD>::C16, PRE2 = RxGeom<RAD>::PRE2;
    constexpr bool C1024 = !C64 && !C256 && !C16;
    constexpr int C16N = RxGeom<RAD>::C16N, SLOTW = C256 ? F256_T : C64 ? FW_S1 : 18;  // slot stride of a half-spectrum (PRE2)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* tw1 = reinterpret_cast<cf*>(smem);  // [16][64]   W1024^{lane*k0}
    cf* tw2 = tw1 + 1024;                   // [16][4]    W64^{c*k1}
    cf* twa = tw2 + 64;                     // [RAD][16]  W_N^{64*wave*k1}
    cf* twb = twa + R * 16;                 // [RAD][64]  W_N^{lane*k1}
    cf* bufs = twb + R * 64;                // [16][BUF]
    const int tid = threadIdx.x;
    const int l = tid & 63, wave = tid >> 6;
    const int q0 = l & 15, q1 = l >> 4;
    cf* buf = bufs + wave * BUF;
    tw1[tid] = tw1g[tid];
    if (tid < 64) tw2[tid] = tw2g[tid];
    if (tid < R * 16) twa[tid] = twag[tid];
    if (tid < R * 64) twb[tid] = twbg[tid];

    // PRE2: lane (h, rest) handles rows 2 rp + h; its two halves sit HALF points apart at offset e2
    constexpr unsigned HALF = RAD == 512 ? 256u : RAD == -32 ? 16u : 64u;
    const unsigned h2 = static_cast<unsigned>(tid) >> 9, rest = static_cast<unsigned>(tid) & 511u;
    const unsigned n2 = rest & (HALF - 1), tr = rest / HALF;       // position in the half, transform of the row
    const unsigned e2 = 2 * HALF * tr + n2;
    cf w2 = cf{1.f, 0.f};
    if constexpr (PRE2) w2 = twbg[n2];  // W_N^{n2}
    cf pre[16];
    // the last tile of a batch may be partly past the end (whole transforms only): zeros in, nothing out
    auto fetch = [&](size_t tix) {
        const cf* src = in + tix * (16u * 1024u);
        const size_t left = n_points - tix * (16u * 1024u);  // > 0
        if constexpr (BLU == 1) {  // gather from the unpadded input, times the chirp
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const unsigned i = PRE2 ? 1024u * (2 * (u >> 1) + h2) + e2 + HALF * (u & 1) : 1024u * u + static_cast<unsigned>(tid);
                const size_t g = tix * (16u * 1024u) + i;
                const unsigned m = static_cast<unsigned>(g) & ((1u << blu.logM) - 1u);
                pre[u] = ((!PART || i < left) && m < blu.n) ? g_mul(in[(g >> blu.logM) * blu.n + m], blu.chirp[m]) : cf{0.f, 0.f};
            }
        } else if constexpr (BLU == 3) {  // lane (j = tid % XPT, tt = tid / XPT) holds column j, rows tt + (1024 / XPT) u
            constexpr unsigned TPX = 1024u / XPT;  // tiles per transform
            const cf* col = in + (tix / TPX) * (static_cast<size_t>(N) * 1024u) + (tix % TPX) * XPT;
            const unsigned o0 = (static_cast<unsigned>(tid) / XPT) * 1024u + static_cast<unsigned>(tid) % XPT;
#pragma unroll
            for (int u = 0; u < 16; ++u) pre[u] = col[o0 + static_cast<unsigned>(u) * (1024u / XPT) * 1024u];
        } else if constexpr (!PART) {
            if constexpr (PRE2) {
#pragma unroll
                for (int u = 0; u < 16; ++u) pre[u] = src[1024u * (2 * (u >> 1) + h2) + e2 + HALF * (u & 1)];  // u = 2*rp + n1
            } else {
#pragma unroll
                for (int u = 0; u < 16; ++u) pre[u] = src[1024u * u + static_cast<unsigned>(tid)];  // u = j*RAD + n1
            }
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const unsigned i = PRE2 ? 1024u * (2 * (u >> 1) + h2) + e2 + HALF * (u & 1) : 1024u * u + static_cast<unsigned>(tid);
                pre[u] = i < left ? src[i] : cf{0.f, 0.f};
            }
        }
    };
    // Tile of this workgroup's g-th step.  Plain launches: blockIdx.x + g * gridDim.x.  Column pass (mode 3) on the full grid
    // of 256 workgroups: the R tiles that share the 128-byte lines of sixteen adjacent columns go to R workgroups of ONE
    // XCD in the same step (workgroup b runs on XCD b % 8), so that a line fetched for one of them is in that XCD's L2 for
    // the others (32-byte pieces: 0.40 -> 0.30 ms per 2^26 points, scripts/probes/strided_tiles.hip).
    const bool grouped = BLU == 3 && gridDim.x == 256 && first_tile == 0;
    auto tile_of = [&](size_t g) -> size_t {
        if (!grouped) return first_tile + blockIdx.x + g * gridDim.x;
        const unsigned x = blockIdx.x & 7u, y = blockIdx.x >> 3;
        return static_cast<size_t>(R) * (x + 8u * (y / R) + (256u / R) * g) + (y % R);
    };
    const size_t n_steps = grouped ? (n_tiles + 255) / 256 : (n_tiles > first_tile + blockIdx.x ? (n_tiles - first_tile - blockIdx.x + gridDim.x - 1) / gridDim.x : 0);
    if (n_steps && tile_of(0) < n_tiles) fetch(tile_of(0));

    for (size_t g = 0; g < n_steps; ++g) {
        const size_t tix = tile_of(g);
        if (tix >= n_tiles) continue;  // (grouped, ragged last step; workgroup-uniform)
        __syncthreads();  // previous tile fully stored (and the tables are in place)
        // ---- radix-RAD over n1, times W_N^{n2*k1}, row k1 -> buffer j*RAD + k1, position n2 = tid
        if constexpr (PRE2) {  // radix-2 over the halves; half-spectrum k1 of transform tr -> slot 2 tr + k1 of row 2 rp + h
#pragma unroll
            for (int rp = 0; rp < 8; ++rp) {
                cf y0 = pre[2 * rp], y1 = pre[2 * rp + 1];
                radix2<DIR>(y0, y1);
                y1 = tw_mul<DIR>(y1, w2);
                cf* row = bufs + (2 * rp + h2) * BUF;
                row[(2 * tr) * SLOTW + n2] = y0;
                row[(2 * tr + 1) * SLOTW + n2] = y1;
            }
        }
        if constexpr (RAD == 0) {  // 64-point rows: block a = wave of row u goes to exchange-1 row a as it is
#pragma unroll
            for (int u = 0; u < 16; ++u) bufs[u * BUF + wave * FW_S1 + l] = pre[u];
        }
        if constexpr (C16 && !PRE2) {  // 16-point blocks of row u in 18-element rows
#pragma unroll
            for (int u = 0; u < 16; ++u) bufs[u * BUF + (tid >> 4) * 18 + (tid & 15)] = pre[u];
        }
        if constexpr (RAD == 256) {  // transform t = tid >> 8 of row u at t*272
#pragma unroll
            for (int u = 0; u < 16; ++u) bufs[u * BUF + (tid >> 8) * F256_T + (tid & 255)] = pre[u];
        }
#pragma unroll
        for (int m = 0; m < (C1024 && BLU == 3 ? XPT : 0); ++m) {  // column j = tid % XPT, rows n1 = 1024 a + t
            cf v[R];
#pragma unroll
            for (int a = 0; a < R; ++a) v[a] = pre[a * XPT + m];
            if constexpr (RAD == 2) radix2<DIR>(v[0], v[1]);
            if constexpr (RAD == 4) radix4<DIR>(v[0], v[1], v[2], v[3]);
            if constexpr (RAD == 8) radix8<DIR>(v);
            if constexpr (RAD == 16) radix16<DIR>(v);
            const unsigned t = static_cast<unsigned>(tid) / XPT + (1024u / XPT) * m, jc = static_cast<unsigned>(tid) % XPT;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                cf x = v[RAD == 16 ? R16_POS(k) : k];
                if (k) x = tw_mul<DIR>(x, cmulf(twa[k * 16 + (t >> 6)], twb[k * 64 + (t & 63u)]));
                bufs[(jc * R + k) * BUF + t] = x;
            }
        }
#pragma unroll
        for (int j = 0; j < (C1024 && BLU != 3 ? XPT : 0); ++j) {
            cf v[R];
#pragma unroll
            for (int a = 0; a < R; ++a) v[a] = pre[j * R + a];
            if constexpr (RAD == 2) radix2<DIR>(v[0], v[1]);  // (RAD == 1: N = 1024, the rows go straight to the waves)
            if constexpr (RAD == 4) radix4<DIR>(v[0], v[1], v[2], v[3]);
            if constexpr (RAD == 8) radix8<DIR>(v);
            if constexpr (RAD == 16) radix16<DIR>(v);
#pragma unroll
            for (int k = 0; k < R; ++k) {
                cf x = v[RAD == 16 ? R16_POS(k) : k];
                if (k) x = tw_mul<DIR>(x, cmulf(twa[k * 16 + wave], twb[k * 64 + l]));
                bufs[(j * R + k) * BUF + tid] = x;
            }
        }
        __syncthreads();
        if (g + 1 < n_steps && tile_of(g + 1) < n_tiles) fetch(tile_of(g + 1));
        // Four-step twiddle of the column pass's store, W_N^{c k} for this lane's column c and k = kk + (1024 / XPT) u:
        // W_N^{c kk} and the step W_N^{c 1024 / XPT}, looked up here, a whole transform ahead of their use
        cf cw0 = cf{1.f, 0.f}, cws = cf{1.f, 0.f};
        if constexpr (BLU == 3) {
            const unsigned t4 = static_cast<unsigned>(tid) >> 4;
            const unsigned c = static_cast<unsigned>(tix % (1024u / XPT)) * XPT + t4 % XPT;
            const unsigned e0 = c * (16u * (t4 / XPT) + (static_cast<unsigned>(tid) & 15u)), es = c * (1024u / XPT);  // < 2^24
            cw0 = g_mul(blu.tw_hi[e0 >> 12], blu.tw_lo[e0 & 4095]);
            cws = g_mul(blu.tw_hi[es >> 12], blu.tw_lo[es & 4095]);
        }
        // ---- this wave's 1024 points: one 1024-point transform, 16 of 64 points or 4 of 256 points
        rx_wave_core<DIR, C64, C256, C16N>(buf, tw1, tw2, l, q0, q1);
        __syncthreads();
        // ---- store: tile element i = j*N + RAD*k2 + k1 is output i
        cf* dst = out + tix * (16u * 1024u);
        const size_t left_out = n_points - tix * (16u * 1024u);
        if constexpr (BLU == 3) {
            // tile element (j, k) -> [k / 16][column][k % 16] of its transform: lane (k % 16 = tid & 15, j = (tid >> 4) % XPT)
            // writes k = kk + (1024 / XPT) u; the twiddle W_N^{c k} = cw0 * cws^u as a product of at most four of
            // cws, cws^2, cws^4, cws^8 (a chain of fifteen products would carry fifteen roundings)
            constexpr unsigned TPX = 1024u / XPT;
            const unsigned t4 = static_cast<unsigned>(tid) >> 4, js = t4 % XPT, kk = 16u * (t4 / XPT) + (static_cast<unsigned>(tid) & 15u);
            cf* xo_base = out + (tix / TPX) * (static_cast<size_t>(N) * 1024u) + static_cast<size_t>(kk >> 4) * 16384u + ((tix % TPX) * XPT + js) * 16u + (kk & 15u);
            const cf* lsrc = bufs + (js * R + kk % R) * BUF + kk / R + ((kk / R) >> 4);
            const cf p1 = cws, p2 = g_mul(p1, p1), p4 = g_mul(p2, p2), p8 = g_mul(p4, p4);
            cf l8 = cw0, l4 = cw0, l2 = cw0;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (u == 8) l8 = g_mul(cw0, p8);
                if ((u & 3) == 0) l4 = (u & 4) ? g_mul(l8, p4) : l8;
                if ((u & 1) == 0) l2 = (u & 2) ? g_mul(l4, p2) : l4;
                const cf w = (u & 1) ? g_mul(l2, p1) : l2;
                // k = kk + 64 R u: row k % R = kk % R, position k / R = kk / R + 64 u of it, padded by one slot per sixteen
                const cf x = lsrc[68 * u];
                xo_base[static_cast<size_t>(u) * (TPX / 16u) * 16384u] = tw_apply<DIR>(x, w);
            }
            continue;
        }
#pragma unroll 8
        for (int u = 0; u < 16; ++u) {
            const unsigned i = static_cast<unsigned>(tid) + 1024u * u;
            if (PART && i >= left_out) break;
            cf* dp = dst + i;
            unsigned mb = 0;  // index inside the padded transform (Bluestein modes)
            if constexpr (BLU != 0) {
                const size_t g = tix * (16u * 1024u) + i;
                mb = static_cast<unsigned>(g) & ((1u << blu.logM) - 1u);
                if constexpr (BLU == 2) {  // keep the first n of every M outputs, compacted
                    if (mb >= blu.n) continue;
                    dp = out + (g >> blu.logM) * blu.n + mb;
                }
            }
            cf xo;
            if constexpr (RAD == 0) {  // row u, block a = wave, frequency f = l sits at a + 17 f
                xo = bufs[u * BUF + wave + 17 * l];
            } else if constexpr (C16 && !PRE2) {
                xo = bufs[u * BUF + (tid >> 4) * 18 + (tid & 15)];
            } else if constexpr (RAD == -32) {  // transform tid >> 5, k = k1 + 2 k2: block 2 tr + k1, point k2
                const unsigned k = static_cast<unsigned>(tid) & 31u;
                xo = bufs[u * BUF + (2 * (static_cast<unsigned>(tid) >> 5) + (k & 1u)) * 18 + (k >> 1)];
            } else if constexpr (RAD == 256) {
                xo = bufs[u * BUF + (tid >> 8) * F256_T + (tid & 255)];
            } else if constexpr (RAD == 128) {  // transform tid >> 7, k = k1 + 2 k2: block 2 tr + k1, frequency k2
                const unsigned k = static_cast<unsigned>(tid) & 127u;
                xo = bufs[u * BUF + 2 * (static_cast<unsigned>(tid) >> 7) + (k & 1u) + 17 * (k >> 1)];
            } else if constexpr (RAD == 512) {  // transform tid >> 9: slot 2 tr + k1, frequency k2
                const unsigned k = static_cast<unsigned>(tid) & 511u;
                xo = bufs[u * BUF + (2 * (static_cast<unsigned>(tid) >> 9) + (k & 1u)) * F256_T + (k >> 1)];
            } else {
                const unsigned k1 = i % R, k2 = (i / R) & 1023u, j = i / N;
                xo = bufs[(j * R + k1) * BUF + k2 + (k2 >> 4)];
            }
            if constexpr (BLU == 1) xo = g_mul(xo, blu.bspec[mb]);
            if constexpr (BLU == 2) xo = g_mul(xo, blu.chirp[mb]);
            *dp = xo;
        }
    }
}

// ---------------------------------------------------------------- N = 32768 in ONE pass (round 4)
// 32768 = 32 x 1024 does not fit the tile (16384 points of LDS), but it fits the workgroup: a thread holds column
// n2 = tid of all 32 rows in registers (64 VGPRs), does the radix-32 over n1 there (a radix-2 split into two
// radix-16: Y[2q] = DFT16(x[r] + x[r+16]), Y[2q+1] = DFT16((x[r] - x[r+16]) W32^r)), and the 32 rows k1 then pass
// through the sixteen wave buffers in TWO phases of sixteen rows (k1 = 0 ... 15, then 16 ... 31): times W_N^{n2 k1}, into
// the buffer of wave k1 % 16, the barrier-free 1024-point wave transform, coalesced store of X[k1 + 32 k2] -- 128-byte
// pieces (sixteen adjacent k1) at a stride of 256 B, the other half of every 256 B following in the second phase.
// The twiddle W_N^{n2 k1} is a product of at most five of p1, p2, p4, p8, p16 = W_N^{n2 2^i}; n2 = tid is the same for
// every tile of the launch, so p1, p4, p16 are three table values per thread for the whole launch (p2, p8 one squaring
// each): no twiddle table in LDS at all.  Once the second phase's rows are in LDS all 64 data registers are free and
// the next transform's 32 rows are requested there, in two halves behind the second wave transform and its store.
// One HBM read and one write per point where the four-step form (fft_cols_kernel<32> + a row pass) moved 32 B/point
// and its 16-point column form spent half of its LDS cycles in bank conflicts (profiles/r03_lds_conflicts.txt).
constexpr int R32_BUF = 1090;  // 2180 dwords = 4 (mod 64 banks): the store's lanes (k1 % 16, k2) read conflict-free
constexpr size_t R32_LDS = (1024 + 64 + 16 * R32_BUF) * sizeof(float2);

template <int R>
__device__ __forceinline__ cf w32() {  // forward W32^R = (cos(2 pi R / 32), -sin(2 pi R / 32)), R = 1 ... 15
    constexpr float C[16] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                             0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f,
                             0.0f, -0.19509032201612826785f, -0.38268343236508977173f, -0.55557023301960222474f,
                             -0.70710678118654752440f, -0.83146961230254523708f, -0.92387953251128675613f, -0.98078528040323044913f};
    constexpr float S[16] = {0.0f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f,
                             0.70710678118654752440f, 0.83146961230254523708f, 0.92387953251128675613f, 0.98078528040323044913f,
                             1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                             0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f};
    return cf{C[R], -S[R]};
}
// W_N^{n2 K}: the factors of K's set bits, folded from the highest down (common prefixes are shared between the rows)
template <int K>
__device__ __forceinline__ cf r32_tw(const cf (&p)[5]) {
    static_assert(K >= 1 && K < 32, "row 1 ... 31");
    constexpr int HI = K >= 16 ? 4 : K >= 8 ? 3 : K >= 4 ? 2 : K >= 2 ? 1 : 0;
    constexpr int REST = K - (1 << HI);
    if constexpr (REST == 0) {
        return p[HI];
    } else {
        constexpr int LOW = (REST & -REST);                  // lowest set bit of the rest
        constexpr int LB = LOW == 1 ? 0 : LOW == 2 ? 1 : LOW == 4 ? 2 : 3;
        if constexpr (REST == LOW) return cmulf(p[HI], p[LB]);
        else return cmulf(r32_tw<K - LOW>(p), p[LB]);
    }
}

template <int DIR>
__global__ __launch_bounds__(1024, 4) void fft_rx32k_kernel(const cf* in, cf* out, size_t n_tiles, const cf* __restrict__ tw1g,
                                                            const cf* __restrict__ tw2g, const cf* __restrict__ twtg, KStamp ks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* tw1 = reinterpret_cast<cf*>(smem);  // [16][64]  W1024^{lane*k0}
    cf* tw2 = tw1 + 1024;                   // [16][4]   W64^{c*k1}
    cf* bufs = tw2 + 64;                    // [16][R32_BUF]
    const int tid = threadIdx.x;
    const int l = tid & 63, wave = tid >> 6;
    const int q0 = l & 15, q1 = l >> 4;
    cf* buf = bufs + wave * R32_BUF;
    kstamp_begin(ks);
    tw1[tid] = tw1g[tid];
    if (tid < 64) tw2[tid] = tw2g[tid];
    cf pw[5];  // W_N^{tid}, ^2, ^4, ^8, ^16 (forward sign; tw_mul<DIR> conjugates for the inverse)
    pw[0] = twtg[tid];
    pw[2] = twtg[1024 + tid];
    pw[4] = twtg[2048 + tid];
    pw[1] = cmulf(pw[0], pw[0]);
    pw[3] = cmulf(pw[2], pw[2]);

    cf pre[32];
    auto fetch = [&](size_t tix, int half) {  // rows of 8 KiB: one resource, one per-lane offset, the row in the scalar offset
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(in + tix * 32768u, 32768u * sizeof(cf));
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            const bv2u x = __builtin_amdgcn_raw_buffer_load_b64(rs, tid * 8, (16 * half + a) * 8192, 0);
            pre[16 * half + a] = cf{__uint_as_float(x.x), __uint_as_float(x.y)};
        }
    };
    // one phase: rows k1 = 16 P + j through the wave buffers and out
    auto store_phase = [&](size_t tix, int phase) {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(out + tix * 32768u, 32768u * sizeof(cf));
        const unsigned j = static_cast<unsigned>(tid) & 15u, kk = static_cast<unsigned>(tid) >> 4;
        const cf* src = bufs + j * R32_BUF;
        const unsigned voff = (j + 32u * kk) * 8u;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const unsigned k2 = kk + 64u * u;
            const cf x = src[k2 + (k2 >> 4)];
            __builtin_amdgcn_raw_buffer_store_b64(bv2u{__float_as_uint(x.x), __float_as_uint(x.y)}, rs, voff,
                                                  (16 * phase + 32 * 64 * u) * 8, 0);
        }
    };
    if (blockIdx.x < n_tiles) {
        fetch(blockIdx.x, 0);
        fetch(blockIdx.x, 1);
    }
    for (size_t tix = blockIdx.x; tix < n_tiles; tix += gridDim.x) {
        // ---- radix-32 over the rows: u -> even k1, w -> odd k1
        cf u[16], w[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            u[r] = cadd(pre[r], pre[r + 16]);
            w[r] = csub(pre[r], pre[r + 16]);
        }
#define COMMS_R32_TW(R) w[R] = tw_mul_s<DIR>(w[R], w32<R>());
        COMMS_R32_TW(1) COMMS_R32_TW(2) COMMS_R32_TW(3) COMMS_R32_TW(4) COMMS_R32_TW(5) COMMS_R32_TW(6) COMMS_R32_TW(7)
        COMMS_R32_TW(8) COMMS_R32_TW(9) COMMS_R32_TW(10) COMMS_R32_TW(11) COMMS_R32_TW(12) COMMS_R32_TW(13) COMMS_R32_TW(14)
        COMMS_R32_TW(15)
#undef COMMS_R32_TW
        radix16<DIR>(u);
        radix16<DIR>(w);
        __syncthreads();  // the previous transform's second store has read the buffers (and the tables are in place)
        // ---- phase 0: rows k1 = j = 2 q + p, q = 0 ... 7
#define COMMS_R32_ROW(K, J)                                                        \
    {                                                                              \
        cf y = ((K) & 1) ? w[R16_POS((K) >> 1)] : u[R16_POS((K) >> 1)];             \
        y = tw_mul<DIR>(y, r32_tw<K>(pw));                                         \
        bufs[(J) * R32_BUF + tid] = y;                                             \
    }
        bufs[tid] = u[R16_POS(0)];  // k1 = 0: no twiddle
        COMMS_R32_ROW(1, 1) COMMS_R32_ROW(2, 2) COMMS_R32_ROW(3, 3) COMMS_R32_ROW(4, 4) COMMS_R32_ROW(5, 5) COMMS_R32_ROW(6, 6)
        COMMS_R32_ROW(7, 7) COMMS_R32_ROW(8, 8) COMMS_R32_ROW(9, 9) COMMS_R32_ROW(10, 10) COMMS_R32_ROW(11, 11) COMMS_R32_ROW(12, 12)
        COMMS_R32_ROW(13, 13) COMMS_R32_ROW(14, 14) COMMS_R32_ROW(15, 15)
        __syncthreads();
        rx_wave_core<DIR, false, false, 0>(buf, tw1, tw2, l, q0, q1);
        __syncthreads();
        store_phase(tix, 0);
        __syncthreads();
        // ---- phase 1: rows k1 = 16 + j
        COMMS_R32_ROW(16, 0) COMMS_R32_ROW(17, 1) COMMS_R32_ROW(18, 2) COMMS_R32_ROW(19, 3) COMMS_R32_ROW(20, 4) COMMS_R32_ROW(21, 5)
        COMMS_R32_ROW(22, 6) COMMS_R32_ROW(23, 7) COMMS_R32_ROW(24, 8) COMMS_R32_ROW(25, 9) COMMS_R32_ROW(26, 10) COMMS_R32_ROW(27, 11)
        COMMS_R32_ROW(28, 12) COMMS_R32_ROW(29, 13) COMMS_R32_ROW(30, 14) COMMS_R32_ROW(31, 15)
#undef COMMS_R32_ROW
        __syncthreads();
        // all 64 data registers are free from here on: the next transform's rows are requested in two halves, the first
        // behind the wave transform (which needs ~60 registers itself), the second behind the store
        // (unconditional -- the last transform fetches itself again -- so that `pre` is redefined on every path: a
        // conditional fetch keeps the old values alive through the wave transform, and they spill)
        const size_t nxt = tix + gridDim.x < n_tiles ? tix + gridDim.x : tix;
        fetch(nxt, 0);
        rx_wave_core<DIR, false, false, 0>(buf, tw1, tw2, l, q0, q1);
        __syncthreads();
        fetch(nxt, 1);
        store_phase(tix, 1);
    }
    kstamp_end(ks);
}

// ---------------------------------------------------------------- four-step pass 1 for N = N1 * 1024, N1 = 64 ... 512
// Column transforms of length N1 (stride 1024) for all 1024 columns, times W_N^{n2*k1}, in place
// positions.  A 16-wave workgroup owns all N1 rows of C = 16384/N1 adjacent columns (runs of
// 256 B ... 2 KiB in HBM); each column lands in a slot of a wave's 1024-point buffer so that the
// waves can run the same 64- / 256-point forms as the row kernel (128 and 512 through the radix-2
// front stage: both halves of a column are in the same lane, 8 loads apart), and the spectra go
// back through the transposed mapping with the four-step twiddle applied on the way.
template <int KIND>
struct ColGeom {
    static constexpr bool C16 = KIND == 32, C64 = KIND == 0 || KIND == 128, C256 = KIND == 256 || KIND == 512;
    static constexpr bool PRE2 = KIND == 128 || KIND == 512 || KIND == 32;
    static constexpr int C16N = C16 ? 16 : 0;
    static constexpr int N1 = KIND == 0 ? 64 : KIND;
    static constexpr int C = 16384 / N1;    // columns per tile
    static constexpr int TPW = 1024 / N1;   // columns per wave buffer
    // Per-wave buffer stride.  Staging and the transposed store walk the lanes over the tile's columns, i.e. over buffers
    // (column / TPW) and slots (column % TPW) at one row: with a stride that is a multiple of 32 slots of 8 B (1120, and
    // 1160 = 8 mod 32 against the 256-point slots' 16 mod 32) the buffers of a half-wave fell on the same banks -- 70 % of
    // the LDS cycles of the 256- and 512-point forms were bank conflicts (SQ_LDS_BANK_CONFLICT).  An odd stride spreads
    // them; the 256-point core reads 16-byte pairs from its buffer and needs an even one (1154 = 2 mod 32), the 16-point form
    // keeps its stride.
    static constexpr int BUF = C16 ? 1160 : C256 ? 1154 : 1089;  // (the 256-point core uses 1150 slots of its buffer, the 64-point one 1087)
    static constexpr int SLOT = C64 ? FW_S1 : C256 ? F256_T : 18;
    static constexpr size_t LDS = (1024 + 64 + 256 + 16 * BUF) * sizeof(float2);
};

template <int DIR, int KIND>
__global__ __launch_bounds__(1024, 4) void fft_cols_kernel(const cf* in, cf* out, size_t n_tiles, unsigned N,
                                                           const cf* __restrict__ tw1g, const cf* __restrict__ tw2g,
                                                           const cf* __restrict__ twrg, const cf* __restrict__ tw_lo,
                                                           const cf* __restrict__ tw_hi, KStamp ks) {
    using G = ColGeom<KIND>;
    kstamp_begin(ks);
    constexpr int C = G::C, TPW = G::TPW, BUF = G::BUF, SLOT = G::SLOT, N1 = G::N1;
    constexpr bool C64 = G::C64, C256 = G::C256, PRE2 = G::PRE2;
    constexpr int C16N = G::C16N;
    constexpr unsigned TPX = 1024 / C;  // tiles per transform
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* tw1 = reinterpret_cast<cf*>(smem);  // the core's stage table (W256^{b*ka} for the 256-point form)
    cf* tw2 = tw1 + 1024;                   // [16][4] W64^{c*k1}
    cf* twr = tw2 + 64;                     // W_N1^{r}, r < N1/2 (radix-2 front stage)
    cf* bufs = twr + 256;                   // [16][BUF]
    const int tid = threadIdx.x;
    const int l = tid & 63, wave = tid >> 6;
    const int q0 = l & 15, q1 = l >> 4;
    cf* buf = bufs + wave * BUF;
    tw1[tid] = tw1g[tid];
    if (tid < 64) tw2[tid] = tw2g[tid];
    if (PRE2 && tid < N1 / 2) twr[tid] = twrg[tid];

    auto base_of = [&](size_t tix) { return (tix / TPX) * static_cast<size_t>(N) + (tix % TPX) * C; };
    cf pre[16];
    auto fetch = [&](size_t tix) {
        const cf* src = in + base_of(tix);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const unsigned e = static_cast<unsigned>(tid) + 1024u * u;
            pre[u] = src[(e / C) * 1024u + (e % C)];
        }
    };
    if (blockIdx.x < n_tiles) fetch(blockIdx.x);

    for (size_t tix = blockIdx.x; tix < n_tiles; tix += gridDim.x) {
        __syncthreads();  // previous tile fully stored (and the tables are in place)
        if constexpr (PRE2) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {  // rows r and r + N1/2 of column c: pre[u], pre[u + 8]
                const unsigned e = static_cast<unsigned>(tid) + 1024u * u;
                const unsigned r = e / C, c = e % C;
                cf y0 = pre[u], y1 = pre[u + 8];
                radix2<DIR>(y0, y1);
                y1 = tw_mul<DIR>(y1, twr[r]);
                cf* row = bufs + (c / TPW) * BUF + (2 * (c % TPW)) * SLOT + r;
                row[0] = y0;
                row[SLOT] = y1;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const unsigned e = static_cast<unsigned>(tid) + 1024u * u;
                const unsigned r = e / C, c = e % C;
                bufs[(c / TPW) * BUF + (c % TPW) * SLOT + r] = pre[u];
            }
        }
        __syncthreads();
        if (tix + gridDim.x < n_tiles) fetch(tix + gridDim.x);
        // Four-step twiddle of the store stage, W_N^{n2 k1} with n2 = col0 + (tid % C) (one column per thread) and
        // k1 = tid / C + S u, S = 1024 / C: three table look-ups per thread and tile -- W^{n2 (tid / C)}, the step W^{n2 S}
        // and the step of four W^{n2 4 S} -- issued here, a transform ahead of their use; the sixteen factors are then
        // stepped from them (at most six products deep: ~4e-7) instead of two dependent gathers per stored element
        // waited for one at a time.  (Seven look-ups kept in registers -- every fourth factor exact -- spilled.)
        const unsigned col0 = static_cast<unsigned>(tix % TPX) * C;
        cf tw_a0, tw_s1, tw_s4;
        auto tw_fetch = [&]() {
            constexpr unsigned S = 1024u / C;
            const unsigned n2 = col0 + (static_cast<unsigned>(tid) % C);
            auto look = [&](unsigned ee) { return g_mul(tw_hi[ee >> 12], tw_lo[ee & 4095]); };  // ee < N1 * 1024 <= 2^19
            tw_a0 = look(n2 * (static_cast<unsigned>(tid) / C));
            tw_s1 = look(n2 * S);
            tw_s4 = look(n2 * S * 4);
        };
        // (the 128- and 512-point forms have no six registers to spare across the transform: they fetch behind it --
        // three independent look-ups, one L2 round trip per tile)
        constexpr bool TW_EARLY = KIND != 128 && KIND != 512;
        if (TW_EARLY) tw_fetch();
        rx_wave_core<DIR, C64, C256, C16N>(buf, tw1, tw2, l, q0, q1);
        if (!TW_EARLY) tw_fetch();
        __syncthreads();
        // ---- store transposed back, times W_N^{n2*k1}
        cf* dst = out + base_of(tix);
        cf tw_g = tw_a0, tw_w = tw_a0;  // factor of u = 4 (u >> 2), and of u
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const unsigned e = static_cast<unsigned>(tid) + 1024u * u;
            const unsigned k1 = e / C, c = e % C, wb = c / TPW, t = c % TPW;
            unsigned pos;
            if constexpr (PRE2) {
                const unsigned kk = k1 & 1u, k2 = k1 >> 1;
                pos = C64 ? (2 * t + kk) + 17 * k2 : (2 * t + kk) * SLOT + k2;  // (256-point slots and 16-point blocks alike)
            } else {
                pos = C64 ? t + 17 * k1 : t * SLOT + k1;
            }
            cf x = bufs[wb * BUF + pos];
            if (u) tw_w = (u & 3) ? g_mul(tw_w, tw_s1) : (tw_g = g_mul(tw_g, tw_s4));
            x = tw_apply<DIR>(x, tw_w);
            dst[k1 * 1024u + c] = x;
        }
    }
    kstamp_end(ks);
}

// ---------------------------------------------------------------- N = 2^21 ... 2^24: third launch
// For N = 1024 * N2 (N2 = 2048 ... 16384) the four-step runs as columns (1024 points at stride N2, on
// fft1024x16_kernel with the twiddle) -> rows (N2 points, contiguous, in place, on the single-pass
// fft_rx1024_kernel) -> this transpose, X[k1 + 1024 k2] = Z[k1][k2]: a row pass with a transposed store would
// write 16 KiB / N2 = 8 ... 1 complex values per piece, so the transposition is its own coalesced pass
// (64 x 64 tiles through LDS, 16 B/point) -- 48 B/point in all, every launch on a kernel that runs near
// the rate its access pattern allows, instead of two passes of the generic radix-4 tile kernel.
constexpr int TR_T = 64;
__global__ __launch_bounds__(256) void fft_transpose_kernel(const cf* __restrict__ in, cf* __restrict__ out,
                                                            unsigned rows, unsigned cols, size_t n_tiles) {
    __shared__ cf tile[TR_T][TR_T + 1];
    const unsigned tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const unsigned tiles_c = cols / TR_T, tiles_r = rows / TR_T;
    const size_t per = static_cast<size_t>(tiles_c) * tiles_r, mat = static_cast<size_t>(rows) * cols;
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const size_t b = t / per;
        const unsigned tt = static_cast<unsigned>(t - b * per);
        const unsigned r0 = (tt / tiles_c) * TR_T, c0 = (tt % tiles_c) * TR_T;
        const cf* src = in + b * mat + static_cast<size_t>(r0) * cols + c0;
        cf* dst = out + b * mat + static_cast<size_t>(c0) * rows + r0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TR_T / 4; ++j) tile[ty + 4 * j][tx] = src[static_cast<size_t>(ty + 4 * j) * cols + tx];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TR_T / 4; ++j) dst[static_cast<size_t>(ty + 4 * j) * rows + tx] = tile[tx][ty + 4 * j];
    }
}

// Exact-index O(N^2) DFT, one transform per workgroup, f64 accumulation.
// Short lengths (N <= 128): a 256-lane workgroup takes G = 256 / N transforms at a time, one lane
// per output bin, so that e.g. a 10-point batch keeps 250 lanes busy instead of 10.
__global__ __launch_bounds__(256) void dft_small_kernel(const cf* __restrict__ in, cf* __restrict__ out, int N,
                                                        size_t batch, const cf* __restrict__ twN, int inverse) {
    __shared__ cf x[256];
    __shared__ cf w[128];
    const int G = 256 / N, tid = threadIdx.x;
    const int t = tid / N, k = tid - t * N;
    if (tid < N) {
        const cf tw = twN[tid];
        w[tid] = inverse ? cf{tw.x, -tw.y} : tw;
    }
    const size_t groups = (batch + G - 1) / G;
    for (size_t g = blockIdx.x; g < groups; g += gridDim.x) {
        const size_t b = g * G + t;
        const bool live = t < G && b < batch;
        __syncthreads();
        if (live) x[tid] = in[b * N + k];
        __syncthreads();
        if (live) {
            const cf* xt = x + t * N;
            double sr = 0.0, si = 0.0;
            int e = 0;
            for (int j = 0; j < N; ++j) {
                const double xr = xt[j].x, xi = xt[j].y, wr = w[e].x, wi = w[e].y;
                sr = fma(xr, wr, sr);
                sr = fma(-xi, wi, sr);
                si = fma(xr, wi, si);
                si = fma(xi, wr, si);
                e += k;
                if (e >= N) e -= N;
            }
            out[b * N + k] = cf{static_cast<float>(sr), static_cast<float>(si)};
        }
    }
}

__global__ __launch_bounds__(256) void dft_direct_kernel(const cf* __restrict__ in,
                                                         cf* __restrict__ out, int N,
                                                         size_t batch,
                                                         const cf* __restrict__ twN,
                                                         int inverse) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* x = reinterpret_cast<cf*>(smem);
    cf* w = x + N;
    for (int i = threadIdx.x; i < N; i += 256) {
        cf t = twN[i];
        w[i] = inverse ? cf{t.x, -t.y} : t;
    }
    for (size_t b = blockIdx.x; b < batch; b += gridDim.x) {
        __syncthreads();
        for (int i = threadIdx.x; i < N; i += 256) x[i] = in[b * N + i];
        __syncthreads();
        for (int k = threadIdx.x; k < N; k += 256) {
            double sr = 0.0, si = 0.0;
            int e = 0;
            for (int j = 0; j < N; ++j) {
                const double xr = x[j].x, xi = x[j].y, wr = w[e].x, wi = w[e].y;
                sr = fma(xr, wr, sr);
                sr = fma(-xi, wi, sr);
                si = fma(xr, wi, si);
                si = fma(xi, wr, si);
                e += k;
                if (e >= N) e -= N;
            }
            out[b * N + k] = cf{static_cast<float>(sr), static_cast<float>(si)};
        }
    }
}

// Bluestein helpers: a[n] = x[n] * chirp[n] zero-padded to M;  y[k] = c[k] * chirp[k]
__global__ void blu_pre_kernel(const cf* __restrict__ in, const cf* __restrict__ chirp,
                               cf* __restrict__ a, size_t N, size_t M, size_t batch) {
    const size_t total = batch * M;
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += stride) {
        const size_t b = i / M, n = i - b * M;
        a[i] = n < N ? g_mul(in[b * N + n], chirp[n]) : cf{0.f, 0.f};
    }
}
__global__ void blu_mul_kernel(cf* __restrict__ a, const cf* __restrict__ bspec, size_t M,
                               size_t batch) {
    const size_t total = batch * M;
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += stride)
        a[i] = g_mul(a[i], bspec[i % M]);
}
__global__ void blu_post_kernel(const cf* __restrict__ a, const cf* __restrict__ chirp,
                                cf* __restrict__ out, size_t N, size_t M, size_t batch) {
    const size_t total = batch * N;
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += stride) {
        const size_t b = i / N, k = i - b * N;
        out[i] = g_mul(a[b * M + k], chirp[k]);
    }
}

}  // namespace comms

using namespace comms;

static const double kPiF = 3.14159265358979323846264338327950288;

static int ilog2(size_t v) {
    int l = 0;
    while ((static_cast<size_t>(1) << l) < v) ++l;
    return l;
}

// Power-of-two plan: one tile pass (N <= 4096) or two (four-step).
struct Pow2Plan {
    size_t N = 0;
    int n_pass = 0;
    FftTileParams pass[2];
    float2* d_tw[4] = {nullptr, nullptr, nullptr, nullptr};  // twL(pass0), twL(pass1), tw_lo, tw_hi
    float2* d_fw1 = nullptr;  // fast 16x1024 kernel: W1024^{lane*k0} [16][64]
    float2* d_fw2 = nullptr;  //                       W64^{c*k1}     [16][4]
    int rx_rad = 0;           // N = rx_rad * 1024 (2, 4, 8, 16): single-pass fft_rx1024_kernel
    float2* d_rxa = nullptr;  //   W_N^{64*wave*k1} [rad][16]
    float2* d_rxb = nullptr;  //   W_N^{lane*k1}    [rad][64]
    float2* d_rx32 = nullptr;  // N = 32768 in one pass (fft_rx32k_kernel): W_N^{t}, W_N^{4 t}, W_N^{16 t}, t < 1024
    int col_kind = -1;         // four-step pass 1 on fft_cols_kernel: 0 (N1 = 64), 128, 256, 512; -1: tile kernel
    float2* d_colw1 = nullptr; //   the core's stage table (W256^{b*ka} for the 256-point form, else unused)
    float2* d_colr = nullptr;  //   W_N1^{r}, r < N1/2 (radix-2 front stage of 128 / 512)
    int threads[2] = {0, 0};
    size_t lds[2] = {0, 0};
    Pow2Plan* rows = nullptr;  // N = 2^21 ... 2^24: the plan of the N2-point row transforms (columns: pass[0]; then the transpose)

    void release() {
        if (rows) {
            rows->release();
            delete rows;
            rows = nullptr;
        }
        for (auto& p : d_tw)
            if (p) {
                (void)hipFree(p);
                p = nullptr;
            }
        if (d_fw1) (void)hipFree(d_fw1);
        if (d_fw2) (void)hipFree(d_fw2);
        if (d_rxa) (void)hipFree(d_rxa);
        if (d_rxb) (void)hipFree(d_rxb);
        if (d_colw1) (void)hipFree(d_colw1);
        if (d_colr) (void)hipFree(d_colr);
        if (d_rx32) (void)hipFree(d_rx32);
        d_fw1 = d_fw2 = d_rxa = d_rxb = d_colw1 = d_colr = d_rx32 = nullptr;
    }
    bool fast(int i) const { return pass[i].L == 1024 && (pass[i].C == 16 || pass[i].C == 8) && d_fw1 != nullptr; }
};

static comms_status_t upload_tw(size_t count, size_t denom, size_t mult, float2** d_out) {
    // table[m] = exp(-2 pi i * (m * mult) / denom), m < count   (forward sign)
    std::vector<float2> t(count);
    for (size_t m = 0; m < count; ++m) {
        size_t e = (m * mult) % denom;
        double a = -2.0 * kPiF * static_cast<double>(e) / static_cast<double>(denom);
        t[m] = make_float2(static_cast<float>(std::cos(a)), static_cast<float>(std::sin(a)));
    }
    COMMS_HIP_TRY(hipMalloc(d_out, count * sizeof(float2)));
    COMMS_HIP_TRY(hipMemcpy(*d_out, t.data(), count * sizeof(float2), hipMemcpyHostToDevice));
    return COMMS_OK;
}

static void tile_geometry(FftTileParams& p, int L, size_t want_c) {
    p.L = L;
    p.logL = ilog2(L);
    size_t C = FT_MAX_POINTS / L;
    if (C > want_c) C = want_c;
    if (C < 1) C = 1;
    p.C = static_cast<int>(C);
    p.logC = ilog2(C);
}

static comms_status_t pow2_plan_build(Pow2Plan& pl, size_t N) {
    pl.N = N;
    const int logN = ilog2(N);
    COMMS_ARG(logN <= 24, "power-of-two FFT supports up to 2^24 points (got 2^%d)", logN);
    if (N <= 4096) {
        pl.n_pass = 1;
        FftTileParams& p = pl.pass[0];
        memset(&p, 0, sizeof(p));
        // 1024-point batches: two 8-wave workgroups per CU overlap load/compute/store (measured
        // 258 vs 226 Gpoints/s); four-step passes keep 16 columns = 128-B row pieces
        // short transforms: full 16384-point tiles (1024 lanes, 128 KiB of LDS) -- with 16 transforms
        // per tile a 64-point batch ran one wave per workgroup (74 -> 173 Gpoints/s)
        static const size_t tile_pts = static_cast<size_t>(diag_knob("COMMS_FFT_TILE_PTS", 16384));
        tile_geometry(p, static_cast<int>(N), N == 1024 ? 8 : (N < 1024 ? tile_pts / N : 16));
        // rows mode: tile = C consecutive transforms; the "transform" seen by the
        // kernel is the tile itself (distance C*N), one tile per transform
        p.in_c_fast = 0;
        p.out_c_fast = 0;
        p.in_cs = N;
        p.in_ls = 1;
        p.out_cs = N;
        p.out_ks = 1;
        p.tiles_per_xform = 1;
        p.tile_step_in = p.tile_step_out = 0;
        p.N = static_cast<size_t>(p.C) * N;
        COMMS_TRY(upload_tw(N, N, 1, &pl.d_tw[0]));
        p.twL = reinterpret_cast<const cf*>(pl.d_tw[0]);
    } else if (logN > 20) {
        // 2^21 .. 2^24: columns of 1024 points (stride N2) -> rows of N2 = N / 1024 points -> transpose
        pl.n_pass = 2;  // (pass[1] is unused; the row transforms have a plan of their own)
        const size_t N2 = N >> 10;
        FftTileParams& a = pl.pass[0];
        memset(&a, 0, sizeof(a));
        memset(&pl.pass[1], 0, sizeof(pl.pass[1]));
        tile_geometry(a, 1024, 16);
        a.in_c_fast = 1;
        a.out_c_fast = 1;
        a.in_cs = 1;
        a.in_ls = N2;
        a.out_cs = 1;
        a.out_ks = N2;
        a.tiles_per_xform = N2 / a.C;
        a.tile_step_in = a.tile_step_out = a.C;
        a.N = N;
        a.apply_tw = 1;
        COMMS_TRY(upload_tw(1024, 1024, 1, &pl.d_tw[0]));
        COMMS_TRY(upload_tw(4096, N, 1, &pl.d_tw[2]));
        COMMS_TRY(upload_tw(N / 4096, N, 4096, &pl.d_tw[3]));
        a.twL = reinterpret_cast<const cf*>(pl.d_tw[0]);
        a.tw_lo = reinterpret_cast<const cf*>(pl.d_tw[2]);
        a.tw_hi = reinterpret_cast<const cf*>(pl.d_tw[3]);
        // row pass of the two-pass form: tile t = the 16 adjacent outputs k of every column, as the column pass left them
        // ([k / 16][column][k % 16]); 1024-point transforms over the columns; X[k + N2 k2] in 128-byte pieces
        FftTileParams& b = pl.pass[1];
        tile_geometry(b, 1024, 16);
        b.in_c_fast = 1;
        b.out_c_fast = 1;
        b.in_cs = 1;
        b.in_ls = 16;
        b.out_cs = 1;
        b.out_ks = N2;
        b.tiles_per_xform = N2 / 16;
        b.tile_step_in = 16384;
        b.tile_step_out = 16;
        b.N = N;
        b.twL = a.twL;
        pl.rows = new (std::nothrow) Pow2Plan;
        COMMS_ARG(pl.rows != nullptr, "out of host memory");
        COMMS_TRY(pow2_plan_build(*pl.rows, N2));
    } else {
        pl.n_pass = 2;
        // 2^15 .. 2^20: N2 = 1024 so that pass 2 (and for 2^20 pass 1 too) runs on fft1024x16_kernel;
        // the short column pass takes wide tiles (2-KiB row pieces at N1 = 64)
        const int log1 = (logN >= 15 && logN <= 20) ? logN - 10 : logN / 2, log2v = logN - log1;
        const size_t N1 = static_cast<size_t>(1) << log1, N2 = static_cast<size_t>(1) << log2v;
        // pass 1: columns n2, FFT over n1 (stride N2), twiddle, in place
        FftTileParams& a = pl.pass[0];
        memset(&a, 0, sizeof(a));
        tile_geometry(a, static_cast<int>(N1), N1 < 1024 ? (FT_MAX_POINTS / N1 < N2 ? FT_MAX_POINTS / N1 : N2) : 16);
        a.in_c_fast = 1;
        a.out_c_fast = 1;
        a.in_cs = 1;
        a.in_ls = N2;
        a.out_cs = 1;
        a.out_ks = N2;
        a.tiles_per_xform = N2 / a.C;
        a.tile_step_in = a.tile_step_out = a.C;
        a.N = N;
        a.apply_tw = 1;
        COMMS_TRY(upload_tw(N1, N1, 1, &pl.d_tw[0]));
        COMMS_TRY(upload_tw(4096, N, 1, &pl.d_tw[2]));
        COMMS_TRY(upload_tw(N / 4096, N, 4096, &pl.d_tw[3]));
        a.twL = reinterpret_cast<const cf*>(pl.d_tw[0]);
        a.tw_lo = reinterpret_cast<const cf*>(pl.d_tw[2]);
        a.tw_hi = reinterpret_cast<const cf*>(pl.d_tw[3]);
        // pass 2: rows k1, FFT over n2, transposed store -> X[k1 + N1*k2]
        FftTileParams& b = pl.pass[1];
        memset(&b, 0, sizeof(b));
        tile_geometry(b, static_cast<int>(N2), 16);
        b.in_c_fast = 0;
        b.out_c_fast = 1;
        b.in_cs = N2;
        b.in_ls = 1;
        b.out_cs = 1;
        b.out_ks = N1;
        b.tiles_per_xform = N1 / b.C;
        b.tile_step_in = static_cast<size_t>(b.C) * N2;
        b.tile_step_out = b.C;
        b.N = N;
        COMMS_TRY(upload_tw(N2, N2, 1, &pl.d_tw[1]));
        b.twL = reinterpret_cast<const cf*>(pl.d_tw[1]);
    }
    for (int i = 0; i < pl.n_pass; ++i) {
        const int npts = pl.pass[i].L * pl.pass[i].C;
        int T = npts / FT_PTS;
        if (T < 64) T = 64;
        if (T > 1024) T = 1024;
        pl.threads[i] = T;
        pl.lds[i] = static_cast<size_t>(npts) * sizeof(float2);
    }
    const bool rx = (N >= 2 && N <= 32) || N == 64 || N == 128 || N == 256 || N == 512 || N == 1024 || N == 2048 || N == 4096 || N == 8192 || N == 16384;
    if (pl.pass[0].L == 1024 || (pl.n_pass == 2 && pl.pass[1].L == 1024) || rx) {
        std::vector<float2> t1(1024), t2(64);
        for (int k0 = 0; k0 < 16; ++k0)
            for (int t = 0; t < 64; ++t) {
                const double a = -2.0 * kPiF * static_cast<double>((t * k0) % 1024) / 1024.0;
                t1[k0 * 64 + t] = make_float2(static_cast<float>(std::cos(a)), static_cast<float>(std::sin(a)));
            }
        for (int k1 = 0; k1 < 16; ++k1)
            for (int c = 0; c < 4; ++c) {
                const double a = -2.0 * kPiF * static_cast<double>((c * k1) % 64) / 64.0;
                t2[k1 * 4 + c] = make_float2(static_cast<float>(std::cos(a)), static_cast<float>(std::sin(a)));
            }
        COMMS_HIP_TRY(hipMalloc(&pl.d_fw1, t1.size() * sizeof(float2)));
        COMMS_HIP_TRY(hipMalloc(&pl.d_fw2, t2.size() * sizeof(float2)));
        COMMS_HIP_TRY(hipMemcpy(pl.d_fw1, t1.data(), t1.size() * sizeof(float2), hipMemcpyHostToDevice));
        COMMS_HIP_TRY(hipMemcpy(pl.d_fw2, t2.data(), t2.size() * sizeof(float2), hipMemcpyHostToDevice));
        const int fw_lds = (1024 + 64 + 16 * FW_BUF + 256) * static_cast<int>(sizeof(float2));
        COMMS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fft1024x16_kernel<1, 16>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, fw_lds));
        COMMS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fft1024x16_kernel<-1, 16>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, fw_lds));
        COMMS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fft1024x16_kernel<1, 16, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, fw_lds));
        COMMS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fft1024x16_kernel<-1, 16, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, fw_lds));
#ifdef COMMS_DIAG
        COMMS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fft1024x16d_kernel<1>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, fw_lds));
        COMMS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fft1024x16d_kernel<-1>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, fw_lds));
        COMMS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fft1024x16d_kernel<1, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, fw_lds));
        COMMS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fft1024x16d_kernel<-1, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, fw_lds));
#endif
        COMMS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fft1024x16_kernel<1, 8>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, fw_lds));
        COMMS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fft1024x16_kernel<-1, 8>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, fw_lds));
    }
    if (rx) {
        const int rad = N < 1024 ? 1 : static_cast<int>(N / 1024);  // (N = 64, 256: tables unused, kept for the common signature)
        std::vector<float2> ta(rad * 16), tb(rad * 64);
        for (int k = 0; k < rad; ++k) {
            for (int w = 0; w < 16; ++w) {
                const double a = -2.0 * kPiF * static_cast<double>((64ull * w * k) % N) / static_cast<double>(N);
                ta[k * 16 + w] = make_float2(static_cast<float>(std::cos(a)), static_cast<float>(std::sin(a)));
            }
            for (int l = 0; l < 64; ++l) {
                const double a = -2.0 * kPiF * static_cast<double>((static_cast<size_t>(l) * k) % N) / static_cast<double>(N);
                tb[k * 64 + l] = make_float2(static_cast<float>(std::cos(a)), static_cast<float>(std::sin(a)));
            }
        }
        COMMS_HIP_TRY(hipMalloc(&pl.d_rxa, ta.size() * sizeof(float2)));
        COMMS_HIP_TRY(hipMalloc(&pl.d_rxb, tb.size() * sizeof(float2)));
        COMMS_HIP_TRY(hipMemcpy(pl.d_rxa, ta.data(), ta.size() * sizeof(float2), hipMemcpyHostToDevice));
        COMMS_HIP_TRY(hipMemcpy(pl.d_rxb, tb.data(), tb.size() * sizeof(float2), hipMemcpyHostToDevice));
        pl.rx_rad = N == 64 ? -1 : N == 256 ? -2 : N == 128 ? -3 : N == 512 ? -4 : N <= 32 ? -static_cast<int>(N) - 100 : rad;  // < 0: the short forms (-100 - N: the tiny ones)
        if (N == 128 || N == 512 || N == 32) {  // W_N^{n2}, n2 < N/2, for the radix-2 front stage
            std::vector<float2> t(N / 2 < 64 ? 64 : N / 2, make_float2(1.f, 0.f));
            for (size_t j = 0; j < N / 2; ++j) {
                const double a = -2.0 * kPiF * static_cast<double>(j) / static_cast<double>(N);
                t[j] = make_float2(static_cast<float>(std::cos(a)), static_cast<float>(std::sin(a)));
            }
            (void)hipFree(pl.d_rxb);
            pl.d_rxb = nullptr;
            COMMS_HIP_TRY(hipMalloc(&pl.d_rxb, t.size() * sizeof(float2)));
            COMMS_HIP_TRY(hipMemcpy(pl.d_rxb, t.data(), t.size() * sizeof(float2), hipMemcpyHostToDevice));
        }
        if (N == 256 || N == 512) {  // its stage twiddle W256^{b*ka} at [ka*16 + b] takes the place of the W1024 table
            std::vector<float2> t1(1024, make_float2(1.f, 0.f));
            for (int k = 0; k < 16; ++k)
                for (int bq = 0; bq < 16; ++bq) {
                    const double a = -2.0 * kPiF * static_cast<double>((bq * k) % 256) / 256.0;
                    t1[k * 16 + bq] = make_float2(static_cast<float>(std::cos(a)), static_cast<float>(std::sin(a)));
                }
            COMMS_HIP_TRY(hipMemcpy(pl.d_fw1, t1.data(), t1.size() * sizeof(float2), hipMemcpyHostToDevice));
        }
    }
    if (pl.n_pass == 2 && pl.pass[1].L == 1024 && pl.d_fw1 &&
        (pl.pass[0].L == 64 || pl.pass[0].L == 128 || pl.pass[0].L == 256 || pl.pass[0].L == 512)) {
        // (N1 = 32, i.e. N = 32768, runs in one pass on fft_rx32k_kernel since round 4; its 16-point column form --
        // half of its LDS cycles bank conflicts -- is retired, COMMS_FFT_NO_RX32K falls back to the generic tile kernel)
        const int n1 = pl.pass[0].L;
        std::vector<float2> t1(1024, make_float2(1.f, 0.f)), tr(256, make_float2(1.f, 0.f));
        for (int k = 0; k < 16; ++k)
            for (int bq = 0; bq < 16; ++bq) {
                const double a = -2.0 * kPiF * static_cast<double>((bq * k) % 256) / 256.0;
                t1[k * 16 + bq] = make_float2(static_cast<float>(std::cos(a)), static_cast<float>(std::sin(a)));
            }
        for (int r = 0; r < n1 / 2; ++r) {
            const double a = -2.0 * kPiF * static_cast<double>(r) / static_cast<double>(n1);
            tr[r] = make_float2(static_cast<float>(std::cos(a)), static_cast<float>(std::sin(a)));
        }
        COMMS_HIP_TRY(hipMalloc(&pl.d_colw1, t1.size() * sizeof(float2)));
        COMMS_HIP_TRY(hipMalloc(&pl.d_colr, tr.size() * sizeof(float2)));
        COMMS_HIP_TRY(hipMemcpy(pl.d_colw1, t1.data(), t1.size() * sizeof(float2), hipMemcpyHostToDevice));
        COMMS_HIP_TRY(hipMemcpy(pl.d_colr, tr.data(), tr.size() * sizeof(float2), hipMemcpyHostToDevice));
        pl.col_kind = n1 == 64 ? 0 : n1;
    }
    if (N == 32768 && pl.d_fw1) {  // the single-pass form: three twiddle values per thread
        std::vector<float2> t(3 * 1024);
        const size_t mult[3] = {1, 4, 16};
        for (int i = 0; i < 3; ++i)
            for (size_t n2 = 0; n2 < 1024; ++n2) {
                const double a = -2.0 * kPiF * static_cast<double>((n2 * mult[i]) % N) / static_cast<double>(N);
                t[i * 1024 + n2] = make_float2(static_cast<float>(std::cos(a)), static_cast<float>(std::sin(a)));
            }
        COMMS_HIP_TRY(hipMalloc(&pl.d_rx32, t.size() * sizeof(float2)));
        COMMS_HIP_TRY(hipMemcpy(pl.d_rx32, t.data(), t.size() * sizeof(float2), hipMemcpyHostToDevice));
        COMMS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fft_rx32k_kernel<1>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(R32_LDS)));
        COMMS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fft_rx32k_kernel<-1>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(R32_LDS)));
    }
    // tiles above 64 KiB need the dynamic-LDS limit raised (160 KiB per CU on gfx950)
    COMMS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fft_tile_kernel<1>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize,
                                      FT_MAX_POINTS * sizeof(float2)));
    COMMS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fft_tile_kernel<-1>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize,
                                      FT_MAX_POINTS * sizeof(float2)));
    return COMMS_OK;
}

// Runs `batch` transforms of length pl.N.  In-place (in == out) is fine: every
// tile is fully read before it is written and tiles do not overlap; pass 2
// reads what pass 1 wrote to `out`.
#ifdef COMMS_DIAG
static bool fft_tile_direct() {  // trial kernel fft1024x16d_kernel (see there): off unless COMMS_FFT_TILE_DIRECT=1
    static const bool v = diag_knob("COMMS_FFT_TILE_DIRECT", 0) != 0;
    return v;
}
#endif
static comms_status_t launch_fast(Pow2Plan& pl, const float2* src, float2* dst, const FftTileParams& p,
                                  bool inverse, hipStream_t s) {
    const cf* t1 = reinterpret_cast<const cf*>(pl.d_fw1);
    const cf* t2 = reinterpret_cast<const cf*>(pl.d_fw2);
    const cf* a = reinterpret_cast<const cf*>(src);
    cf* d = reinterpret_cast<cf*>(dst);
#ifdef COMMS_DIAG
    if (p.C == 16 && fft_tile_direct()) {  // trial: register-layout loads, last radix-4 across lanes
        const size_t lds = (1024 + 64 + 16 * FW_BUF + 256) * sizeof(float2);
        const unsigned blocks = static_cast<unsigned>(p.n_tiles < static_cast<size_t>(kNumCU) ? p.n_tiles : kNumCU);
        if (p.apply_tw && p.out_c_fast) {
            if (inverse)
                fft1024x16d_kernel<1, true><<<dim3(blocks), dim3(1024), lds, s>>>(a, d, p, t1, t2);
            else
                fft1024x16d_kernel<-1, true><<<dim3(blocks), dim3(1024), lds, s>>>(a, d, p, t1, t2);
        } else if (inverse)
            fft1024x16d_kernel<1><<<dim3(blocks), dim3(1024), lds, s>>>(a, d, p, t1, t2);
        else
            fft1024x16d_kernel<-1><<<dim3(blocks), dim3(1024), lds, s>>>(a, d, p, t1, t2);
        return launch_ok("fft1024x16d_kernel");
    }
#endif
    if (p.C == 16) {  // one 16-wave workgroup per CU
        const size_t lds = (1024 + 64 + 16 * FW_BUF + 256) * sizeof(float2);
        const unsigned blocks = static_cast<unsigned>(p.n_tiles < static_cast<size_t>(kNumCU) ? p.n_tiles : kNumCU);
        if (p.apply_tw && p.out_c_fast) {
            if (inverse)
                fft1024x16_kernel<1, 16, true><<<dim3(blocks), dim3(1024), lds, s>>>(a, d, p, t1, t2);
            else
                fft1024x16_kernel<-1, 16, true><<<dim3(blocks), dim3(1024), lds, s>>>(a, d, p, t1, t2);
        } else if (inverse)
            fft1024x16_kernel<1, 16><<<dim3(blocks), dim3(1024), lds, s>>>(a, d, p, t1, t2);
        else
            fft1024x16_kernel<-1, 16><<<dim3(blocks), dim3(1024), lds, s>>>(a, d, p, t1, t2);
    } else {  // two 8-wave workgroups per CU: one loads/stores while the other computes
        const size_t lds = (1024 + 64 + 8 * FW_BUF + 256) * sizeof(float2);
        const size_t slots = 2 * static_cast<size_t>(kNumCU);
        const unsigned blocks = static_cast<unsigned>(p.n_tiles < slots ? p.n_tiles : slots);
        if (inverse)
            fft1024x16_kernel<1, 8><<<dim3(blocks), dim3(512), lds, s>>>(a, d, p, t1, t2);
        else
            fft1024x16_kernel<-1, 8><<<dim3(blocks), dim3(512), lds, s>>>(a, d, p, t1, t2);
    }
    return launch_ok("fft1024x16_kernel");
}

template <int RAD>
static comms_status_t launch_rx(Pow2Plan& pl, const float2* src, float2* dst, size_t n_points, bool inverse,
                                hipStream_t s, const BluArgs& blu = BluArgs{0, 0, 0, nullptr, nullptr}) {
    const size_t n_full = n_points / 16384, rem = n_points % 16384;
    const cf* a = reinterpret_cast<const cf*>(src);
    cf* d = reinterpret_cast<cf*>(dst);
    const cf* t1 = reinterpret_cast<const cf*>(pl.d_fw1);
    const cf* t2 = reinterpret_cast<const cf*>(pl.d_fw2);
    const cf* ta = reinterpret_cast<const cf*>(pl.d_rxa);
    const cf* tb = reinterpret_cast<const cf*>(pl.d_rxb);
    constexpr size_t lds = RxGeom<RAD>::LDS;
    static DeviceOnce attr_once;
    if (attr_once.need()) {
#define COMMS_RX_ATTR(...)                                                                             \
    COMMS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fft_rx1024_kernel<__VA_ARGS__>), \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)))
        COMMS_RX_ATTR(1, RAD);
        COMMS_RX_ATTR(-1, RAD);
        COMMS_RX_ATTR(-1, RAD, 1);
        COMMS_RX_ATTR(1, RAD, 2);
        COMMS_RX_ATTR(1, RAD, 0, true);
        COMMS_RX_ATTR(-1, RAD, 0, true);
        COMMS_RX_ATTR(-1, RAD, 1, true);
        COMMS_RX_ATTR(1, RAD, 2, true);
#undef COMMS_RX_ATTR
    }
    // full tiles on the persistent grid; a partly filled last tile as one more workgroup.  In the
    // Bluestein modes the two launches address in / out by the global element index, so the second
    // one simply starts at tile n_full.
    auto go = [&](auto kern_full, auto kern_part) {
        if (n_full) {
            const unsigned blocks = static_cast<unsigned>(n_full < static_cast<size_t>(kNumCU) ? n_full : kNumCU);
            kern_full<<<dim3(blocks), dim3(1024), lds, s>>>(a, d, n_full, 0, n_full * 16384, t1, t2, ta, tb, blu);
        }
        if (rem) kern_part<<<dim3(1), dim3(1024), lds, s>>>(a, d, n_full + 1, n_full, n_points, t1, t2, ta, tb, blu);
    };
    if (blu.mode == 1)  // Bluestein, first half: always the forward transform
        go(fft_rx1024_kernel<-1, RAD, 1>, fft_rx1024_kernel<-1, RAD, 1, true>);
    else if (blu.mode == 2)  // second half: always the inverse
        go(fft_rx1024_kernel<1, RAD, 2>, fft_rx1024_kernel<1, RAD, 2, true>);
    else if (inverse)
        go(fft_rx1024_kernel<1, RAD>, fft_rx1024_kernel<1, RAD, 0, true>);
    else
        go(fft_rx1024_kernel<-1, RAD>, fft_rx1024_kernel<-1, RAD, 0, true>);
    return launch_ok("fft_rx1024_kernel");
}

// Column pass of N = 2^21 ... 2^24 (BluArgs mode 3) on the row plan's tables: `rows` is the plan of the N / 1024-point transforms.
template <int RAD>
static comms_status_t launch_rx_cols(Pow2Plan& rows, const float2* src, float2* dst, size_t n_points, bool inverse, hipStream_t s,
                                     const cf* tw_lo, const cf* tw_hi) {
    constexpr size_t lds = rx_cols_lds(RAD);
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        COMMS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fft_rx1024_kernel<1, RAD, 3>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        COMMS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fft_rx1024_kernel<-1, RAD, 3>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    }
    BluArgs blu{3, 0, 0, nullptr, nullptr, tw_lo, tw_hi};
    const size_t n_tiles = n_points / 16384;  // whole transforms of at least 2^21 points: no partial tile
    const unsigned blocks = static_cast<unsigned>(n_tiles < static_cast<size_t>(kNumCU) ? n_tiles : kNumCU);
    const cf* a = reinterpret_cast<const cf*>(src);
    cf* d = reinterpret_cast<cf*>(dst);
    const cf* t1 = reinterpret_cast<const cf*>(rows.d_fw1);
    const cf* t2 = reinterpret_cast<const cf*>(rows.d_fw2);
    const cf* ta = reinterpret_cast<const cf*>(rows.d_rxa);
    const cf* tb = reinterpret_cast<const cf*>(rows.d_rxb);
    if (inverse)
        fft_rx1024_kernel<1, RAD, 3><<<dim3(blocks), dim3(1024), lds, s>>>(a, d, n_tiles, 0, n_points, t1, t2, ta, tb, blu);
    else
        fft_rx1024_kernel<-1, RAD, 3><<<dim3(blocks), dim3(1024), lds, s>>>(a, d, n_tiles, 0, n_points, t1, t2, ta, tb, blu);
    return launch_ok("fft_rx1024_kernel (columns)");
}

template <int KIND>
static comms_status_t launch_cols(Pow2Plan& pl, const float2* src, float2* dst, size_t batch, bool inverse,
                                  hipStream_t s) {
    constexpr size_t lds = ColGeom<KIND>::LDS;
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        COMMS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fft_cols_kernel<1, KIND>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        COMMS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fft_cols_kernel<-1, KIND>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    }
    const size_t n_tiles = batch * (1024 / ColGeom<KIND>::C);
    const unsigned blocks = static_cast<unsigned>(n_tiles < static_cast<size_t>(kNumCU) ? n_tiles : kNumCU);
    const cf* a = reinterpret_cast<const cf*>(src);
    cf* d = reinterpret_cast<cf*>(dst);
    const cf* t1 = reinterpret_cast<const cf*>(pl.d_colw1);
    const cf* t2 = reinterpret_cast<const cf*>(pl.d_fw2);
    const cf* tr = reinterpret_cast<const cf*>(pl.d_colr);
    const FftTileParams& p = pl.pass[0];
    if (inverse)
        fft_cols_kernel<1, KIND><<<dim3(blocks), dim3(1024), lds, s>>>(a, d, n_tiles, static_cast<unsigned>(pl.N), t1, t2, tr, p.tw_lo, p.tw_hi, p.ks);
    else
        fft_cols_kernel<-1, KIND><<<dim3(blocks), dim3(1024), lds, s>>>(a, d, n_tiles, static_cast<unsigned>(pl.N), t1, t2, tr, p.tw_lo, p.tw_hi, p.ks);
    return launch_ok("fft_cols_kernel");
}

// All of a batch on the single-pass kernel (plain, or as one half of a Bluestein pair).
static comms_status_t run_rx(Pow2Plan& pl, const float2* in, float2* out, size_t n_points, bool inverse, hipStream_t s,
                             const BluArgs& blu = BluArgs{0, 0, 0, nullptr, nullptr}) {
    switch (pl.rx_rad) {
            case -1: COMMS_TRY(launch_rx<0>(pl, in, out, n_points, inverse, s, blu)); break;
            case -2: COMMS_TRY(launch_rx<256>(pl, in, out, n_points, inverse, s, blu)); break;
            case -3: COMMS_TRY(launch_rx<128>(pl, in, out, n_points, inverse, s, blu)); break;
            case -4: COMMS_TRY(launch_rx<512>(pl, in, out, n_points, inverse, s, blu)); break;
            case -102: COMMS_TRY(launch_rx<-2>(pl, in, out, n_points, inverse, s, blu)); break;
            case -104: COMMS_TRY(launch_rx<-4>(pl, in, out, n_points, inverse, s, blu)); break;
            case -108: COMMS_TRY(launch_rx<-8>(pl, in, out, n_points, inverse, s, blu)); break;
            case -116: COMMS_TRY(launch_rx<-16>(pl, in, out, n_points, inverse, s, blu)); break;
            case -132: COMMS_TRY(launch_rx<-32>(pl, in, out, n_points, inverse, s, blu)); break;
            case 1: COMMS_TRY(launch_rx<1>(pl, in, out, n_points, inverse, s, blu)); break;
            case 2: COMMS_TRY(launch_rx<2>(pl, in, out, n_points, inverse, s, blu)); break;
            case 4: COMMS_TRY(launch_rx<4>(pl, in, out, n_points, inverse, s, blu)); break;
            case 8: COMMS_TRY(launch_rx<8>(pl, in, out, n_points, inverse, s, blu)); break;
            default: COMMS_TRY(launch_rx<16>(pl, in, out, n_points, inverse, s, blu)); break;
    }
    return COMMS_OK;
}

static comms_status_t pow2_run(Pow2Plan& pl, const float2* in, float2* out, size_t batch,
                               bool inverse, hipStream_t s, float2* scratch, KStamp ks = KStamp{nullptr, nullptr}) {
    // only the four-step passes of N = 2^15 ... 2^20 stamp (both of them: the call from its first workgroup in to its last
    // wave out); the column pass of the larger lengths does not, so those stay unstamped altogether
    pl.pass[0].ks = pl.pass[1].ks = pl.rows ? KStamp{nullptr, nullptr} : ks;
    static const bool no_rx = diag_knob("COMMS_FFT_NO_RX", 0) != 0;
    if (pl.rx_rad && !no_rx) return run_rx(pl, in, out, batch * pl.N, inverse, s);
    static const bool no_rx32k = diag_knob("COMMS_FFT_NO_RX32K", 0) != 0;
    if (pl.d_rx32 && !no_rx32k) {  // N = 32768: one pass, a transform per workgroup and step
        const unsigned blocks = static_cast<unsigned>(batch < static_cast<size_t>(kNumCU) ? batch : kNumCU);
        const cf* t1 = reinterpret_cast<const cf*>(pl.d_fw1);
        const cf* t2 = reinterpret_cast<const cf*>(pl.d_fw2);
        const cf* tt = reinterpret_cast<const cf*>(pl.d_rx32);
        if (inverse)
            fft_rx32k_kernel<1><<<dim3(blocks), dim3(1024), R32_LDS, s>>>(reinterpret_cast<const cf*>(in), reinterpret_cast<cf*>(out), batch, t1, t2, tt, ks);
        else
            fft_rx32k_kernel<-1><<<dim3(blocks), dim3(1024), R32_LDS, s>>>(reinterpret_cast<const cf*>(in), reinterpret_cast<cf*>(out), batch, t1, t2, tt, ks);
        return launch_ok("fft_rx32k_kernel");
    }
    if (pl.rows) {
        // N = 2^21 ... 2^24 in two passes: N / 1024-point columns gathered in pieces of 64 ... 8 B (small pieces cost far less
        // on the read side than on the write side: scripts/probes/strided_tiles.hip), spectra out in runs of 1 KiB ... 128 B;
        // then 1024-point rows, sixteen adjacent ones per tile, stored transposed in 128-byte pieces.
        static const int gather_max = diag_knob("COMMS_FFT_LARGE_GATHER", 23);
        if (ilog2(pl.N) <= gather_max && pl.rows->rx_rad >= 2) {
            const cf* lo = pl.pass[0].tw_lo;
            const cf* hi = pl.pass[0].tw_hi;
            switch (pl.rows->rx_rad) {
                case 2: COMMS_TRY(launch_rx_cols<2>(*pl.rows, in, scratch, batch * pl.N, inverse, s, lo, hi)); break;
                case 4: COMMS_TRY(launch_rx_cols<4>(*pl.rows, in, scratch, batch * pl.N, inverse, s, lo, hi)); break;
                case 8: COMMS_TRY(launch_rx_cols<8>(*pl.rows, in, scratch, batch * pl.N, inverse, s, lo, hi)); break;
                default: COMMS_TRY(launch_rx_cols<16>(*pl.rows, in, scratch, batch * pl.N, inverse, s, lo, hi)); break;
            }
            FftTileParams q = pl.pass[1];
            q.n_tiles = batch * q.tiles_per_xform;
            return launch_fast(pl, scratch, out, q, inverse, s);
        }
        // N = 2^24 (8-byte pieces): three launches on 128-byte pieces throughout -- 1024-point columns (in -> scratch), rows (in
        // place), transpose (scratch -> out) -- are faster (0.78 against 0.82 ms per 2^26 points)
        FftTileParams p = pl.pass[0];
        p.n_tiles = batch * p.tiles_per_xform;
        COMMS_TRY(launch_fast(pl, in, scratch, p, inverse, s));
        COMMS_TRY(run_rx(*pl.rows, scratch, scratch, batch * pl.N, inverse, s));
        const unsigned cols = static_cast<unsigned>(pl.N >> 10);
        const size_t n_tiles = batch * (1024 / TR_T) * (cols / TR_T);
        const size_t slots = static_cast<size_t>(16) * kNumCU;
        const unsigned blocks = static_cast<unsigned>(n_tiles < slots ? n_tiles : slots);
        fft_transpose_kernel<<<dim3(blocks), dim3(256), 0, s>>>(reinterpret_cast<const cf*>(scratch), reinterpret_cast<cf*>(out), 1024u,
                                                               cols, n_tiles);
        return launch_ok("fft_transpose_kernel");
    }
    for (int i = 0; i < pl.n_pass; ++i) {
        FftTileParams p = pl.pass[i];
        const float2* src = in;
        float2* dst = out;
        if (pl.n_pass == 1) {
            // group C transforms per tile; a ragged tail runs with C = 1 tiles
            const size_t full = batch / p.C;
            p.n_tiles = full;
            if (full && pl.fast(i)) {
                COMMS_TRY(launch_fast(pl, src, dst, p, inverse, s));
            } else if (full) {
                static const unsigned wgs = static_cast<unsigned>(diag_knob("COMMS_FFT_TILE_WGS", 4));
                unsigned blocks = static_cast<unsigned>(full < wgs * kNumCU ? full : wgs * kNumCU);
                if (inverse)
                    fft_tile_kernel<1><<<dim3(blocks), dim3(pl.threads[i]), pl.lds[i], s>>>(reinterpret_cast<const cf*>(src), reinterpret_cast<cf*>(dst), p);
                else
                    fft_tile_kernel<-1><<<dim3(blocks), dim3(pl.threads[i]), pl.lds[i], s>>>(reinterpret_cast<const cf*>(src), reinterpret_cast<cf*>(dst), p);
                COMMS_TRY(launch_ok("fft_tile_kernel"));
            }
            const size_t rem = batch - full * p.C;
            if (rem) {
                FftTileParams q = p;
                q.C = 1;
                q.logC = 0;
                q.N = pl.N;
                q.n_tiles = rem;
                const size_t off = full * p.C * pl.N;
                int T = static_cast<int>(pl.N) / FT_PTS;
                if (T < 64) T = 64;
                unsigned blocks = static_cast<unsigned>(rem < 4u * kNumCU ? rem : 4u * kNumCU);
                if (inverse)
                    fft_tile_kernel<1><<<dim3(blocks), dim3(T), pl.N * sizeof(float2), s>>>(reinterpret_cast<const cf*>(src + off), reinterpret_cast<cf*>(dst + off), q);
                else
                    fft_tile_kernel<-1><<<dim3(blocks), dim3(T), pl.N * sizeof(float2), s>>>(reinterpret_cast<const cf*>(src + off), reinterpret_cast<cf*>(dst + off), q);
                COMMS_TRY(launch_ok("fft_tile_kernel"));
            }
        } else {
            // pass 1: in -> scratch (same positions, twiddled); pass 2: scratch -> out
            // (transposed).  in == out is therefore fine.
            if (i == 0) {
                dst = scratch;
            } else {
                src = scratch;
            }
            p.n_tiles = batch * p.tiles_per_xform;
            static const bool no_cols = diag_knob("COMMS_FFT_NO_COLS", 0) != 0;
            if (i == 0 && pl.col_kind >= 0 && !no_cols) {
                switch (pl.col_kind) {
                    case 0: COMMS_TRY(launch_cols<0>(pl, src, dst, batch, inverse, s)); break;
                    case 128: COMMS_TRY(launch_cols<128>(pl, src, dst, batch, inverse, s)); break;
                    case 256: COMMS_TRY(launch_cols<256>(pl, src, dst, batch, inverse, s)); break;
                    default: COMMS_TRY(launch_cols<512>(pl, src, dst, batch, inverse, s)); break;
                }
                continue;
            }
            if (pl.fast(i)) {
                COMMS_TRY(launch_fast(pl, src, dst, p, inverse, s));
                continue;
            }
            unsigned blocks = static_cast<unsigned>(p.n_tiles < 4u * kNumCU ? p.n_tiles : 4u * kNumCU);
            if (inverse)
                fft_tile_kernel<1><<<dim3(blocks), dim3(pl.threads[i]), pl.lds[i], s>>>(reinterpret_cast<const cf*>(src), reinterpret_cast<cf*>(dst), p);
            else
                fft_tile_kernel<-1><<<dim3(blocks), dim3(pl.threads[i]), pl.lds[i], s>>>(reinterpret_cast<const cf*>(src), reinterpret_cast<cf*>(dst), p);
            COMMS_TRY(launch_ok("fft_tile_kernel"));
        }
    }
    return COMMS_OK;
}

struct comms_fft : Handle {
    size_t N = 0;
    bool inverse = false;
    int kind = 0;  // 0 pow2, 1 direct O(N^2), 2 Bluestein
    Pow2Plan plan;           // pow2: length N;  Bluestein: length M
    float2* d_twN = nullptr;     // direct: W_N^m
    size_t M = 0;                // Bluestein padded length
    float2* d_chirp = nullptr;   // exp(-/+ i pi n^2 / N)
    float2* d_bspec = nullptr;   // FFT_M(conj chirp, wrapped) / M
    Scratch work;                // four-step transpose buffer / Bluestein work
    Scratch work2;
};

static void free_fft(comms_fft* h) {
    (void)use_device(h->device);
    h->plan.release();
    if (h->d_twN) (void)hipFree(h->d_twN);
    if (h->d_chirp) (void)hipFree(h->d_chirp);
    if (h->d_bspec) (void)hipFree(h->d_bspec);
    h->work.release();
    h->work2.release();
    h->fini();
    delete h;
}

static comms_status_t fft_setup(comms_fft* h) {
    const size_t N = h->N;
    if ((N & (N - 1)) == 0) {
        h->kind = 0;
        if (N >= 2) COMMS_TRY(pow2_plan_build(h->plan, N));
        return COMMS_OK;
    }
    // short odd lengths: the exact-index O(N^2) DFT (f64 accumulation); above that Bluestein on the
    // power-of-two kernels is faster by far (N = 1000: 2.3 -> 20 Gpoints/s) -- measured crossover ~64
    static const size_t direct_max = static_cast<size_t>(diag_knob("COMMS_FFT_DIRECT_MAX", 64));
    if (N <= direct_max && N <= 4096) {
        h->kind = 1;
        return upload_tw(N, N, 1, &h->d_twN);
    }
    // Bluestein: with chirp[n] = W^{n^2/2} (W = e^{-/+ 2 pi i/N} by direction),
    //   X[k] = chirp[k] * sum_n (x[n] chirp[n]) * conj(chirp)[k-n]
    // i.e. one circular convolution of length M >= 2N-1 on the power-of-two path.
    h->kind = 2;
    COMMS_ARG(N <= (static_cast<size_t>(1) << 23), "FFT length %zu too large", N);
    size_t M = 1;
    while (M < 2 * N - 1) M <<= 1;
    h->M = M;
    COMMS_TRY(pow2_plan_build(h->plan, M));
    const double sgn = h->inverse ? 1.0 : -1.0;
    std::vector<float2> chirp(N);
    std::vector<double> cr(N), ci(N);
    for (size_t n = 0; n < N; ++n) {
        const size_t e = (n * n) % (2 * N);  // n^2 mod 2N keeps the angle exact
        const double a = sgn * kPiF * static_cast<double>(e) / static_cast<double>(N);
        cr[n] = std::cos(a);
        ci[n] = std::sin(a);
        chirp[n] = make_float2(static_cast<float>(cr[n]), static_cast<float>(ci[n]));
    }
    // b[m] = conj(chirp)[|m|] wrapped to length M; its spectrum via the device plan
    std::vector<float2> bvec(M, make_float2(0.f, 0.f));
    for (size_t n = 0; n < N; ++n) {
        float2 v = make_float2(static_cast<float>(cr[n]), static_cast<float>(-ci[n]));
        bvec[n] = v;
        if (n) bvec[M - n] = v;
    }
    COMMS_HIP_TRY(hipMalloc(&h->d_chirp, N * sizeof(float2)));
    COMMS_HIP_TRY(hipMemcpy(h->d_chirp, chirp.data(), N * sizeof(float2), hipMemcpyHostToDevice));
    COMMS_HIP_TRY(hipMalloc(&h->d_bspec, M * sizeof(float2)));
    COMMS_HIP_TRY(hipMemcpy(h->d_bspec, bvec.data(), M * sizeof(float2), hipMemcpyHostToDevice));
    COMMS_TRY(h->work.reserve(M * sizeof(float2)));
    COMMS_TRY(pow2_run(h->plan, h->d_bspec, h->d_bspec, 1, false, h->stream, static_cast<float2*>(h->work.p)));
    COMMS_HIP_TRY(hipStreamSynchronize(h->stream));
    // fold the 1/M of the inverse transform into the spectrum
    std::vector<float2> spec(M);
    COMMS_HIP_TRY(hipMemcpy(spec.data(), h->d_bspec, M * sizeof(float2), hipMemcpyDeviceToHost));
    const float inv = 1.0f / static_cast<float>(M);
    for (auto& v : spec) {
        v.x *= inv;
        v.y *= inv;
    }
    COMMS_HIP_TRY(hipMemcpy(h->d_bspec, spec.data(), M * sizeof(float2), hipMemcpyHostToDevice));
    return COMMS_OK;
}

extern "C" {

comms_status_t comms_fft_create(size_t fft_size, int32_t inverse, int32_t device,
                                comms_fft_t** out) {
    COMMS_ARG(out != nullptr, "out is NULL");
    *out = nullptr;
    COMMS_ARG(fft_size >= 1, "fft_size must be >= 1");
    comms_fft* h = new (std::nothrow) comms_fft;
    COMMS_ARG(h != nullptr, "out of host memory");
    comms_status_t st = h->init(device);
    if (st != COMMS_OK) {
        delete h;
        return st;
    }
    h->N = fft_size;
    h->inverse = inverse != 0;
    st = fft_setup(h);
    if (st != COMMS_OK) {
        free_fft(h);
        return st;
    }
    *out = h;
    return COMMS_OK;
}

comms_status_t comms_fft_run_dev(comms_fft_t* h, const comms_c32* d_in, size_t n,
                                 comms_c32* d_out, void* stream) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG(n > 0 && n % h->N == 0,
              "input length %zu is not a multiple of fft_size %zu (the reference panics)", n, h->N);
    COMMS_ARG(d_in && d_out, "NULL device pointer");
    COMMS_ARG(d_in == d_out || !ranges_overlap(d_in, n * 8, d_out, n * 8), "input and output overlap without being the same buffer");
    COMMS_TRY(use_device(h->device));
    hipStream_t s = nullptr;
    COMMS_TRY(h->enter(stream, &s));
    const float2* in = reinterpret_cast<const float2*>(d_in);
    float2* o = reinterpret_cast<float2*>(d_out);
    const size_t batch = n / h->N;
    if (h->N == 1) {
        if (d_in != d_out) COMMS_HIP_TRY(hipMemcpyAsync(o, in, n * sizeof(float2), hipMemcpyDefault, s));
        return COMMS_OK;
    }
    if (h->kind == 0) {
        if (h->plan.n_pass == 1) {
            h->tic(s);
            comms_status_t st = pow2_run(h->plan, in, o, batch, h->inverse, s, nullptr);
            h->toc(s);
            return st;
        }
        // four-step: pass 1 -> scratch -> pass 2, in groups that bound the scratch buffer.
        // (Small groups, sized so the intermediate stays in the 256 MiB Infinity Cache, were
        // measured and lose: 2^20 x 64 runs 0.56 ms ungrouped, 0.66 ms at 64 MiB, 2.3 ms at 8 MiB.)
        static const size_t group_bytes = static_cast<size_t>(diag_knob("COMMS_FFT_GROUP_MB", 2048)) << 20;
        size_t group = group_bytes / (h->N * sizeof(float2));
        if (group < 1) group = 1;
        if (group > batch) group = batch;
        COMMS_TRY(h->work.reserve(group * h->N * sizeof(float2)));
        float2* scratch = static_cast<float2*>(h->work.p);
        h->tic(s);
        const KStamp ks = h->next_stamp();
        for (size_t b0 = 0; b0 < batch; b0 += group) {
            const size_t nb = batch - b0 < group ? batch - b0 : group;
            COMMS_TRY(pow2_run(h->plan, in + b0 * h->N, o + b0 * h->N, nb, h->inverse, s, scratch, ks));
        }
        h->toc(s);
        return COMMS_OK;
    }
    if (h->kind == 1) {
        COMMS_ARG(!ranges_overlap(d_in, n * 8, d_out, n * 8) , "this fft_size cannot run in place");
        if (h->N <= 128) {
            const size_t groups = (batch + 256 / h->N - 1) / (256 / h->N);
            const unsigned gb = static_cast<unsigned>(groups < 8u * kNumCU ? groups : 8u * kNumCU);
            h->tic(s);
            dft_small_kernel<<<dim3(gb), dim3(256), 0, s>>>(reinterpret_cast<const cf*>(in), reinterpret_cast<cf*>(o),
                                                            static_cast<int>(h->N), batch,
                                                            reinterpret_cast<const cf*>(h->d_twN), h->inverse ? 1 : 0);
            h->toc(s);
            return launch_ok("dft_small_kernel");
        }
        unsigned blocks = static_cast<unsigned>(batch < 4u * kNumCU ? batch : 4u * kNumCU);
        h->tic(s);
        dft_direct_kernel<<<dim3(blocks), dim3(256), 2 * h->N * sizeof(float2), s>>>(
            reinterpret_cast<const cf*>(in), reinterpret_cast<cf*>(o), static_cast<int>(h->N), batch,
            reinterpret_cast<const cf*>(h->d_twN), h->inverse ? 1 : 0);
        h->toc(s);
        return launch_ok("dft_direct_kernel");
    }
    // Bluestein, in chunks that bound the work buffer
    const size_t M = h->M;
    size_t chunk = (static_cast<size_t>(1) << 24) / M;
    if (chunk < 1) chunk = 1;
    if (chunk > batch) chunk = batch;
    COMMS_TRY(h->work.reserve(chunk * M * sizeof(float2)));
    if (h->plan.n_pass == 2) COMMS_TRY(h->work2.reserve(chunk * M * sizeof(float2)));
    float2* a = static_cast<float2*>(h->work.p);
    float2* sc = static_cast<float2*>(h->work2.p);
    static const bool no_fuse = diag_knob("COMMS_FFT_BLU_UNFUSED", 0) != 0;
    const bool fuse = h->plan.rx_rad != 0 && !no_fuse;  // padded length on the single-pass kernel (M <= 16384)
    unsigned logM = 0;
    while ((static_cast<size_t>(1) << logM) < M) ++logM;
    // (a timer brackets all the launches of the call, as on the power-of-two paths; the pair is closed on every exit)
    struct Bracket {
        comms_fft* h;
        hipStream_t s;
        Bracket(comms_fft* hh, hipStream_t ss) : h(hh), s(ss) { h->tic(s); }
        ~Bracket() { h->toc(s); }
    } bracket(h, s);
    for (size_t b0 = 0; b0 < batch; b0 += chunk) {
        const size_t nb = batch - b0 < chunk ? batch - b0 : chunk;
        if (fuse) {
            // two launches: [x * chirp, pad, forward, * bspec] -> work;  [inverse, * chirp, first N] -> out
            const cf* chirp = reinterpret_cast<const cf*>(h->d_chirp);
            COMMS_TRY(run_rx(h->plan, in + b0 * h->N, a, nb * M, false, s,
                             BluArgs{1, static_cast<unsigned>(h->N), logM, chirp, reinterpret_cast<const cf*>(h->d_bspec)}));
            COMMS_TRY(run_rx(h->plan, a, o + b0 * h->N, nb * M, true, s,
                             BluArgs{2, static_cast<unsigned>(h->N), logM, chirp, nullptr}));
            continue;
        }
        blu_pre_kernel<<<dim3(4 * kNumCU), dim3(256), 0, s>>>(reinterpret_cast<const cf*>(in + b0 * h->N), reinterpret_cast<const cf*>(h->d_chirp), reinterpret_cast<cf*>(a), h->N, M, nb);
        COMMS_TRY(launch_ok("blu_pre_kernel"));
        COMMS_TRY(pow2_run(h->plan, a, a, nb, false, s, sc));
        blu_mul_kernel<<<dim3(4 * kNumCU), dim3(256), 0, s>>>(reinterpret_cast<cf*>(a), reinterpret_cast<const cf*>(h->d_bspec), M, nb);
        COMMS_TRY(launch_ok("blu_mul_kernel"));
        COMMS_TRY(pow2_run(h->plan, a, a, nb, true, s, sc));
        blu_post_kernel<<<dim3(4 * kNumCU), dim3(256), 0, s>>>(reinterpret_cast<const cf*>(a), reinterpret_cast<const cf*>(h->d_chirp), reinterpret_cast<cf*>(o + b0 * h->N), h->N, M, nb);
        COMMS_TRY(launch_ok("blu_post_kernel"));
    }
    return COMMS_OK;
}

comms_status_t comms_fft_run(comms_fft_t* h, const comms_c32* in, size_t n, comms_c32* out) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG(n > 0 && n % h->N == 0,
              "input length %zu is not a multiple of fft_size %zu (the reference panics)", n, h->N);
    COMMS_ARG(in && out, "NULL host pointer");
    COMMS_TRY(use_device(h->device));
    const size_t xform = h->N * sizeof(comms_c32);  // (chunks of whole transforms)
    return h->run_host_units(in, n * sizeof(comms_c32), xform, out, n * sizeof(comms_c32), xform, [&](void* d_in, void* d_out, size_t ib, size_t) {
        return comms_fft_run_dev(h, static_cast<const comms_c32*>(d_in), ib / sizeof(comms_c32), static_cast<comms_c32*>(d_out), COMMS_STREAM_HANDLE);
    });
}

comms_status_t comms_fft_set_timer(comms_fft_t* h, comms_timer_t* t) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    h->timer = t;
    return COMMS_OK;
}

comms_status_t comms_fft_destroy(comms_fft_t* h) {
    if (!h) return COMMS_OK;
    free_fft(h);
    return COMMS_OK;
}

}  // extern "C"
