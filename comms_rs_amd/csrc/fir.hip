// fir.hip -- FIR filtering of interleaved Complex<f32> streams on gfx950.
//
// Replaces fir()/batch_fir() (reference src/filter/fir.rs:43-54, :87-102):
//     y[n] = sum_{k<N} taps[k] * x[n-k],   history persists across calls.
//
// Kernels behind one handle (fir_pick chooses from a measured cost model):
//   * fir_direct_kernel   time domain: 256 lanes x 8 consecutive outputs, input tile + taps in LDS,
//                         a sliding window per lane in VGPRs, packed FMAs.  Very short filters and
//                         radio-sized batches of short ones (launch-bound there: ~8 us).
//   * fir_os1024_kernel   overlap-save, one WAVE per 1024-point segment, barrier-free (<= 257 taps;
//                         two-row halo up to 129 taps).  The headline kernel; MODE != 0 fuses the
//                         mixer / decimator / FM demod of comms_chain_* into the same launch.
//   * fir_os4096_kernel   overlap-save, one workgroup per 4096-point segment (258 ... 2049 taps).
//   * fir_os16k_kernel    overlap-save, 16384 = 16 x 1024 per 16-wave workgroup (2050 ... 4097 taps,
//                         config 5; longer filters as 4097-tap partitions).
//   * pulse_poly_kernel   polyphase pulse shaper (taps in SGPR pairs); pulse_kernel as fallback.
// (fir_decim.hip holds the time-domain decimating chain kernel.)
//
// HBM layout: input/output are plain contiguous float2 streams.  The handle
// keeps the last N input samples in a small device ring (two buffers,
// ping-pong) in TIME order (oldest first); the reference's `state` vector is
// the same data newest-first.
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <utility>
#include <vector>

#include <hip/hip_ext.h>

#include <atomic>
#include <cstdio>

#include "common.hpp"
#include "fft_radix.hpp"
#include "fir_handle.hpp"
#include "sgpr_mac.hpp"

namespace comms {

// ---------------------------------------------------------------- direct form
constexpr int DT = 8;             // consecutive outputs per lane
constexpr int DTILE = 256 * DT;   // outputs per workgroup
constexpr int DROW = 10;          // LDS row: 8 samples + 2 pad (80 B): conflict-free ds_read_b128
constexpr int DIRECT_MAX_TAPS = 1024;

// acc += h * x as packed-f32 FMAs on {re, im} pairs (see fft_radix.hpp for why by hand):
//   real tap h: one v_pk_fma_f32, the tap picked from the lo / hi half of a register pair;
//   complex tap: two, with the swizzle and the sign of -h.im*x.im in the modifiers.
__device__ __forceinline__ void mac_real_lo(cf& acc, cf x, cf hp) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(x), "v"(hp));
}
__device__ __forceinline__ void mac_real_hi(cf& acc, cf x, cf hp) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(x), "v"(hp));
}
__device__ __forceinline__ void mac_cplx(cf& acc, cf h, cf x) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(h), "v"(x));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(acc) : "v"(h), "v"(x));
}
typedef float cf2 __attribute__((ext_vector_type(4)));  // two packed complex values (16 B)

template <bool REAL_TAPS, class In = const float2*>
__global__ __launch_bounds__(256) void fir_direct_kernel(In in,
                                                         const float2* __restrict__ hist,
                                                         int hist_len,
                                                         const float2* __restrict__ taps_pad,
                                                         int NP, float2* __restrict__ out,
                                                         size_t n,
                                                         float2* __restrict__ new_hist) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x;
    hist_advance(hist, in, n, new_hist, hist_len);
    const int nrows = (DTILE + NP) / 8;
    cf* xt = reinterpret_cast<cf*>(smem);
    cf* tp = xt + nrows * DROW;  // taps (complex) or, if REAL_TAPS, NP floats

    const size_t o0 = static_cast<size_t>(blockIdx.x) * DTILE;
    if (REAL_TAPS) {
        float* tr = reinterpret_cast<float*>(tp);
        for (int k = t; k < NP; k += 256) tr[k] = taps_pad[k].x;
    } else {
        for (int k = t; k < NP; k += 256) tp[k] = to_cf(taps_pad[k]);
    }
    const long long g0 = static_cast<long long>(o0) - NP;
    for (int q = t; q < DTILE + NP; q += 256)
        xt[(q >> 3) * DROW + (q & 7)] = to_cf(stream_at(in, hist, hist_len, g0 + q, n));
    __syncthreads();

    cf acc[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i) acc[i] = cf{0.f, 0.f};

    cf wh[8], wl[8];
    {
        const cf2* r = reinterpret_cast<const cf2*>(xt + (NP / 8 + t) * DROW);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const cf2 v = r[j];
            wh[2 * j] = cf{v.x, v.y};
            wh[2 * j + 1] = cf{v.z, v.w};
        }
    }
    const int nchunks = NP / 8;
    for (int c = 0; c < nchunks; ++c) {
        const cf2* r = reinterpret_cast<const cf2*>(xt + (NP / 8 + t - c - 1) * DROW);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const cf2 v = r[j];
            wl[2 * j] = cf{v.x, v.y};
            wl[2 * j + 1] = cf{v.z, v.w};
        }
        if (REAL_TAPS) {
            const cf2* hp = reinterpret_cast<const cf2*>(reinterpret_cast<const float*>(tp) + 8 * c);
            const cf2 ha = hp[0], hb = hp[1];
            const cf h[4] = {cf{ha.x, ha.y}, cf{ha.z, ha.w}, cf{hb.x, hb.y}, cf{hb.z, hb.w}};  // (h0,h1) (h2,h3) ...
#pragma unroll
            for (int o = 0; o < DT; ++o) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int idx = 8 + o - j;
                    const cf x = idx >= 8 ? wh[idx - 8] : wl[idx];
                    if (j & 1)
                        mac_real_hi(acc[o], x, h[j >> 1]);
                    else
                        mac_real_lo(acc[o], x, h[j >> 1]);
                }
            }
        } else {
            const cf2* hp = reinterpret_cast<const cf2*>(tp + 8 * c);
            cf h[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const cf2 v = hp[j];
                h[2 * j] = cf{v.x, v.y};
                h[2 * j + 1] = cf{v.z, v.w};
            }
#pragma unroll
            for (int o = 0; o < DT; ++o) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int idx = 8 + o - j;
                    mac_cplx(acc[o], h[j], idx >= 8 ? wh[idx - 8] : wl[idx]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) wh[j] = wl[j];
    }

    // A lane's DT outputs are 64 contiguous bytes: stored as they are, one instruction would write 16-byte pieces at a
    // stride of 64 B.  They go through LDS instead (the sample window is dead once every wave has left the tap loop; a
    // wave reads back only what it wrote), so that each store instruction covers 1 KiB of whole lines.
    __syncthreads();
    cf* ex = xt + (t & ~63) * DROW;           // the wave's 512 outputs in output order, a lane's eight in an 80-byte row
    const int lane = t & 63;
    {
        cf2* w4 = reinterpret_cast<cf2*>(ex + lane * DROW);
#pragma unroll
        for (int j = 0; j < 4; ++j) w4[j] = cf2{acc[2 * j].x, acc[2 * j].y, acc[2 * j + 1].x, acc[2 * j + 1].y};
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const size_t ow = o0 + static_cast<size_t>(t & ~63) * DT;  // the wave's first output
    const unsigned left = ow < n ? static_cast<unsigned>(n - ow < 64u * DT ? n - ow : 64u * DT) : 0u;
#pragma unroll
    for (int i = 0; i < DT / 2; ++i) {
        const unsigned e = (static_cast<unsigned>(i) * 64u + lane) * 2u;
        const cf* src = ex + (e >> 3) * DROW + (e & 7u);
        if (e + 1 < left) {
            const cf2 v = *reinterpret_cast<const cf2*>(src);
            float2 q[2] = {make_float2(v.x, v.y), make_float2(v.z, v.w)};
            __builtin_memcpy(out + ow + e, q, 16);
        } else if (e < left) {
            out[ow + e] = to_f2(src[0]);
        }
    }
}

// ---------------------------------------------------------------- overlap-save, F = 4096
constexpr int OSF = 4096;
constexpr int XF = 16384;
constexpr int OS_S1 = 272;  // [k0][256 + 16]: odd k0 rows land 32 banks away (ds_read_b64)
constexpr int OS_S2 = 18;   // [row][16 + 2]: 144-B rows, conflict-free ds_read_b128
constexpr int OS_LDS = 4608;

struct OsTables {
    const cf* tw1;   // [16][256]  W4096^{t*k0}
    const cf* tw2;   // [16][16]   W256^{lo*j}, index [j][lo]
    const cf* hdev;  // [16][256]  H[k0 + 16*k1 + 256*k2] / 4096 at [k2][16*k0 + k1]
};

__device__ __forceinline__ void lds_read16_contig(const cf* __restrict__ p, cf (&v)[16]) {
    const cf2* r = reinterpret_cast<const cf2*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const cf2 x = r[j];
        v[2 * j] = cf{x.x, x.y};
        v[2 * j + 1] = cf{x.z, x.w};
    }
}

// DEC: the decimating chains of filters too long for the one-wave kernels (comms_chain_*: 258 ... 1537 taps at the rates the
// polyphase kernel does not run) -- mixer and decimator in the store stage: of a segment's outputs only those whose index the
// rate divides leave, times the oscillator's rotor at that index, to out[index / rate].  (Until round 5 these chains stored every
// output and ran a mixer-decimator pass over them.)
struct OsDec {
    uint64_t turns0, frac;  // oscillator phase of the call's first sample and its step, in turns x 2^64
    unsigned rate;
    unsigned c1, d1;        // 256 % rate, 256 / rate: from one row of a segment to the next
    unsigned dr;            // ((segments per step) x V) % rate and / rate: from one of a workgroup's segments to its next
    unsigned long long dq;
    float2 step[16];        // e^{i 256 j dphi}
};

// (cos, sin) of 2 pi u / 2^32: a 64-entry table for the upper six bits, a short series for the rest (|error| ~ 1e-7)
__device__ __forceinline__ cf os_table_rotor(unsigned u, const cf* sc) {
    const cf tq = sc[u >> 26];
    const float th = static_cast<float>(u & 0x3FFFFFFu) * 1.4629180792671596e-09f;  // 2 pi / 2^32
    const float z = th * th;
    float sp = __builtin_fmaf(z, 8.3333333e-3f, -1.6666667e-1f);
    sp = __builtin_fmaf(z, sp, 1.0f);
    const float sn = th * sp;
    float cp = __builtin_fmaf(z, -1.3888889e-3f, 4.1666667e-2f);
    cp = __builtin_fmaf(z, cp, -0.5f);
    const float cs = __builtin_fmaf(z, cp, 1.0f);
    return cmulf(tq, cf{cs, sn});
}

template <int WPS, class In = const float2*, bool DEC = false>
__global__ __launch_bounds__(256, WPS) void fir_os4096_kernel(In in,
                                                            const float2* __restrict__ hist,
                                                            int hist_len, float2* __restrict__ out,
                                                            size_t n, int hblk, size_t nseg,
                                                            OsTables tb, float2* __restrict__ new_hist,
                                                            int delay, int accumulate, int interleave, OsDec dec) {
    __shared__ __attribute__((aligned(16))) cf lds[OS_LDS];
    __shared__ cf sc[DEC ? 64 : 1];  // e^{2 pi i m / 64}
    const int t = threadIdx.x;
    if (!accumulate) hist_advance(hist, in, n, new_hist, hist_len);  // once per call (first partition)
    const int hi = t >> 4, lo = t & 15;
    if (DEC && t < 64) {  // from the stage-1 twiddles: W4096^{8 x 8m} = e^{-2 pi i m / 64}, m < 32; the other half by symmetry
        const cf w = tb.tw1[8 * 256 + 8 * (t & 31)];
        sc[t] = t < 32 ? cf{w.x, -w.y} : cf{-w.x, w.y};
    }

    // persistent per-lane constants, all in VGPRs: stage-1 twiddles, the filter spectrum and (round 3) the lane's
    // column of the stage-2 twiddle table W256^{lo*j} -- it depends on the lane only, and as an LDS table (in the same
    // array as the exchange buffers, so every read had to wait for the write before it) it cost two chains of fifteen
    // read -> wait -> multiply -> write steps per segment
    cf tw1r[16], hr[16], tw2r[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        tw1r[j] = tb.tw1[j * 256 + t];
        hr[j] = tb.hdev[j * 256 + t];
        tw2r[j] = tb.tw2[j * 16 + lo];
    }

    const int H = 256 * hblk;
    const int V = OSF - H;
    cf v[16];

    // workgroup b of the persistent grid owns segments [b*nseg/G, (b+1)*nseg/G)
    // (interleave: segments b, b + G, ... instead -- segments are independent, every one loads its own halo)
    const size_t seg_lo = interleave ? blockIdx.x : static_cast<size_t>(blockIdx.x) * nseg / gridDim.x;
    const size_t seg_hi = interleave ? nseg : static_cast<size_t>(blockIdx.x + 1) * nseg / gridDim.x;
    const size_t seg_step = interleave ? gridDim.x : 1;
    // DEC: output index of the lane's first row = rate x kq + kr, carried from segment to segment
    unsigned long long kq = 0;
    unsigned kr = 0;
    if (DEC) {
        const unsigned long long o_first = static_cast<unsigned long long>(seg_lo) * V + t;
        kq = o_first / dec.rate;
        kr = static_cast<unsigned>(o_first - kq * dec.rate);
    }
    for (size_t seg = seg_lo; seg < seg_hi; seg += seg_step) {
        // partition p of a long filter sees the stream delayed by p*2049 samples
        const long long base = static_cast<long long>(seg) * V - H - delay;
        // ---- forward stage 1: lane (b,c) = t holds x[256a + t]; DFT over a -> k0
        const bool interior = base >= 0 && static_cast<size_t>(base) + OSF <= n;
        if (interior) {
#pragma unroll
            for (int a = 0; a < 16; ++a) v[a] = to_cf(in[base + 256 * a + t]);
        } else {
#pragma unroll
            for (int a = 0; a < 16; ++a) v[a] = to_cf(stream_at(in, hist, hist_len, base + 256 * a + t, n));
        }
        radix16<-1>(v);
        __syncthreads();  // previous segment's last LDS reads are done
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            cf x = v[R16_POS(k)];
            if (k) x = cmulf(x, tw1r[k]);
            lds[k * OS_S1 + t] = x;
        }
        __syncthreads();
        // ---- stage 2: lane (k0,c): DFT over b -> k1
#pragma unroll
        for (int b = 0; b < 16; ++b) v[b] = lds[hi * OS_S1 + 16 * b + lo];
        __syncthreads();
        radix16<-1>(v);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            cf x = v[R16_POS(k)];
            if (k) x = cmulf(x, tw2r[k]);
            lds[(hi * 16 + k) * OS_S2 + lo] = x;
        }
        __syncthreads();
        // ---- stage 3: lane (k0,k1): DFT over c -> k2
        lds_read16_contig(lds + t * OS_S2, v);
        __syncthreads();
        radix16<-1>(v);
        // ---- spectrum multiply (1/4096 folded into hr) and inverse stage 3': over k2 -> c
        cf w[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) w[k] = cmulf(v[R16_POS(k)], hr[k]);
        radix16<1>(w);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            cf x = w[R16_POS(c)];
            if (c) x = cmulcf(x, tw2r[c]);
            lds[hi * 288 + c * OS_S2 + lo] = x;  // [k0][c][k1]
        }
        __syncthreads();
        // ---- inverse stage 2': lane (k0,c): over k1 -> b
        lds_read16_contig(lds + hi * 288 + lo * OS_S2, v);
        __syncthreads();
        radix16<1>(v);
#pragma unroll
        for (int b = 0; b < 16; ++b) lds[hi * OS_S1 + 16 * b + lo] = v[R16_POS(b)];
        __syncthreads();
        // ---- inverse stage 1': lane (b,c) = t: over k0 -> a
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = lds[k * OS_S1 + t];  // (all sixteen reads out before the first multiply)
#pragma unroll
        for (int k = 1; k < 16; ++k) v[k] = cmulcf(v[k], tw1r[k]);
        radix16<1>(v);
        // ---- store the V valid outputs: y[256a + t], a >= hblk
        const size_t obase = seg * static_cast<size_t>(V) + t;
        if (DEC) {
            const uint64_t tl = dec.turns0 + static_cast<uint64_t>(obase) * dec.frac;
            const cf rot0 = os_table_rotor(static_cast<unsigned>(tl >> 32), sc);
            float2* outq = out + kq;  // (rows: 32-bit offsets from the segment's first kept output)
            unsigned q = 0, r = kr;
            const int rows = obase < n ? static_cast<int>((n - obase + 255) >> 8 < 16 ? (n - obase + 255) >> 8 : 16) : 0;
#pragma unroll
            for (int a = 1; a < 16; ++a) {
                if (a >= hblk) {
                    if (r == 0 && a - hblk < rows) {
                        const cf rot = cmulf(rot0, to_cf(dec.step[a - hblk]));
                        outq[q] = to_f2(cmulf(v[R16_POS(a)], rot));
                    }
                    r += dec.c1;
                    const unsigned wrap = r >= dec.rate ? 1u : 0u;
                    r -= wrap ? dec.rate : 0u;
                    q += dec.d1 + wrap;
                    __builtin_amdgcn_sched_barrier(0);  // (row by row: hoisting the fifteen rotors costs thirty registers the kernel does not have)
                }
            }
            kr += dec.dr;
            const unsigned wrap = kr >= dec.rate ? 1u : 0u;
            kr -= wrap ? dec.rate : 0u;
            kq += dec.dq + wrap;
            continue;
        }
#pragma unroll
        for (int a = 1; a < 16; ++a) {
            if (a >= hblk) {
                size_t o = obase + static_cast<size_t>(256 * (a - hblk));
                if (o < n) {
                    cf y = v[R16_POS(a)];
                    if (accumulate) y = y + to_cf(out[o]);
                    out[o] = to_f2(y);
                }
            }
        }
    }
}


// ---------------------------------------------------------------- overlap-save, F = 1024, one WAVE per segment
// For filters of up to 257 taps.  A 64-lane workgroup (one wavefront) owns a run
// of consecutive 1024-point segments (768 new samples each + 256 of halo that it
// keeps in VGPRs from the previous segment), so there is no workgroup barrier
// anywhere: all latency hiding is wave-level multithreading (3-4 waves/SIMD).
//   n = 64a + 4b + c   (a,b < 16, c < 4)      k = k0 + 16 k1 + 256 k2   (k2 < 4)
//   fwd:  R16 over a | x W1024^{lane*k0} | LDS | R16 over b | x W64^{c*k1} | R4 over c ACROSS lanes
//   inv:  the mirror image; the 1/1024 is folded into the filter spectrum.
// The radix-4 over c = lane >> 4 needs no LDS: v_permlane32_swap / v_permlane16_swap bring the four
// lanes' values together (os1024_core); measured 2-4 % faster than the LDS exchange it replaced
// (scripts/ab_libs.py), and it leaves one LDS round trip per transform instead of two.
// LDS per wave: [k0][64+2] rows for the exchange (stride chosen against bank conflicts): 8.5 KiB.
constexpr int WF = 1024;
constexpr int WV = 768;        // new samples per segment
constexpr int W_S1 = 66;   // exchange 1: [k0][64+2] -- conflict-free ds_read_b64 by lanes (k0,c)
constexpr int W_S4 = 65;   // exchange 4: [k0][64+1] -- conflict-free ds_write_b64 by lanes (k0,c)
constexpr int W_P = 272;
constexpr int W_LDS = 4 * W_P;  // 1088 >= 16*66

struct WTables {
    const cf* tw1;   // [16][64]  W1024^{lane*k0}
    const cf* tw2;   // [16][4]   W64^{c*k1}, index [k1][c]
    const cf* hdev;  // [16][64]  H[k0 + 16 k1 + 256 t]/1024 at [4t + m][lane], k0 = lane&15, k1 = 2m + 8((lane>>4)&1) + (lane>>5)
};

// Optional stages fused around the 1024-point overlap-save FIR (comms_chain_*):
//   MODE bit 0: mixer BEFORE the FIR (on load)      bit 1: mixer AFTER the FIR
//   MODE bit 2: keep every `rate`-th output          bit 3: FM demod of the kept outputs
// The mixer rotor of sample i (relative to this call) is rot(turns0 + i*frac); a lane
// evaluates it once per run with an f64 sincos, advances it per segment with one f64
// rotor, and reaches the 16 rows of a segment with the wave-uniform f32 rotors
// step_a[a] = e^{i*64a*dphi}.
constexpr int CH_STAMP = 16;  // diagnostic: per-phase cycle stamps into fm_prev_new (scripts/stamp_fir.py)
constexpr int CH_TRACE = 32;  // diagnostic: per-wave start / set-up / end times, production geometry (scripts/trace_fir.py)
constexpr int CH_PRE = COMMS_CHAIN_PRE, CH_POST = COMMS_CHAIN_POST, CH_DEC = COMMS_CHAIN_DEC, CH_FM = COMMS_CHAIN_FM;
struct ChainArgs {
    uint64_t turns0, frac;
    double seg_c, seg_s;      // e^{i*768*dphi}
    float2 step_a[16];        // e^{i*64a*dphi}
    unsigned rate;            // decimation rate (>= 1)
    unsigned q_a[16], r_a[16];  // (64*(a-4)) / rate and % rate for a = 4..15
    unsigned q_seg, r_seg;      // 768 / rate and 768 % rate
    const float2* fm_prev;    // FM.prev before this call
    float2* fm_prev_new;      //   ... and after it (ping-pong)
};
constexpr double kTwoPiF = 2.0 * 3.14159265358979323846264338327950288;

// Orders one wave's LDS traffic (other lanes' writes -> this lane's reads).  A
// wavefront's DS instructions execute in issue order, so no s_waitcnt or
// s_barrier is needed -- only a compiler-level fence.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The NR new 64-sample rows of the segment that starts at sample nb (zeros past the end)
template <int NR, class In>
__device__ __forceinline__ void load_rows(In in, size_t nb, int l, size_t n,
                                          cf (&r)[NR]) {
    if (nb + 64 * NR <= n) {
#pragma unroll
        for (int a = 0; a < NR; ++a) r[a] = to_cf(in[nb + 64 * a + l]);
    } else {
#pragma unroll
        for (int a = 0; a < NR; ++a) {
            const size_t g = nb + 64 * a + l;
            r[a] = g < n ? to_cf(in[g]) : cf{0.f, 0.f};
        }
    }
}

// One segment through the filter: the 16 rows v[a] (samples 64a + lane of the 1024-point segment)
// -> forward transform, spectrum multiply, inverse transform -> v[R16_POS(a)] = filtered row a.
// `lds` is the calling wave's private exchange buffer; stamp(i) marks the diagnostic phases.
// hsp[i * HS + l] is the spectrum factor of register i (LDS table, HS = 64; or global memory, HS = 1024: the
// 16384-point kernel, whose 16 waves each filter one 1024-point slice with their own part of the spectrum).
// t1r / t2r: this lane's column of the stage-1 / stage-2 twiddle tables held in registers by the caller (the tables are read-only,
// but the compiler must reload them from LDS for every segment), or null
template <int HS = 64, int RT = 0, class Stamp>
__device__ __forceinline__ void os1024_core_rt(cf (&v)[16], cf* lds, const cf* tw1, const cf* hsp, const cf* tw2, int l,
                                               Stamp&& stamp, int hl, const cf (&t1r)[16], const cf (&t2r)[16]) {
    if (HS == 64) hl = l;  // hl: this lane's column of the spectrum table (the 16384-point kernel: its thread index,
                           // on a workgroup-uniform base pointer -- the loads then take the SGPR-base form and share
                           // one offset register instead of sixteen 64-bit addresses)
    const int q0 = l & 15, q1 = l >> 4;  // stage 2 and 3: lane (k0, c) = (q0, q1)
    // ---- forward
    radix16<-1>(v);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        cf x = v[R16_POS(k)];
        if (k) x = cmulf(x, (RT & 2) ? t1r[k] : tw1[k * 64 + l]);
        lds[k * W_S1 + l] = x;
    }
    wave_lds_sync();
    stamp(1);  // R16 + twiddle + exchange-1 writes
#pragma unroll
    for (int b = 0; b < 16; ++b) v[b] = lds[q0 * W_S1 + 4 * b + q1];
    wave_lds_sync();
    stamp(2);  // exchange-1 reads
    // the 16384-point kernel's spectrum lives in global memory (L2): all sixteen loads go out here, a
    // radix-16 and the cross-lane radix-4 ahead of their use (one at a time at the multiply they cost
    // sixteen L2 round trips per segment)
    cf hh[HS == 64 ? 1 : 16];
    // ---- stage 3 without LDS: lane (k0, c) keeps its 16 k1 values and the radix-4 over c = lane >> 4 runs
    // ACROSS lanes: v_permlane32_swap pairs registers so that the c1 halves meet in one lane, v_permlane16_swap
    // does the same for c0; the W4^{c0} twiddle of the odd outputs is folded into the second butterfly.
    // After it, register 4t + m of lane k0 + 16g + 32h holds Z[k0 + 16 (2m + 8g + h) + 256 t].
    radix16<-1>(v);
    cf r[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        r[k] = v[R16_POS(k)];
        if (k) r[k] = cmulf(r[k], (RT & 1) ? t2r[k] : tw2[k * 4 + q1]);
    }
    stamp(3);  // R16 + twiddle (stage 2)
    // The 16384-point kernel's spectrum lives in global memory (L2).  Its sixteen loads go out in two batches
    // ahead of their use -- here and between the two swap layers (the lane-swap fences keep them in place) --
    // instead of one L2 round trip at a time at the multiply.
    __amdgpu_buffer_rsrc_t hrs = make_rsrc(hsp, HS == 64 ? 0 : 16 * HS * sizeof(cf));  // (workgroup-uniform base)
    if (HS != 64) {
#pragma unroll
        for (int i = 0; i < 8; ++i) hh[i] = to_cf(BufRows<const float2*>::get_from(hrs, hl * 8, i * HS * 8));
    }
    cf sd[16];  // [0..7] sums (even k2), [8..15] differences (odd k2) of the c1 halves
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        cf p = r[2 * m], q = r[2 * m + 1];
        lane_swap32(p, q);
        sd[m] = p + q;
        sd[8 + m] = p - q;
    }
    if (HS != 64) {
#pragma unroll
        for (int i = 8; i < 16; ++i) hh[i] = to_cf(BufRows<const float2*>::get_from(hrs, hl * 8, i * HS * 8));
    }
    cf z[16];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        cf a = sd[m], b = sd[m + 4];
        lane_swap16(a, b);
        z[m] = a + b;        // k2 = 0
        z[8 + m] = a - b;    // k2 = 2
        cf c = sd[8 + m], d = sd[12 + m];
        lane_swap16(c, d);
        z[4 + m] = cadd_mi(c, d);   // k2 = 1: c + (-i) d
        z[12 + m] = cadd_pi(c, d);  // k2 = 3: c - (-i) d
    }
    stamp(4);  // radix-4 across lanes
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = cmulf(z[i], HS == 64 ? hsp[i * HS + hl] : hh[i]);
    // ---- the mirror image
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        cf a = z[m] + z[8 + m], b = z[m] - z[8 + m];
        lane_swap16(a, b);
        sd[m] = a;
        sd[m + 4] = b;
        const cf e = z[4 + m], f = z[12 + m];
        cf c = e + f, d = crot_sub(e, f);  // d = i (e - f)
        lane_swap16(c, d);
        sd[8 + m] = c;
        sd[12 + m] = d;
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        cf p = sd[m] + sd[8 + m], q = sd[m] - sd[8 + m];
        lane_swap32(p, q);
        r[2 * m] = p;
        r[2 * m + 1] = q;
    }
    stamp(5);  // spectrum multiply + inverse radix-4 across lanes
    // ---- inverse: lane (k0,c) = (q0,q1): conj W64^{c*k1}, R16 over k1 -> b
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = k ? cmulcf(r[k], (RT & 1) ? t2r[k] : tw2[k * 4 + q1]) : r[k];
    stamp(6);  // conj twiddle (stage 2)
    radix16<1>(v);
#pragma unroll
    for (int b = 0; b < 16; ++b) lds[q0 * W_S4 + 4 * b + q1] = v[R16_POS(b)];
    wave_lds_sync();
    stamp(7);  // R16 + exchange-4 writes
    // ---- inverse: lane t: conj W1024^{t*k0}, R16 over k0 -> a
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        cf x = lds[k * W_S4 + l];
        v[k] = k ? cmulcf(x, (RT & 2) ? t1r[k] : tw1[k * 64 + l]) : x;
    }
    wave_lds_sync();
    stamp(8);  // exchange-4 reads + twiddle
    radix16<1>(v);
}

template <int HS = 64, class Stamp>
__device__ __forceinline__ void os1024_core(cf (&v)[16], cf* lds, const cf* tw1, const cf* hsp, const cf* tw2, int l,
                                            Stamp&& stamp, int hl = -1) {
    const cf none[16] = {};
    os1024_core_rt<HS, 0>(v, lds, tw1, hsp, tw2, l, stamp, hl, none, none);
}

// Diagnostic builds only: when a wave started, finished the workgroup's set-up and ended
// (s_memrealtime, 100 MHz), its shader cycles and where it ran; 8 words per wave slot.
template <bool ON>
struct WaveTrace {
    unsigned long long r0 = 0, c0 = 0, r1 = 0;
    __device__ __forceinline__ WaveTrace() {
        if (ON) {
            r0 = __builtin_amdgcn_s_memrealtime();
            c0 = __builtin_amdgcn_s_memtime();
        }
    }
    __device__ __forceinline__ void setup_done() {
        if (ON) r1 = __builtin_amdgcn_s_memrealtime();
    }
    __device__ __forceinline__ void write(void* buf, size_t slot, int lane, size_t segments) const {
        if (!ON) return;
        const unsigned long long r2 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long c2 = __builtin_amdgcn_s_memtime();
        if (lane == 0) {
            unsigned long long* t = static_cast<unsigned long long*>(buf) + slot * 8;
            t[0] = r0;
            t[1] = r1;
            t[2] = r2;
            t[3] = c2 - c0;
            t[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);  // HW_ID
            t[5] = __builtin_amdgcn_s_getreg((3 << 11) | 20);  // XCC_ID
            t[6] = segments;
            t[7] = slot;
        }
    }
};

// WPB waves per workgroup share the read-only tables in LDS (stage-1 twiddles,
// filter spectrum, W64 table: 16.5 KiB); every wave has a private 8.5 KiB
// exchange buffer and runs on its own -- no workgroup barrier after set-up.
// HR = halo rows of 64 samples: 4 (up to 257 taps, 768 new samples per segment) or, for the plain
// FIR, as few as the taps need: 3 / 2 / 1 rows for <= 193 / 129 / 65 taps (832 / 896 / 960 new samples
// per segment from the same two transforms).
template <int WPB, int MINW, int MODE, int HR = 4, class In = const float2*>
__global__ __launch_bounds__(64 * WPB, MINW) void fir_os1024_kernel(In in,
                                                                    const float2* __restrict__ hist,
                                                                    int hist_len,
                                                                    float2* __restrict__ out, size_t n,
                                                                    size_t nseg, size_t n_runs,
                                                                    WTables tb,
                                                                    float2* __restrict__ new_hist,
                                                                    ChainArgs ch) {
    static_assert(HR == 4 || (HR >= 1 && HR < 4 && (MODE & ~CH_TRACE) == 0), "the short halos are for the plain FIR only");
    constexpr int WVK = 1024 - 64 * HR, HALO = 64 * HR, NEWR = 16 - HR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    WaveTrace<(MODE & CH_TRACE) != 0> trace;
    hist_advance(hist, in, n, new_hist, hist_len);
    cf* tw1 = reinterpret_cast<cf*>(smem);  // [16][64]
    cf* hsp = tw1 + 1024;                       // [16][64]
    cf* tw2 = hsp + 1024;                       // [16][4]
    const int l = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    cf* lds = tw2 + 64 + wave * W_LDS;          // this wave's exchange buffer

    for (int i = threadIdx.x; i < 1024; i += 64 * WPB) {
        tw1[i] = tb.tw1[i];
        hsp[i] = tb.hdev[i];
    }
    if (threadIdx.x < 64) tw2[threadIdx.x] = tb.tw2[threadIdx.x];
    __syncthreads();
    trace.setup_done();

    // run r of n_runs owns segments [r*nseg/n_runs, (r+1)*nseg/n_runs): equal shares
    // (+-1) for every wave, so every CU carries the same load
    const size_t run = static_cast<size_t>(blockIdx.x) * WPB + wave;
    const size_t seg0 = run < n_runs ? run * nseg / n_runs : nseg;
    const size_t seg1 = run < n_runs ? (run + 1) * nseg / n_runs : nseg;

    // mixer rotor of this lane's first row (sample seg*768 - 256 + l), f64
    double rot_c = 1.0, rot_s = 0.0;
    if ((MODE & (CH_PRE | CH_POST)) && seg0 < seg1) {
        const long long i0 = static_cast<long long>(seg0 * WVK) - HALO + l;
        const uint64_t turns = ch.turns0 + static_cast<uint64_t>(i0) * ch.frac;
        sincos(static_cast<double>(turns >> 11) * (kTwoPiF * 0x1.0p-53), &rot_s, &rot_c);
    }

    // diagnostic build only (MODE bit 4): cycle stamps per phase, summed per wave
    unsigned long long st_acc[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_prev = 0;
#define OS_STAMP(i)                                                        \
    if (MODE & CH_STAMP) {                                                 \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");        \
        const unsigned long long st_now = __builtin_amdgcn_s_memtime();    \
        st_acc[i] += st_now - st_prev;                                     \
        st_prev = st_now;                                                  \
    }
    if (MODE & CH_STAMP) st_prev = __builtin_amdgcn_s_memtime();

    cf v[16], carry[HR], nxt[NEWR];
    size_t dec_q = 0;
    unsigned dec_r = 0;
    for (size_t seg = seg0; seg < seg1; ++seg) {
        const size_t nb = seg * WVK;  // first new sample of this segment
        cf rot = cf{static_cast<float>(rot_c), static_cast<float>(rot_s)};
        if (MODE & (CH_PRE | CH_POST)) {  // advance to the next segment's first row
            const double nc = rot_c * ch.seg_c - rot_s * ch.seg_s;
            rot_s = rot_c * ch.seg_s + rot_s * ch.seg_c;
            rot_c = nc;
        }
        if (seg == seg0) {
#pragma unroll
            for (int a = 0; a < HR; ++a) {
                v[a] = to_cf(stream_at(in, hist, hist_len, static_cast<long long>(nb) - HALO + 64 * a + l, n));
                if (MODE & CH_PRE) v[a] = cmulf(v[a], a ? cmulf(rot, to_cf(ch.step_a[a])) : rot);
            }
        } else {
#pragma unroll
            for (int a = 0; a < HR; ++a) v[a] = carry[a];
        }
        // (A register prefetch of the next segment's rows was tried and lost: vmcnt is one
        // in-order counter for loads and stores, so the wait for prefetched rows also
        // drains the previous segment's 12 stores -- 57.6 -> 65 us.)
        load_rows(in, nb, l, n, nxt);
#pragma unroll
        for (int a = HR; a < 16; ++a) v[a] = nxt[a - HR];
        if (MODE & CH_PRE) {
#pragma unroll
            for (int a = 4; a < 16; ++a) v[a] = cmulf(v[a], cmulf(rot, to_cf(ch.step_a[a])));
        }
#pragma unroll
        for (int a = 0; a < HR; ++a) carry[a] = v[16 - HR + a];
        OS_STAMP(0)  // global loads landed

        os1024_core(v, lds, tw1, hsp, tw2, l, [&](int i) { OS_STAMP(i) });
        if ((MODE & ~(CH_STAMP | CH_TRACE)) == 0) {
            if (nb + WVK <= n) {
#pragma unroll
                for (int a = HR; a < 16; ++a) out[nb + 64 * (a - HR) + l] = to_f2(v[R16_POS(a)]);
            } else {
#pragma unroll
                for (int a = HR; a < 16; ++a) {
                    const size_t o = nb + 64 * (a - HR) + l;
                    if (o < n) out[o] = to_f2(v[R16_POS(a)]);
                }
            }
        } else {
            if (MODE & CH_POST) {
#pragma unroll
                for (int a = 4; a < 16; ++a)
                    v[R16_POS(a)] = cmulf(v[R16_POS(a)], cmulf(rot, to_cf(ch.step_a[a])));
                // FM demod of row 4 reaches back into the tail of row 3 (valid outputs of the previous
                // segment's span, recomputed here): they must carry the mixer rotation too
                if (MODE & CH_FM) v[R16_POS(3)] = cmulf(v[R16_POS(3)], cmulf(rot, to_cf(ch.step_a[3])));
            }
            // FM needs y[idx - rate]: lane l - rate of the same row, or the tail of row a-1
            const int src = (l - static_cast<int>(ch.rate)) & 63;
            float2 sh_prev = make_float2(0.f, 0.f);
            if (MODE & CH_FM) {
                const float2 y3 = to_f2(v[R16_POS(3)]);
                sh_prev = make_float2(__shfl(y3.x, src), __shfl(y3.y, src));
            }
            // idx = nb + 64(a-4) + l; kept when idx % rate == 0, written at idx / rate
            // (nb + l) / rate and % rate: one 64-bit division per run, then per-segment increments
            if (seg == seg0) {
                dec_q = (nb + l) / ch.rate;
                dec_r = static_cast<unsigned>((nb + l) - dec_q * ch.rate);
            }
            const size_t q0 = dec_q;
            const unsigned r0 = dec_r;
            dec_q += ch.q_seg;
            dec_r += ch.r_seg;
            if (dec_r >= ch.rate) {
                dec_r -= ch.rate;
                ++dec_q;
            }
#pragma unroll
            for (int a = 4; a < 16; ++a) {
                unsigned r = r0 + ch.r_a[a];
                size_t q = q0 + ch.q_a[a];
                if (r >= ch.rate) {
                    r -= ch.rate;
                    ++q;
                }
                const size_t idx = nb + 64 * (a - 4) + l;
                float2 sh_cur = make_float2(0.f, 0.f);
                if (MODE & CH_FM) {
                    const float2 ya = to_f2(v[R16_POS(a)]);
                    sh_cur = make_float2(__shfl(ya.x, src), __shfl(ya.y, src));
                }
                const float2 p_row = l >= static_cast<int>(ch.rate) ? sh_cur : sh_prev;
                sh_prev = sh_cur;
                if (r == 0 && idx < n) {
                    const float2 y = to_f2(v[R16_POS(a)]);
                    if (MODE & CH_FM) {
                        float2 p = p_row;
                        if (idx == 0) p = ch.fm_prev[0];
                        reinterpret_cast<float*>(out)[q] = fm_step(y, p);
                        if (idx + ch.rate >= n) ch.fm_prev_new[0] = y;  // last kept sample of the call
                    } else {
                        out[q] = y;
                    }
                }
            }
        }
        OS_STAMP(9)  // R16 + stores retired
    }
    if (MODE & CH_STAMP) {
        if (l == 0 && run < n_runs)
            for (int i = 0; i < 10; ++i) reinterpret_cast<unsigned long long*>(ch.fm_prev_new)[run * 10 + i] = st_acc[i];
    }
    if (MODE & CH_TRACE) trace.write(ch.fm_prev_new, blockIdx.x * WPB + wave, l, seg1 - seg0);
#undef OS_STAMP
}

// The plain FIR (no fused stages) with ticketed segments.  One 16-wave workgroup per CU owns a
// contiguous range of the stream's interior segments; its waves draw them one at a time from a
// counter in LDS and load all 16 rows of a segment themselves (no halo carried in registers: the
// halo rows were just read by the neighbouring wave, an L2 hit).  Why: a SIMD issues oldest wave
// first, so with equal fixed runs its four waves finish staggered -- 29 / 36 / 42 / 48 us of a
// 55 us launch (scripts/trace_fir.py, profiles/r01_trace_fir_os1024.txt) -- and the tail of every
// CU runs under-occupied; with tickets the older waves simply take more segments (7.3 / 6.1 /
// 4.4 / 3.5 at 2^24 samples) and all sixteen end together: 11-12 % less kernel time at 2^24 and 2^26
// samples (scripts/ab_fir.py, the variants interleaved launch by launch).  The stream's first segment
// (history) and its partial last one take the guarded path after the loop.
// (Loading the NEXT segment's rows into spare registers before transforming this one -- the wait
// then sits after the 16 - HR stores as a counted vmcnt -- was measured too: 3 us SLOWER at 2^24,
// the 32 register moves per segment cost more than the covered latency.)
#ifndef COMMS_OS1024_PREFETCH
#define COMMS_OS1024_PREFETCH 0  // trial (round 5, second form): measured level or slower, see the loop
#endif
template <int HR, bool TRACE, class In = const float2*>
__global__ __launch_bounds__(1024, 4) void fir_os1024_dyn_kernel(In in,
                                                                 const float2* __restrict__ hist, int hist_len,
                                                                 float2* __restrict__ out, size_t n, WTables tb,
                                                                 float2* __restrict__ new_hist, void* trace_buf,
                                                                 unsigned chunk_log2, KStamp ks) {
    constexpr int WVK = 1024 - 64 * HR, HALO = 64 * HR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    WaveTrace<TRACE> trace;
    kstamp_begin(ks);
    hist_advance(hist, in, n, new_hist, hist_len);
    cf* tw1 = reinterpret_cast<cf*>(smem);  // [16][64]
    cf* hsp = tw1 + 1024;                   // [16][64]
    cf* tw2 = hsp + 1024;                   // [16][4]
    const int l = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    cf* lds = tw2 + 64 + wave * W_LDS;      // this wave's exchange buffer
    unsigned* ticket = reinterpret_cast<unsigned*>(tw2 + 64 + 16 * W_LDS);
    for (int i = threadIdx.x; i < 1024; i += 1024) {
        tw1[i] = tb.tw1[i];
        hsp[i] = tb.hdev[i];
    }
    if (threadIdx.x < 64) tw2[threadIdx.x] = tb.tw2[threadIdx.x];
    if (threadIdx.x == 0) *ticket = 0;
    __syncthreads();
    trace.setup_done();

    // interior segments 1 .. nfull-1 (all 16 rows inside `in`, all outputs inside `out`).
    // chunk_log2 >= 32: every workgroup owns one contiguous share.  Otherwise the stream is cut into chunks of
    // 2^chunk_log2 segments dealt round-robin: chunk c goes to workgroup c mod G in sweep order, so the whole chip
    // reads one window of G chunks and writes the same window of the output (DRAM sees two sequential streams
    // instead of 2 G of them, spaced a power of two apart when n is one).  Sweep order runs through the
    // workgroups of one XCD first (blocks b, b + 8, ... share one under the observed round-robin placement: speed
    // only), so that the halo rows at a chunk's edge were just read into the same L2 by the neighbouring chunk.
    const size_t nfull = n / WVK;
    const size_t inner = nfull > 1 ? nfull - 1 : 0;
    const bool contiguous = chunk_log2 >= 32u;
    const unsigned G = gridDim.x;
    const unsigned wg = (G % 8u == 0u) ? (blockIdx.x % 8u) * (G / 8u) + blockIdx.x / 8u : blockIdx.x;
    const size_t lo = 1 + blockIdx.x * inner / gridDim.x;
    const size_t hi = contiguous ? 1 + (blockIdx.x + 1) * inner / gridDim.x : nfull;
    auto draw = [&]() -> size_t {  // one lane draws, the wave follows
        unsigned t = 0;
        if (l == 0) t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const unsigned tk = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(t)));
        if (contiguous) return lo + tk;
        const size_t c = static_cast<size_t>(tk >> chunk_log2) * G + wg;
        return 1 + (c << chunk_log2) + (tk & ((1u << chunk_log2) - 1u));
    };
    auto fetch = [&](size_t sg, cf (&r)[16]) {
        const size_t p = sg * WVK - HALO + l;
#pragma unroll
        for (int a = 0; a < 16; ++a) r[a] = to_cf(in[p + 64 * a]);
    };
    auto nostamp = [](int) {};
#ifndef COMMS_OS1024_REG_TW
#define COMMS_OS1024_REG_TW 1
#endif
    // The lane's column of the stage-2 twiddle table (bit 0) in registers instead of 30 LDS reads per segment: the tables are
    // read-only, but the compiler must reload them for every segment; the kernel has 64 VGPRs to spare at its sixteen waves per
    // CU.  Same values, same arithmetic.  Three builds alternating launch by launch (scripts/build_variant.sh, ab_libs.py; 255
    // taps): 2^22 samples 19.2 -> 18.8 us, 2^24 50.0 -> 49.3, 2^26 199.2 -> 198.7; with the stage-1 column as well (bit 1: 124
    // VGPRs) 18.2 / 49.5 / 198.0 -- no better where it matters, not used.
    cf t1r[16], t2r[16];
    if (COMMS_OS1024_REG_TW & 1) {
#pragma unroll
        for (int k = 1; k < 16; ++k) t2r[k] = tw2[k * 4 + (l >> 4)];
    }
    if (COMMS_OS1024_REG_TW & 2) {
#pragma unroll
        for (int k = 1; k < 16; ++k) t1r[k] = tw1[k * 64 + l];
    }

    size_t count = 0;
    cf v[16];
    // The stream's first segment (halo from the history) and its partial last one, on a guarded path, by wave 0 of
    // the first / last workgroup BEFORE it joins the ticket loop: the other fifteen waves simply draw more tickets
    // meanwhile.  (Done after the loop, as it was, that wave worked alone for one more segment -- ~5 us at the
    // single-wave rate -- while the rest of the chip had finished.)
    if (wave == 0) {
        const size_t nseg = (n + WVK - 1) / WVK;
        for (int e = 0; e < 2; ++e) {
            const size_t sg = e ? nseg - 1 : 0;
            if (e ? (blockIdx.x != gridDim.x - 1 || nseg < 2 || nseg == nfull) : blockIdx.x != 0) continue;
            const size_t nb = sg * WVK;
#pragma unroll
            for (int a = 0; a < HR; ++a)
                v[a] = to_cf(stream_at(in, hist, hist_len, static_cast<long long>(nb) - HALO + 64 * a + l, n));
            cf nw[16 - HR];
            load_rows(in, nb, l, n, nw);
#pragma unroll
            for (int a = HR; a < 16; ++a) v[a] = nw[a - HR];
            os1024_core_rt<64, COMMS_OS1024_REG_TW>(v, lds, tw1, hsp, tw2, l, nostamp, -1, t1r, t2r);
#pragma unroll
            for (int a = HR; a < 16; ++a) {
                const size_t i = nb + 64 * (a - HR) + l;
                if (i < n) out[i] = to_f2(v[R16_POS(a)]);
            }
            ++count;
        }
    }
#if COMMS_OS1024_PREFETCH
    // TRIAL, not the product (-DCOMMS_OS1024_PREFETCH=1; scripts/build_variant.sh + ab_libs.py).  Two register sets taking
    // turns (the kernel needs ~64 of the 128 VGPRs its sixteen waves per CU may have): the NEXT segment's rows are requested
    // before this one is transformed, so a wave's own load latency leaves its critical path; those loads are OLDER than this
    // segment's twelve stores in the in-order vmcnt queue, so the wait for them leaves the stores in flight; no register
    // moves (round 1's attempt kept one set and moved the prefetched rows into it: 3 us slower).  Measured, the two builds
    // alternating launch by launch: 255 taps 2^22 19.5 -> 20.8 us, 2^24 50.0 -> 52.1, 2^26 206.3 -> 206.5, 2^28 801 -> 805; 127
    // taps 2^24 47.6 -> 49.2.  The same reordering is worth 9 % on fir_poly8_kernel, whose waves are short of work while they
    // wait; this kernel's four waves per SIMD already cover each other's loads -- what bounds it is what the memory system
    // gives its 1.33x-overlapped read stream and its write stream together (NOTES.md).
    cf w[16];
    auto finish = [&](size_t sg, cf (&r)[16]) {
        os1024_core_rt<64, COMMS_OS1024_REG_TW>(r, lds, tw1, hsp, tw2, l, nostamp, -1, t1r, t2r);
        float2* o = out + sg * WVK + l;
#pragma unroll
        for (int a = HR; a < 16; ++a) o[64 * (a - HR)] = to_f2(r[R16_POS(a)]);
        ++count;
    };
    size_t seg = draw();
    if (seg < hi) fetch(seg, v);
    while (seg < hi) {
        const size_t s1 = draw();
        if (s1 < hi) fetch(s1, w);
        finish(seg, v);
        if (!(s1 < hi)) break;
        seg = draw();
        if (seg < hi) fetch(seg, v);
        finish(s1, w);
    }
#else
    size_t seg = draw();
    while (seg < hi) {
        fetch(seg, v);
        const size_t seg_next = draw();  // the ticket's LDS round trip hides behind the loads
        os1024_core_rt<64, COMMS_OS1024_REG_TW>(v, lds, tw1, hsp, tw2, l, nostamp, -1, t1r, t2r);
        float2* o = out + seg * WVK + l;
#pragma unroll
        for (int a = HR; a < 16; ++a) o[64 * (a - HR)] = to_f2(v[R16_POS(a)]);
        seg = seg_next;
        ++count;
    }
#endif

    if (TRACE) trace.write(trace_buf, blockIdx.x * 16 + wave, l, count);
    kstamp_end(ks);
}

// ---------------------------------------------------------------- overlap-save, F = 16384 (long filters)
// 1538..4097 taps (config 5).  16384 = 16 x 1024: one 16-wave workgroup owns a segment
// (16 - HR rows of 1024 new samples + HR halo rows carried in VGPRs): every lane does a radix-16 over
// the 1024-strided rows, the 16 k0-slices are exchanged through LDS so that wave k0 holds slice k0 (and applies
// the stage-1 twiddle W16384^{t*k0} there), each wave then runs the same barrier-free
// 1024-point transform as fir_os1024_kernel on its slice, multiplies by its part of the
// filter spectrum, and the mirror image brings the samples back.  No workgroup barrier after set-up: two sets of
// LDS counters order the two exchanges of a segment (below).  LDS: 16 x 1090 slice buffers + W1024 / W64
// tables + counters = 158 KiB, one workgroup per CU.
constexpr int X_BUF = 1090;    // per-wave slice buffer (>= 1088; 2180 dwords = 4 mod 64 banks)
constexpr size_t X_LDS_BYTES = (1024 + 64 + 16 * X_BUF) * sizeof(float2) + 32 * sizeof(unsigned) + 3 * sizeof(unsigned long long);  // tables, slices, counters, aux
#ifndef COMMS_OS16K_CARRY
#define COMMS_OS16K_CARRY 1  // the four halo rows of a segment stay in registers from the previous one
#endif

struct XTables {
    const cf* tw1;   // [16][64]   W1024^{lane*k}          (per-wave 1024-point transform)
    const cf* tw2;   // [16][4]    W64^{c*k1}
    const cf* ta;    // [16][16]   W256^{wave*k0}           (stage-1 twiddle, high part)
    const cf* tb;    // [16][64]   W16384^{lane*k0}         (stage-1 twiddle, low part), index [k0][lane]
    const cf* hdev;  // [16][1024] H[k0 + 16 k']/16384 at [4t + m][tid], k0 = tid>>6, k' = the bin os1024_core leaves in register 4t + m of lane tid&63
};

// Workgroup synchronisation of the 16384-point kernel without s_barrier: two sets of sixteen
// monotonic counters in LDS.  slice_in[k] counts the waves that have written their part of slice k
// (stage 1 of the current segment), slice_out[k] the segments wave k has finished; a consumer polls
// (ds_read + s_sleep) until the count it needs is there.  What the three barriers per segment cost
// (profiles/r02_pmc_config5.txt: 59 % of wave cycles waiting) was not the waiting itself but its
// shape: every phase ended with the CU idling until its slowest wave arrived, and one of the three
// (before stage 1's writes) ordered nothing -- thread tid writes exactly the words it has read itself
// in the previous segment's inverse stage.  All sixteen waves of the workgroup are resident (one
// workgroup per CU) and run the same number of segments, so every count is reached.  The poll is bounded all the
// same (a hung GPU is worse than a failed call): a wave that gives up raises the handle's sticky error word (host-
// mapped, written at system scope) and goes on; the outputs of that launch are then wrong, and every later entry on
// the handle -- and the state getters -- return COMMS_ERR_DEVICE until the handle is destroyed.
__device__ __forceinline__ void x_signal(unsigned* cnt, bool lane_on) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the whole wave's LDS writes first
    if (lane_on) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// (aux: three 64-bit words in LDS -- [0] the stamp timer's end slots and [1] the error word's address, written once by
// thread 0, [2] the workgroup's give-up code.  A wave whose wait runs out only ORs its code into aux[2]; every wave
// looks at that word once, when it ends, and the one that finds it set raises the host-mapped error word.  Kept out of
// the wait loops and out of SGPRs: with the raise itself -- a 64-bit address and a system-scope atomic -- behind each
// wait the kernel ran 6 % slower (614 against 579 us at 2^27 samples, scripts/ab_libs.py on five builds), and the kernel
// sits at the scalar-register limit.)
constexpr int X_SPINS = 1 << 22;  // x ~100 cycles per poll: a few tenths of a second
#ifndef COMMS_OS16K_SLEEP_IN
#define COMMS_OS16K_SLEEP_IN 1  // s_sleep argument (x 64 cycles) between two polls of the first / second wait
#endif
#ifndef COMMS_OS16K_SLEEP_OUT
#define COMMS_OS16K_SLEEP_OUT 4  // (1 / 2 / 4 / 8 here: 591 / 589 / 585 / 588 us at 2^27 samples, seven builds in one process; IN: no difference)
#endif
// Both waits return false when they ran out: the wave then LEAVES its segment loop (the counters are cumulative: once a
// signal is missing, every later wait on that counter would run out as well, X_SPINS polls each -- a launch with
// hundreds of segments per workgroup would take minutes to report what it knew after the first).  Its own signals stop
// with it, so the other waves of the workgroup run out at their next wait and leave too: a failed launch costs a few
// timeouts, whatever its length.  The exit hangs off the cold path only.
__device__ __forceinline__ bool x_wait_one(const unsigned* cnt, unsigned target, unsigned* gave_up) {
    bool ok = false;
    for (int spin = 0; spin < X_SPINS; ++spin) {
        const unsigned c = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (static_cast<int>(__builtin_amdgcn_readfirstlane(static_cast<int>(c)) - target) >= 0) {
            ok = true;
            break;
        }
        __builtin_amdgcn_s_sleep(COMMS_OS16K_SLEEP_IN);
    }
    if (!ok) __hip_atomic_fetch_or(gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return ok;
}
__device__ __forceinline__ bool x_wait_all16(const unsigned* cnt, unsigned target, int l, unsigned* gave_up) {
    bool ok = false;
    for (int spin = 0; spin < X_SPINS; ++spin) {
        const unsigned c = __hip_atomic_load(cnt + (l & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (__all(static_cast<int>(c - target) >= 0)) {
            ok = true;
            break;
        }
        __builtin_amdgcn_s_sleep(COMMS_OS16K_SLEEP_OUT);
    }
    if (!ok) __hip_atomic_fetch_or(gave_up, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return ok;
}
// at a wave's end: report what any wait of the workgroup gave up on
__device__ __forceinline__ void x_report(const unsigned long long* aux) {
#ifndef COMMS_OS16K_NO_ERR
    const unsigned code = static_cast<unsigned>(aux[2]);
    unsigned* err = reinterpret_cast<unsigned*>(aux[1]);
    if (code && err && (threadIdx.x & 63) == 0) __hip_atomic_fetch_or(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
}

// diagnostic build: phase times of the first X_TRACE_SEGS segments of every wave (scripts/trace_os16k.py), s_memtime at
// eleven points of a segment, [workgroup][wave][segment][16] -- comms_debug_os16k_trace(buffer) arms it.
#ifdef COMMS_DIAG
constexpr unsigned X_TRACE_SEGS = 40;
__device__ unsigned long long* g_x_trace = nullptr;
#define X_MARK(i)                                                                  \
    do {                                                                           \
        if (xt && done < X_TRACE_SEGS) {                                           \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();            \
            if (l == 0) xt[done * 16 + (i)] = t_;                                  \
        }                                                                          \
    } while (0)
// (after the vector work that produced `reg`: the clock read is a scalar instruction and would float above it)
#define X_MARK_AFTER(i, reg)                      \
    do {                                          \
        if (xt) {                                 \
            asm volatile("" : "+v"(reg));         \
            X_MARK(i);                            \
        }                                         \
    } while (0)
#else
#define X_MARK(i)
#define X_MARK_AFTER(i, reg)
#endif

// DEC: mixer and decimator in the store stage (the decimating chains of 1538 ... 4097 taps: see OsDec / fir_os4096_kernel).
struct OsNoDec {};
template <int HR, class In = const float2*, bool DEC = false>
__global__ __launch_bounds__(1024, 4) void fir_os16k_kernel(In in,
                                                            const float2* __restrict__ hist, int hist_len,
                                                            float2* __restrict__ out, size_t n, size_t nseg,
                                                            XTables tb, float2* __restrict__ new_hist, int delay,
                                                            int accumulate, KStamp ks, unsigned* err, int fault,
                                                            std::conditional_t<DEC, OsDec, OsNoDec> dec = {}) {
    // err: the handle's sticky error word.  fault (diagnostic build, else 0): workgroup 0's wave 3 withholds one
    // signal, so that the waits above run out and the error path can be tested.
#ifndef COMMS_OS16K_NO_STAMP
    kstamp_begin(ks);
#endif
    // HR = halo rows of 1024 samples (1 ... 4: up to 1025 / 2049 / 3073 / 4097 taps): a segment keeps 16 - HR rows.
    // (A compile-time value: as a kernel argument the row loops turned into chains of uniform branches and the
    // kernel lost 11 % -- 591 -> 658 us at 2^27 samples, 4097 taps.)
    constexpr int hr = HR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* tw1 = reinterpret_cast<cf*>(smem);  // [16][64]
    cf* tw2 = tw1 + 1024;                   // [16][4]
    cf* bufs = tw2 + 64;                    // [16][X_BUF]
    const size_t xv = static_cast<size_t>(16 - hr) * 1024;  // new samples per segment
    unsigned* slice_in = reinterpret_cast<unsigned*>(bufs + 16 * X_BUF);  // [16]
    unsigned* slice_out = slice_in + 16;                                   // [16]
    unsigned long long* aux = reinterpret_cast<unsigned long long*>(slice_in + 32);  // [3], see x_wait_one
    unsigned* gave_up = reinterpret_cast<unsigned*>(aux + 2);
    const int tid = threadIdx.x;
    const int l = tid & 63, wave = tid >> 6;
    cf* buf = bufs + wave * X_BUF;
    if (tid == 0) {
        aux[0] = reinterpret_cast<unsigned long long>(ks.end);
        aux[1] = reinterpret_cast<unsigned long long>(err);
        aux[2] = 0;
    }
    const bool withhold = (fault & 1) != 0 && blockIdx.x == 0 && wave == 3;
    // Diagnostic build, fault bits 1...: issue priority among the four waves of a SIMD (waves w, w + 4, w + 8, w + 12).
    // The arbiter takes the oldest wave first, so with equal priorities the four finish every phase staggered
    // (scripts/trace_os16k.py).  Rotating s_setprio by step and wave group does even them out -- and the launch gets
    // 2 ... 6 % longer (profiles/r04_trace_os16k.txt): the slice phase takes its ~20 k cycles either way.
#ifdef COMMS_DIAG
    const int pmode = fault >> 1;
    const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
    auto prio = [&](int step) {
        if (pmode == 0) return;
        const int p = pmode == 1 ? ((step + grp) & 3) : pmode == 2 ? grp : pmode == 3 ? 3 - grp : ((step - grp) & 3);
        switch (p) {
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
        }
    };
#else
    auto prio = [](int) {};
#endif
    if (!accumulate) hist_advance(hist, in, n, new_hist, hist_len);
    tw1[tid] = tb.tw1[tid];
    if (tid < 64) tw2[tid] = tb.tw2[tid];
    if (tid < 32) slice_in[tid] = 0;
    // The stage-1 twiddle W16384^{t*k0} of point t = 64 a + l of slice k0 is applied by the wave that OWNS
    // slice k0 (k0 = wave), where it factors into W256^{a*k0} -- wave-uniform, indexed by the register: sixteen
    // SGPR pairs read through the scalar cache -- and W16384^{l*k0}, one per-lane constant.  (Applied by the
    // producers it is fifteen per-lane values per thread: two LDS table reads and a product per point.)
    typedef const __attribute__((address_space(4))) cf* const_cf_ptr;  // constant address space: scalar loads
    const_cf_ptr sta_p = (const_cf_ptr)(tb.ta + 16 * __builtin_amdgcn_readfirstlane(wave));
    cf sta[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) sta[a] = sta_p[a];
    const cf lane_tw = tb.tb[wave * 64 + l];
    __syncthreads();  // tables and counters: the only barrier of the launch

    // workgroup g of the persistent grid owns segments [g*nseg/G, (g+1)*nseg/G).  (Round-robin chunks of 1 ... 16 segments,
    // what the 1024-point ticketed kernel gains from, LOSE here: 625 -> 650 ... 696 us at 2^27 samples -- a chunk's first
    // segment reads its four halo rows again, and sixteen rows of 8 KiB per segment are long runs already;
    // profiles/r04_probe_chunks.txt.)
    const size_t seg_lo = static_cast<size_t>(blockIdx.x) * nseg / gridDim.x;
    const size_t seg_hi = static_cast<size_t>(blockIdx.x + 1) * nseg / gridDim.x;
    // The new rows of a segment are requested one phase ahead -- between the slice work and the inverse stage 1 of the
    // segment before, where the registers are free -- so their HBM latency runs behind the workgroup's second
    // wait instead of in front of an idle CU.
    const int R0 = COMMS_OS16K_CARRY ? hr : 0;  // first row that is fetched per segment
    cf v[16], rows[16];
    auto fetch_rows = [&](size_t sg, int first) {
        const long long base = static_cast<long long>(sg * xv) - 1024 * hr - delay;
        if (base >= 0) {  // no history involved: buffer loads, zeros past the end of the stream
            const BufRows<In> br(in, static_cast<size_t>(base), static_cast<size_t>(base) < n ? n - static_cast<size_t>(base) : 0);
#pragma unroll
            for (int a = 0; a < 16; ++a)
                if (a >= first) rows[a] = to_cf(br.get(tid * BufRows<In>::E, a * 1024 * BufRows<In>::E));
        } else {
#pragma unroll
            for (int a = 0; a < 16; ++a)
                if (a >= first) rows[a] = to_cf(stream_at(in, hist, hist_len, base + 1024 * a + tid, n));
        }
    };
    if (seg_lo < seg_hi) fetch_rows(seg_lo, 0);
    unsigned done = 0;  // segments this workgroup has finished
    // DEC: output index of the thread's first row = rate x kq + kr, carried from segment to segment
    unsigned long long kq = 0;
    unsigned kr = 0;
    if constexpr (DEC) {
        const unsigned long long o_first = static_cast<unsigned long long>(seg_lo) * xv + tid;
        kq = o_first / dec.rate;
        kr = static_cast<unsigned>(o_first - kq * dec.rate);
    }
#ifdef COMMS_DIAG
    unsigned long long* xt = g_x_trace;
    if (xt) xt += (static_cast<size_t>(blockIdx.x) * 16 + wave) * X_TRACE_SEGS * 16;
#endif
    for (size_t seg = seg_lo; seg < seg_hi; ++seg, ++done) {
        const size_t nb = seg * xv;
        X_MARK(0);
        prio(0);
        // ---- stage 1: radix-16 over the rows; value k of thread tid is point tid of slice k.  No wait in front
        // of the writes: word (k, tid) was last read by this very thread (inverse stage 1 of the previous
        // segment), and wave k is past its slice work for that segment or nobody could have read it.
#pragma unroll
        for (int a = 0; a < 16; ++a) v[a] = rows[a];
#ifdef COMMS_DIAG
        if (xt) {  // the rows have arrived (behind them in the queue: the 16 - HR stores of the segment before)
            if (done == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(16 - HR) : "memory");
            X_MARK(1);
        }
#endif
#if COMMS_OS16K_CARRY
        // the next segment's halo = this segment's last hr rows (HBM would see them again otherwise: the XCD streams
        // 7 MiB per segment time through a 4 MiB L2)
#pragma unroll
        for (int a = 0; a < HR; ++a) rows[a] = rows[16 - HR + a];
#endif
        radix16<-1>(v);
        X_MARK_AFTER(8, v[R16_POS(15)].x);
#pragma unroll
        for (int k = 0; k < 16; ++k) bufs[k * X_BUF + tid] = v[R16_POS(k)];
        x_signal(slice_in + (l & 15), l < 16);  // one ds_add, sixteen lanes, sixteen counters
        X_MARK(2);
        // ---- this wave's slice, once all sixteen waves have delivered their 64 points of it: stage-1 twiddle,
        // 1024-point transform, spectrum multiply, inverse, conjugate twiddle
        if (!x_wait_one(slice_in + wave, 16u * (done + 1), gave_up)) break;
        X_MARK(3);
        prio(1);
#pragma unroll
        for (int a = 0; a < 16; ++a) v[a] = buf[64 * a + l];
        wave_lds_sync();
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            if (a) v[a] = cmulf_s(v[a], sta[a]);
            v[a] = cmulf(v[a], lane_tw);
        }
        os1024_core<1024>(v, buf, tw1, tb.hdev, tw2, l, [&](int i) { prio(1 + i); }, tid);
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            cf y = cmulcf(v[R16_POS(a)], lane_tw);
            if (a) y = cmulcf_s(y, sta[a]);
            buf[64 * a + l] = y;
        }
        x_signal(slice_out + wave, l == 0 && !(withhold && done == 0));
        X_MARK(4);
        // (unconditional -- the last segment fetches itself again -- so that `rows` is redefined on every path
        // and its registers are free during the slice work)
        fetch_rows(seg + 1 < seg_hi ? seg + 1 : seg, R0);
        // ---- inverse stage 1: thread tid gathers point tid of every slice, radix-16 back to the rows
        X_MARK(5);
        if (!x_wait_all16(slice_out, done + 1, l, gave_up)) break;
        X_MARK(6);
        prio(10);
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = bufs[k * X_BUF + tid];
#ifdef COMMS_DIAG
        if (xt) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            X_MARK(9);
        }
#endif
        radix16<1>(v);
        X_MARK_AFTER(10, v[R16_POS(15)].x);
        if constexpr (DEC) {
            const size_t obase = nb + tid;
            const uint64_t tl = dec.turns0 + static_cast<uint64_t>(obase) * dec.frac;
            // e^{2 pi i m / 64} from the stage table W1024^{8 x 2m}, m < 32; the other half by symmetry
            const unsigned u = static_cast<unsigned>(tl >> 32), m = u >> 26;
            const cf w = tw1[8 * 64 + 2 * (m & 31u)];
            const cf tq = m < 32u ? cf{w.x, -w.y} : cf{-w.x, w.y};
            const float th = static_cast<float>(u & 0x3FFFFFFu) * 1.4629180792671596e-09f;  // 2 pi / 2^32
            const float z = th * th;
            float sp = __builtin_fmaf(z, 8.3333333e-3f, -1.6666667e-1f);
            sp = __builtin_fmaf(z, sp, 1.0f);
            float cp = __builtin_fmaf(z, -1.3888889e-3f, 4.1666667e-2f);
            cp = __builtin_fmaf(z, cp, -0.5f);
            const cf rot0 = cmulf(tq, cf{__builtin_fmaf(z, cp, 1.0f), th * sp});
            float2* outq = out + kq;
            unsigned q = 0, r = kr;
            const int nrows = obase < n ? static_cast<int>((n - obase + 1023) >> 10 < 16 ? (n - obase + 1023) >> 10 : 16) : 0;
#pragma unroll
            for (int a = 1; a < 16; ++a) {
                if (a < hr) continue;
                if (r == 0 && a - hr < nrows) {
                    const cf rot = cmulf(rot0, to_cf(dec.step[a - hr]));
                    outq[q] = to_f2(cmulf(v[R16_POS(a)], rot));
                }
                r += dec.c1;
                const unsigned wrap = r >= dec.rate ? 1u : 0u;
                r -= wrap ? dec.rate : 0u;
                q += dec.d1 + wrap;
                __builtin_amdgcn_sched_barrier(0);
            }
            kr += dec.dr;
            const unsigned wrap = kr >= dec.rate ? 1u : 0u;
            kr -= wrap ? dec.rate : 0u;
            kq += dec.dq + wrap;
            X_MARK(7);
            continue;
        }
        // stores (and the accumulating pass's loads) through a buffer resource that ends at sample n: nothing past it
        const __amdgpu_buffer_rsrc_t ors = make_rsrc(out + nb, nb < n ? (n - nb) * sizeof(float2) : 0);
#pragma unroll
        for (int a = 1; a < 16; ++a) {
            if (a < hr) continue;  // (workgroup-uniform)
            const unsigned row = static_cast<unsigned>(a - hr) * 8192u;
            cf y = v[R16_POS(a)];
            if (accumulate) y = y + to_cf(BufRows<const float2*>::get_from(ors, tid * 8, row));
            __builtin_amdgcn_raw_buffer_store_b64(bv2u{__float_as_uint(y.x), __float_as_uint(y.y)}, ors, tid * 8, row, 0);
        }
        X_MARK(7);
    }
    x_report(aux);
#ifndef COMMS_OS16K_NO_STAMP
    kstamp_end(KStamp{nullptr, reinterpret_cast<unsigned long long*>(aux[0])});
#endif
}

// ---------------------------------------------------------------- pulse shaping (polyphase)
// Reference: PulseNode::run (src/pulse.rs:82-92) = zero-stuff by sps, then FIR.
//   out[m*sps + p] = sum_j taps[p + j*sps] * sym[m - j]
// Output mixer fused into the pulse kernels (comms_pulse_set_mixer): out[i] *= rot(turns0 + i*frac),
// the MixerNode that follows the PulseNode in a transmit chain (BASELINE config 1).  Phases are the
// mixer node's 64-bit turns; a lane evaluates its first rotor with one f64 sincos and steps it per
// grid sweep with a constant f64 rotor; inside a symbol the sps outputs use f32 step rotors.
struct PulseMix {
    uint64_t turns0, frac;
    double sweep_c, sweep_s;  // rot(outputs per grid sweep * frac)
    int on;
    int out_i16;              // store `(out_scale * y) as i16` pairs instead of Complex<f32> (comms_pulse_set_output_format)
    float out_scale;
};
__device__ __forceinline__ void pulse_rotor_at(uint64_t turns, double& c, double& s) {
    sincos(static_cast<double>(turns >> 11) * (kTwoPiF * 0x1.0p-53), &s, &c);
}

__global__ __launch_bounds__(256) void pulse_kernel(const float2* __restrict__ sym,
                                                    const float2* __restrict__ hist, int hist_len,
                                                    const float2* __restrict__ taps, int n_taps,
                                                    int sps, float2* __restrict__ out,
                                                    size_t n_sym, float2* __restrict__ new_hist, PulseMix mx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    hist_advance(hist, sym, n_sym, new_hist, hist_len);
    float2* tp = reinterpret_cast<float2*>(smem);
    for (int k = threadIdx.x; k < n_taps; k += 256) tp[k] = taps[k];
    __syncthreads();
    const size_t n_out = n_sym * static_cast<size_t>(sps);
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    double rc = 1.0, rs = 0.0;
    if (mx.on) pulse_rotor_at(mx.turns0 + (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) * mx.frac, rc, rs);
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n_out;
         i += stride) {
        const size_t m = i / sps;
        const int p = static_cast<int>(i - m * sps);
        float2 acc = make_float2(0.f, 0.f);
        long long s = static_cast<long long>(m);
        for (int k = p; k < n_taps; k += sps, --s) {
            const float2 x = stream_at(sym, hist, hist_len, s, n_sym);
            const float2 h = tp[k];
            acc.x = __builtin_fmaf(h.x, x.x, acc.x);
            acc.x = __builtin_fmaf(-h.y, x.y, acc.x);
            acc.y = __builtin_fmaf(h.x, x.y, acc.y);
            acc.y = __builtin_fmaf(h.y, x.x, acc.y);
        }
        if (mx.on) {  // Mixer::mix arithmetic: f64 product rounded once (src/mixer.rs:77-78)
            const double yr = acc.x, yi = acc.y;
            acc = make_float2(static_cast<float>(yr * rc - yi * rs), static_cast<float>(yr * rs + yi * rc));
            const double nc = rc * mx.sweep_c - rs * mx.sweep_s;
            rs = rc * mx.sweep_s + rs * mx.sweep_c;
            rc = nc;
        }
        if (mx.out_i16)
            reinterpret_cast<short2*>(out)[i] = c32_as_i16(acc, mx.out_scale);
        else
            out[i] = acc;
    }
}

// Polyphase pulse shaper for the usual sam_per_sym (2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 32):
//   out[m*SPS + p] = sum_j taps[p + j*SPS] * sym[m - j]          (k = p + j*SPS ascending, as fir())
// A lane owns one symbol m and its SPS outputs: each symbol of the window (from LDS, consecutive
// lanes -> consecutive addresses) feeds SPS packed FMAs whose taps sit in SGPR pairs (kernel
// arguments, rows of SPS taps padded to an even count), and the lane stores its SPS outputs as
// one contiguous run.  2/SPS + 8 bytes of HBM traffic per output; the generic pulse_kernel
// (any SPS, any tap count) stays as the fallback.
constexpr int PP_AMAX = 384;  // floats per tap array (rows * padded row)
constexpr int PP_JB = 4;      // tap rows per loop iteration
constexpr int PP_JMAX = 128;  // rows: bounds the LDS window (256 + PP_JMAX symbols)
struct PulseArgs {
    const float2* sym;
    const float2* hist;
    float2* out;
    float2* new_hist;
    size_t n_sym;
    PulseMix mx;           // sweep = rot(gridDim.x * 256 * SPS * frac)
    float2 step[32];       // rot(p * frac), p < SPS: the outputs of one symbol
    int hist_len, J;       // J rows of taps (multiple of PP_JB, zero rows appended)
    float are[PP_AMAX];    // A[j*SPSP + p] = Re taps[p + j*SPS]
    float aim[PP_AMAX];
    KStamp ks;             // in-kernel begin / end stamps of a stamps timer, or null
};

template <int SPS, bool REAL, bool MIX>
__global__ __launch_bounds__(256) void pulse_poly_kernel(const PulseArgs a) {
    constexpr int SPSP = SPS + (SPS & 1);
    // outputs leave through a per-wave LDS block, CW phases at a time (CW * 2 KiB per workgroup)
    constexpr int CW = SPS <= 8 ? SPS : SPS % 8 == 0 ? 8 : SPS % 6 == 0 ? 6 : SPS % 5 == 0 ? 5 : SPS % 4 == 0 ? 4 : SPS % 3 == 0 ? 3 : SPS % 2 == 0 ? 2 : 1;
    __shared__ cf sh[256 + PP_JMAX];
    __shared__ __attribute__((aligned(16))) cf xch[256 * CW];
    const int tid = threadIdx.x;
    const int halo = a.J - 1;
    const size_t ntiles = (a.n_sym + 255) / 256;
    kstamp_begin(a.ks);
    hist_advance(a.hist, a.sym, a.n_sym, a.new_hist, a.hist_len);
    double rc = 1.0, rs = 0.0;  // rotor of this lane's first output of the current tile
    if (MIX) pulse_rotor_at(a.mx.turns0 + (static_cast<uint64_t>(blockIdx.x) * 256 + tid) * SPS * a.mx.frac, rc, rs);
    // the next tile's symbols are requested before this tile's taps run and land in LDS at the top of the next step: a
    // tile's own work is short, and without this every step began with an exposed round trip to HBM
    cf nx0 = cf{0.f, 0.f}, nx1 = cf{0.f, 0.f};  // window elements tid and 256 + tid (the latter: tid < halo <= 127)
    auto fetch = [&](size_t t) {
        const long long w0 = static_cast<long long>(t) * 256 - halo;
        nx0 = to_cf(stream_at(a.sym, a.hist, a.hist_len, w0 + tid, a.n_sym));
        if (tid < halo) nx1 = to_cf(stream_at(a.sym, a.hist, a.hist_len, w0 + 256 + tid, a.n_sym));
    };
    if (blockIdx.x < ntiles) fetch(blockIdx.x);
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const long long m0 = static_cast<long long>(t) * 256;
        __syncthreads();
        sh[tid] = nx0;
        if (tid < halo) sh[256 + tid] = nx1;
        __syncthreads();
        if (t + gridDim.x < ntiles) fetch(t + gridDim.x);
        cf acc[SPS];
#pragma unroll
        for (int p = 0; p < SPS; ++p) acc[p] = cf{0.f, 0.f};
        const cf* sp = sh + halo + tid;
        for (int j0 = 0; j0 < a.J; j0 += PP_JB) {
            cf sv[PP_JB];
            v2f tr[PP_JB * SPSP / 2], ti[PP_JB * SPSP / 2];
#pragma unroll
            for (int i = 0; i < PP_JB * SPSP / 2; ++i) {
                tr[i] = v2f{a.are[j0 * SPSP + 2 * i], a.are[j0 * SPSP + 2 * i + 1]};
                if (!REAL) ti[i] = v2f{a.aim[j0 * SPSP + 2 * i], a.aim[j0 * SPSP + 2 * i + 1]};
            }
#pragma unroll
            for (int jj = 0; jj < PP_JB; ++jj) sv[jj] = sp[-(j0 + jj)];
#pragma unroll
            for (int jj = 0; jj < PP_JB; ++jj)
#pragma unroll
                for (int p = 0; p < SPS; ++p) {
                    const int e = jj * SPSP + p;
                    mac_tap<REAL>(acc[p], sv[jj], tr[e >> 1], ti[e >> 1], (e & 1) != 0);
                }
        }
        if (MIX) {
            const cf r0 = cf{static_cast<float>(rc), static_cast<float>(rs)};
#pragma unroll
            for (int p = 0; p < SPS; ++p) acc[p] = cmulf(acc[p], p ? cmulf(r0, to_cf(a.step[p])) : r0);
            const double nc = rc * a.mx.sweep_c - rs * a.mx.sweep_s;
            rs = rc * a.mx.sweep_s + rs * a.mx.sweep_c;
            rc = nc;
        }
        {
            // A lane's SPS outputs are one run of SPS * 8 B, so a plain store instruction covers a wave's 64 SPS outputs in
            // pieces of 16 B at a stride of SPS * 8 B.  Through the wave's own LDS block instead, CW <= 8 phases at a time:
            // instruction i writes elements 64 i ... 64 i + 63 of the block [64 symbols][CW], i.e. whole lines for SPS <= 8
            // and runs of CW * 8 B above (63 taps x 4, 2^26 outputs: 183 -> 134 us; 127 taps x 8, 2^24: 48.8 -> 29.7 us).
            cf* ex = xch + (tid & ~63) * CW;
            const int lane = tid & 63;
            const size_t mw = static_cast<size_t>(m0) + (tid & ~63);                      // the wave's first symbol
            const unsigned nel = mw < a.n_sym ? static_cast<unsigned>(a.n_sym - mw < 64 ? a.n_sym - mw : 64) * CW : 0u;  // valid elements per chunk
#pragma unroll
            for (int c = 0; c < SPS / CW; ++c) {
                if (c) {  // the previous chunk has been read
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
#pragma unroll
                for (int p = 0; p < CW; ++p) ex[lane * CW + p] = acc[c * CW + p];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (a.mx.out_i16) {  // the transmit chain straight into the IQOutput wire format: 4 B per output
                    constexpr int W = CW % 4 == 0 ? 4 : CW % 2 == 0 ? 2 : 1;  // outputs per lane and store
                    short2* o = reinterpret_cast<short2*>(a.out) + mw * SPS + c * CW;
#pragma unroll
                    for (int i = 0; i < CW / W; ++i) {
                        const unsigned e = (static_cast<unsigned>(i) * 64u + lane) * W;
                        if (e < nel) {  // (nel is a multiple of CW, hence of W)
                            short2 q[W];
#pragma unroll
                            for (int w = 0; w < W; ++w) q[w] = c32_as_i16(to_f2(ex[e + w]), a.mx.out_scale);
                            short2* dp = o + (e / CW) * SPS + e % CW;
                            if constexpr (W == 4) {
                                u32x4 qv;
                                __builtin_memcpy(&qv, q, 16);
                                store_b128_dword_aligned(dp, qv);
                            } else if constexpr (W == 2) {
                                __builtin_memcpy(dp, q, 8);
                            } else {
                                dp[0] = q[0];
                            }
                        }
                    }
                } else {
                    constexpr int W = CW % 2 == 0 ? 2 : 1;
                    float2* o = a.out + mw * SPS + c * CW;
#pragma unroll
                    for (int i = 0; i < CW / W; ++i) {
                        const unsigned e = (static_cast<unsigned>(i) * 64u + lane) * W;
                        if (e < nel) {
                            float2* dp = o + (e / CW) * SPS + e % CW;
                            if constexpr (W == 2) {
                                const cf u0 = ex[e], u1 = ex[e + 1];
                                float2 q[2] = {to_f2(u0), to_f2(u1)};
                                __builtin_memcpy(dp, q, 16);
                            } else {
                                dp[0] = to_f2(ex[e]);
                            }
                        }
                    }
                }
            }
        }
    }
    kstamp_end(a.ks);
}

}  // namespace comms

using namespace comms;

// ================================================================= FIR handle (struct comms_fir: fir_handle.hpp)
static void free_fir(comms_fir* h) {
    h->conv.release();
    if (h->d_qt) (void)hipFree(h->d_qt);
    if (h->d_any_taps) (void)hipFree(h->d_any_taps);
    if (h->d_p8) (void)hipFree(h->d_p8);
    (void)use_device(h->device);
    if (h->d_taps_pad) (void)hipFree(h->d_taps_pad);
    if (h->d_wtw1) (void)hipFree(h->d_wtw1);
    if (h->d_wtw2) (void)hipFree(h->d_wtw2);
    if (h->d_whdev) (void)hipFree(h->d_whdev);
    if (h->d_tw1) (void)hipFree(h->d_tw1);
    if (h->d_tw2) (void)hipFree(h->d_tw2);
    for (float2* q : h->d_hparts)
        if (q) (void)hipFree(q);
    for (float2* q : h->d_xh)
        if (q) (void)hipFree(q);
    for (float2* q : h->d_xt)
        if (q) (void)hipFree(q);
    if (h->d_hist[0]) (void)hipFree(h->d_hist[0]);
    if (h->d_hist[1]) (void)hipFree(h->d_hist[1]);
    if (h->err_host) (void)hipHostFree(h->err_host);
    h->fini();
    delete h;
}

// One launch of fir_os1024_kernel<.., MODE>: 16-wave workgroups (one per CU, 156 KiB of
// LDS: shared tables + 16 private exchange buffers) by default, 4-wave workgroups
// (three per CU) with COMMS_OS1024_WPB=4.
template <int MODE, int HR = 4, class In = const float2*>
static comms_status_t launch_os1024(int wpb, size_t runs, hipStream_t s, In in, const float2* hist,
                                    int n_eff, float2* o, size_t n, size_t nseg, const comms::WTables& tb,
                                    float2* nh, const comms::ChainArgs& ch, hipEvent_t ev_start = nullptr,
                                    hipEvent_t ev_stop = nullptr) {
    using namespace comms;
    if (wpb == 16) {
        const size_t lds = (2112 + 16 * W_LDS) * sizeof(float2);
        static DeviceOnce attr_once;
        if (attr_once.need()) {
            COMMS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fir_os1024_kernel<16, 4, MODE, HR, In>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        }
        if (ev_start) {  // timed launch: the events take the kernel's own begin / end timestamps
            hipExtLaunchKernelGGL((fir_os1024_kernel<16, 4, MODE, HR, In>), dim3(static_cast<unsigned>((runs + 15) / 16)),
                                  dim3(1024), static_cast<uint32_t>(lds), s, ev_start, ev_stop, 0u, in, hist, n_eff, o, n,
                                  nseg, runs, tb, nh, ch);
            return COMMS_OK;
        }
        fir_os1024_kernel<16, 4, MODE, HR, In><<<dim3(static_cast<unsigned>((runs + 15) / 16)), dim3(1024), lds, s>>>(
            in, hist, n_eff, o, n, nseg, runs, tb, nh, ch);
    } else {
        const size_t lds = (2112 + 4 * W_LDS) * sizeof(float2);
        fir_os1024_kernel<4, 3, MODE, HR, In><<<dim3(static_cast<unsigned>((runs + 3) / 4)), dim3(256), lds, s>>>(
            in, hist, n_eff, o, n, nseg, runs, tb, nh, ch);
    }
    return COMMS_OK;
}
static size_t os1024_runs(int wpb, size_t nseg, size_t min_run) {
    // persistent: every wave slot of the chip gets one run (fewer for short inputs, where a
    // run is at least min_run segments to amortise its halo load)
    size_t runs = static_cast<size_t>(wpb == 16 ? 16 : 12) * comms::kNumCU;
    if (runs * min_run > nseg) runs = (nseg + min_run - 1) / min_run;
    return runs;
}

// 0: fixed runs (fir_os1024_kernel), 1: ticketed segments (fir_os1024_dyn_kernel).  COMMS_OS1024_DYNAMIC
// sets it; comms_debug_os1024_dynamic() switches it at run time so that scripts/ab_fir.py can interleave
// the variants launch by launch.
static int tune_int(const char* name, int dflt);
static std::atomic<int> g_os1024_dynamic{-1};
static int os1024_dynamic_mode() {
    int m = g_os1024_dynamic.load(std::memory_order_relaxed);
    if (m < 0) {
        m = tune_int("COMMS_OS1024_DYNAMIC", 1);
        g_os1024_dynamic.store(m, std::memory_order_relaxed);
    }
    return m;
}
#ifdef COMMS_DIAG
extern "C" void comms_debug_os1024_dynamic(int mode) { g_os1024_dynamic.store(mode, std::memory_order_relaxed); }
#endif

// How the ticketed kernel deals its segments to the workgroups: chunks of 2^k segments round-robin (k < 32) or one
// contiguous share each (32).  Measured with the variants interleaved launch by launch (scripts/probe_chunks.py,
// profiles/r04_probe_chunks.txt; 255 taps): 2^24 samples 52.9 -> 51.0 us with chunks of 2 (and 57.1 -> 50.4 where the
// output buffer sat 1 MiB further from the input: shares that start a power of two apart make the time depend on the
// buffers' relative placement, chunks do not), 2^26 neutral, 2^28 875 -> 809 us with chunks of 8, 2^30 3.236 -> 3.213 ms.
// Chunks of 2 keep the workgroups' loads within one segment of each other at 2^24 (85 segments each); from 16 up the
// last round's imbalance shows (2^24: 55.9 us), at 64 everywhere.  COMMS_OS1024_CHUNK_LOG2 overrides.
static std::atomic<int> g_os1024_chunk{-2};
static unsigned os1024_chunk_log2(size_t nseg, unsigned grid) {
    int env = g_os1024_chunk.load(std::memory_order_relaxed);
    if (env == -2) {
        env = tune_int("COMMS_OS1024_CHUNK_LOG2", -1);
        g_os1024_chunk.store(env, std::memory_order_relaxed);
    }
    if (env >= 0) return static_cast<unsigned>(env);
    return nseg < 160u * static_cast<size_t>(grid) ? 1u : 3u;  // (160 segments per workgroup: 2^25 samples at 255 taps)
}
#ifdef COMMS_DIAG
extern "C" void comms_debug_os1024_chunk_log2(int k) { g_os1024_chunk.store(k, std::memory_order_relaxed); }
#endif

// One launch of fir_os1024_dyn_kernel: one 16-wave workgroup per CU (fewer for short inputs).
template <int HR, bool TRACE = false, class In = const float2*>
static comms_status_t launch_os1024_dyn(hipStream_t s, In in, const float2* hist, int n_eff, float2* o,
                                        size_t n, const comms::WTables& tb, float2* nh, void* trace_buf = nullptr,
                                        hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr,
                                        KStamp ks = KStamp{nullptr, nullptr}) {
    using namespace comms;
    const size_t lds = (2112 + 16 * W_LDS + 1) * sizeof(float2);  // tables, exchange buffers, ticket counter
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        COMMS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fir_os1024_dyn_kernel<HR, TRACE, In>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    }
    const size_t nseg = (n + (1024 - 64 * HR) - 1) / (1024 - 64 * HR);
    const size_t want = (nseg + 15) / 16;
    const dim3 grid(static_cast<unsigned>(want < static_cast<size_t>(kNumCU) ? want : kNumCU));
    const unsigned chunk_log2 = os1024_chunk_log2(nseg, grid.x);
    if (ev_start)  // timed launch: the events take the kernel's own begin / end timestamps
        hipExtLaunchKernelGGL((fir_os1024_dyn_kernel<HR, TRACE, In>), grid, dim3(1024), static_cast<uint32_t>(lds), s, ev_start,
                              ev_stop, 0u, in, hist, n_eff, o, n, tb, nh, trace_buf, chunk_log2, ks);
    else
        fir_os1024_dyn_kernel<HR, TRACE, In><<<grid, dim3(1024), lds, s>>>(in, hist, n_eff, o, n, tb, nh, trace_buf, chunk_log2, ks);
    return COMMS_OK;
}

static const double kPi = 3.14159265358979323846264338327950288;

static float2 unit_root_os(long long e, int denom) {
    e %= denom;
    const double a = -2.0 * kPi * static_cast<double>(e) / denom;
    return make_float2(static_cast<float>(std::cos(a)), static_cast<float>(std::sin(a)));
}

// Tuning knobs (scripts/bench_fir.py sweeps them in the diagnostic build; compile-time defaults in the product: common.hpp)
static int tune_int(const char* name, int dflt) { return diag_knob(name, dflt); }

// Filter spectrum + twiddle tables for the 4096-point overlap-save kernel (f64 on
// the host, rounded once to f32).
constexpr int OS_PART = 2049;  // taps per partition of a long filter: halo 2048, 2048 outputs per segment

static void tap_spectrum_range(const comms_fir* h, int first, int count, int F, std::vector<double>& re,
                               std::vector<double>& im);

static comms_status_t fir_prepare_os(comms_fir* h) {
    if (h->os_ready) return COMMS_OK;
    const int N = h->n_eff;
    int per = N;
    h->n_part = 1;
    if (N > 3841) {  // partitioned convolution: y = sum_p FIR(taps[p*P .. ), x delayed by p*P)
        per = OS_PART;
        h->n_part = (N + OS_PART - 1) / OS_PART;
    }
    h->hblk = (per - 1 + 255) / 256;
    if (h->hblk < 1) h->hblk = 1;
    std::vector<float2> tw1(16 * 256), tw2(16 * 16), hdev(16 * 256);
    for (int k0 = 0; k0 < 16; ++k0)
        for (int t = 0; t < 256; ++t) tw1[k0 * 256 + t] = unit_root_os(static_cast<long long>(t) * k0, OSF);
    for (int j = 0; j < 16; ++j)
        for (int lo = 0; lo < 16; ++lo) tw2[j * 16 + lo] = unit_root_os(lo * j, 256);
    COMMS_HIP_TRY(hipMalloc(&h->d_tw1, tw1.size() * sizeof(float2)));
    COMMS_HIP_TRY(hipMalloc(&h->d_tw2, tw2.size() * sizeof(float2)));
    COMMS_HIP_TRY(hipMemcpy(h->d_tw1, tw1.data(), tw1.size() * sizeof(float2), hipMemcpyHostToDevice));
    COMMS_HIP_TRY(hipMemcpy(h->d_tw2, tw2.data(), tw2.size() * sizeof(float2), hipMemcpyHostToDevice));
    std::vector<double> re, im;
    for (int pt = 0; pt < h->n_part; ++pt) {
        const int first = pt * per;
        const int count = N - first < per ? N - first : per;
        tap_spectrum_range(h, first, count, OSF, re, im);
        for (int k = 0; k < OSF; ++k) {
            const int k0 = k & 15, k1 = (k >> 4) & 15, k2 = k >> 8;
            hdev[k2 * 256 + 16 * k0 + k1] =
                make_float2(static_cast<float>(re[k] / OSF), static_cast<float>(im[k] / OSF));
        }
        float2* d = nullptr;
        COMMS_HIP_TRY(hipMalloc(&d, hdev.size() * sizeof(float2)));
        h->d_hparts.push_back(d);
        COMMS_HIP_TRY(hipMemcpy(d, hdev.data(), hdev.size() * sizeof(float2), hipMemcpyHostToDevice));
    }
    h->d_hdev = h->d_hparts[0];
    h->os_ready = true;
    return COMMS_OK;
}

// Spectrum of taps[first, first+count) zero-padded to F points (F a power of two), f64:
// iterative radix-2 FFT with twiddles taken straight from cos/sin per index.
static void tap_spectrum_range(const comms_fir* h, int first, int count, int F, std::vector<double>& re,
                               std::vector<double>& im) {
    re.assign(F, 0.0);
    im.assign(F, 0.0);
    for (int j = 0; j < count; ++j) {
        re[j] = h->taps[first + j].re;
        im[j] = h->taps[first + j].im;
    }
    for (int i = 1, j = 0; i < F; ++i) {  // bit reversal
        int bit = F >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            std::swap(re[i], re[j]);
            std::swap(im[i], im[j]);
        }
    }
    std::vector<double> cs(F / 2), sn(F / 2);
    for (int e = 0; e < F / 2; ++e) {
        const double a = -2.0 * kPi * static_cast<double>(e) / F;
        cs[e] = std::cos(a);
        sn[e] = std::sin(a);
    }
    for (int len = 2; len <= F; len <<= 1) {
        const int half = len / 2, step = F / len;
        for (int i = 0; i < F; i += len)
            for (int k = 0; k < half; ++k) {
                const double wr = cs[k * step], wi = sn[k * step];
                const double xr = re[i + k + half], xi = im[i + k + half];
                const double vr = xr * wr - xi * wi, vi = xr * wi + xi * wr;
                re[i + k + half] = re[i + k] - vr;
                im[i + k + half] = im[i + k] - vi;
                re[i + k] += vr;
                im[i + k] += vi;
            }
    }
}

static void tap_spectrum(const comms_fir* h, int F, std::vector<double>& re, std::vector<double>& im) {
    tap_spectrum_range(h, 0, h->n_eff, F, re, im);
}

static comms_status_t upload_f2(const std::vector<float2>& v, float2** d) {
    COMMS_HIP_TRY(hipMalloc(d, v.size() * sizeof(float2)));
    COMMS_HIP_TRY(hipMemcpy(*d, v.data(), v.size() * sizeof(float2), hipMemcpyHostToDevice));
    return COMMS_OK;
}

static float2 unit_root(long long e, int denom) {
    e %= denom;
    double a = -2.0 * kPi * static_cast<double>(e) / denom;
    return make_float2(static_cast<float>(std::cos(a)), static_cast<float>(std::sin(a)));
}

static comms_status_t fir_prepare_os1024(comms_fir* h) {
    if (h->w_ready) return COMMS_OK;
    COMMS_ARG(h->n_eff <= 257, "the 1024-point overlap-save kernel supports at most 257 taps, got %d", h->n_eff);
    std::vector<float2> tw1(16 * 64), tw2(16 * 4), hdev(16 * 64);
    for (int k0 = 0; k0 < 16; ++k0)
        for (int t = 0; t < 64; ++t) tw1[k0 * 64 + t] = unit_root(static_cast<long long>(t) * k0, 1024);
    for (int k1 = 0; k1 < 16; ++k1)
        for (int c = 0; c < 4; ++c) tw2[k1 * 4 + c] = unit_root(c * k1, 64);
    std::vector<double> re, im;
    tap_spectrum(h, WF, re, im);
    // register 4t + m of lane k0 + 16g + 32h holds Z[k0 + 16 (2m + 8g + h) + 256 t] (os1024_core, stage 3)
    for (int t = 0; t < 4; ++t)
        for (int m = 0; m < 4; ++m)
            for (int l = 0; l < 64; ++l) {
                const int k0 = l & 15, g = (l >> 4) & 1, hh = l >> 5;
                const int k = k0 + 16 * (2 * m + 8 * g + hh) + 256 * t;
                hdev[(4 * t + m) * 64 + l] =
                    make_float2(static_cast<float>(re[k] / WF), static_cast<float>(im[k] / WF));
            }
    COMMS_TRY(upload_f2(tw1, &h->d_wtw1));
    COMMS_TRY(upload_f2(tw2, &h->d_wtw2));
    COMMS_TRY(upload_f2(hdev, &h->d_whdev));
    h->w_ready = true;
    return COMMS_OK;
}

constexpr int X_PART = 4097;  // taps per pass of the 16384-point kernel (halo 4096)

static comms_status_t fir_prepare_os16k(comms_fir* h) {
    if (h->x_ready) return COMMS_OK;
    if (!h->err_host) {  // sticky error word: pinned, coherent, mapped -- the kernel raises it, the host reads it without a sync
        COMMS_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->err_host), 64, hipHostMallocMapped | hipHostMallocCoherent));
        *h->err_host = 0;
        void* d = nullptr;
        COMMS_HIP_TRY(hipHostGetDevicePointer(&d, h->err_host, 0));
        h->d_err = static_cast<unsigned*>(d);
    }
    const int N = h->n_eff;
    h->x_part = (N + X_PART - 1) / X_PART;
    std::vector<float2> tw1(16 * 64), tw2(16 * 4), ta(16 * 16), tb(16 * 64), hdev(16 * 1024);
    for (int k = 0; k < 16; ++k)
        for (int t = 0; t < 64; ++t) {
            tw1[k * 64 + t] = unit_root(static_cast<long long>(t) * k, 1024);
            tb[k * 64 + t] = unit_root(static_cast<long long>(t) * k, XF);
        }
    for (int k1 = 0; k1 < 16; ++k1)
        for (int c = 0; c < 4; ++c) tw2[k1 * 4 + c] = unit_root(c * k1, 64);
    for (int w = 0; w < 16; ++w)
        for (int k = 0; k < 16; ++k) ta[w * 16 + k] = unit_root(w * k, 256);
    COMMS_TRY(upload_f2(tw1, &h->d_xt[0]));
    COMMS_TRY(upload_f2(tw2, &h->d_xt[1]));
    COMMS_TRY(upload_f2(ta, &h->d_xt[2]));
    COMMS_TRY(upload_f2(tb, &h->d_xt[3]));
    std::vector<double> re, im;
    for (int pt = 0; pt < h->x_part; ++pt) {
        const int first = pt * X_PART;
        const int count = N - first < X_PART ? N - first : X_PART;
        tap_spectrum_range(h, first, count, XF, re, im);
        for (int tid = 0; tid < 1024; ++tid) {
            const int k0 = tid >> 6, l = tid & 63;
            for (int t = 0; t < 4; ++t)
                for (int m = 0; m < 4; ++m) {
                    // bin of the slice transform in register 4t + m of lane l (os1024_core, stage 3)
                    const int kp = (l & 15) + 16 * (2 * m + 8 * ((l >> 4) & 1) + (l >> 5)) + 256 * t;
                    const int k = k0 + 16 * kp;
                    hdev[(4 * t + m) * 1024 + tid] =
                        make_float2(static_cast<float>(re[k] / XF), static_cast<float>(im[k] / XF));
                }
        }
        float2* d = nullptr;
        COMMS_TRY(upload_f2(hdev, &d));
        h->d_xh.push_back(d);
    }
    const int lds = static_cast<int>(X_LDS_BYTES);
#define COMMS_X_ATTR(HRV, INV) \
    COMMS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fir_os16k_kernel<HRV, INV>), hipFuncAttributeMaxDynamicSharedMemorySize, lds))
#define COMMS_X_ATTR4(INV) COMMS_X_ATTR(1, INV); COMMS_X_ATTR(2, INV); COMMS_X_ATTR(3, INV); COMMS_X_ATTR(4, INV)
    COMMS_X_ATTR4(const float2*);
    COMMS_X_ATTR4(InI16);
    COMMS_X_ATTR4(InU8);
#undef COMMS_X_ATTR4
#undef COMMS_X_ATTR
    h->x_ready = true;
    return COMMS_OK;
}

static comms_status_t fir_prepare_direct(comms_fir* h) {
    if (h->d_taps_pad) return COMMS_OK;
    COMMS_ARG(h->n_eff <= DIRECT_MAX_TAPS, "direct-form FIR supports at most %d taps, got %d",
              DIRECT_MAX_TAPS, h->n_eff);
    h->NP = (h->n_eff + 7) / 8 * 8;
    std::vector<float2> tp(h->NP, make_float2(0.f, 0.f));
    for (int k = 0; k < h->n_eff; ++k) tp[k] = make_float2(h->taps[k].re, h->taps[k].im);
    COMMS_HIP_TRY(hipMalloc(&h->d_taps_pad, tp.size() * sizeof(float2)));
    COMMS_HIP_TRY(hipMemcpy(h->d_taps_pad, tp.data(), tp.size() * sizeof(float2), hipMemcpyHostToDevice));
    return COMMS_OK;
}

// Kernel choice.  COMMS_FIR_OVERLAP_SAVE means "the right overlap-save size":
// one wave per 1024-point segment up to 257 taps, one workgroup per 4096-point
// segment above.  Crossovers measured on MI355X.
static int fir_pick(const comms_fir* h, size_t n) {
    int algo = h->algo;
    if (algo == COMMS_FIR_AUTO) {
        // Measured on MI355X (launch to completion, `scripts/bench_fir.py` / `scripts/ab_libs.py`), 8 ... 255 taps, 2^16 ...
        // 2^24 samples: the direct kernel costs about 6.5 us + 0.025 us/tap + n * (2.2 + 0.024 * taps) ps (its stores
        // coalesced through LDS since round 3: 3.3 -> 2.2 ps), the 1024-point overlap-save kernel about 12.4 us + n * 2.0 ps
        // for any tap count up to 257.  Radio-sized batches (2^18 ... 2^20 samples) of the 32- and 63-tap filters the
        // reference's examples use are therefore direct-form work (7-9 us against 9-13); long streams are not.
        if (h->n_eff > DIRECT_MAX_TAPS) {
            algo = COMMS_FIR_OVERLAP_SAVE;
        } else if (n < 1024) {
            algo = COMMS_FIR_DIRECT;  // less than one segment
        } else {
            const double t = static_cast<double>(h->n_eff), nn = static_cast<double>(n);
            const double direct_ps = 6.5e6 + 0.025e6 * t + nn * (2.2 + 0.024 * t);
            // (4-wave workgroups, used up to 1024 segments, take about 1.1 us off the fixed part)
            const double os_ps = (nn <= 768.0 * 4 * kNumCU ? 11.3e6 : 12.4e6) + nn * 2.0;
            algo = direct_ps < os_ps ? COMMS_FIR_DIRECT : COMMS_FIR_OVERLAP_SAVE;
        }
    }
    if (algo == COMMS_FIR_OVERLAP_SAVE) {
        // 16384-point kernel (halo 1024 * ceil((taps - 1) / 1024)): 72-74 us per 2^24 samples up to 2049 taps, 74-78 at
        // 4097.  4096-point kernel (halo 256 * ceil((taps - 1) / 256)): 56-58 us up to 769 taps, 62 at 1025, 66 at 1281, 70
        // at 1537, 74 at 1793, 80 at 2049 (2^24; the same order at 2^26).  Below 2^23 samples the 16384-point segments are
        // too few to fill the chip and the 4096-point kernel wins at every tap count -- scripts/sweep_os.py,
        // profiles/r03_sweep_os.txt.
        const bool big = h->n_eff > 2049 || (h->n_eff > 1537 && n >= (static_cast<size_t>(1) << 23));
        algo = h->n_eff <= 257 ? COMMS_FIR_OS1024 : big ? COMMS_FIR_OS16K : COMMS_FIR_OS4096;
    }
    return algo;
}


static comms_status_t fir_upload_state(comms_fir* h, const comms_c32* state, size_t n_state) {
    // reference layout: state[0] newest ... -> device ring is time-ordered (oldest first)
    std::vector<float2> ring(h->n_eff, make_float2(0.f, 0.f));
    for (int k = 0; k < h->n_eff && static_cast<size_t>(k) < n_state; ++k)
        ring[h->n_eff - 1 - k] = make_float2(state[k].re, state[k].im);
    COMMS_HIP_TRY(hipMemcpy(h->d_hist[h->cur], ring.data(), ring.size() * sizeof(float2), hipMemcpyHostToDevice));
    return COMMS_OK;
}

extern "C" {

comms_status_t comms_fir_create(const comms_c32* taps, size_t n_taps, const comms_c32* state,
                                size_t n_state, int32_t device, comms_fir_t** out) {
    COMMS_ARG(out != nullptr, "out is NULL");
    *out = nullptr;
    COMMS_ARG(taps != nullptr && n_taps > 0, "taps must hold at least one tap (the reference panics on an empty state)");
    COMMS_ARG(state == nullptr || n_state > 0, "a user state must hold at least one sample");
    size_t n_eff = n_taps;
    if (state && n_state < n_eff) n_eff = n_state;  // zip(taps, state), fir.rs:53
    COMMS_ARG(n_eff <= (1u << 20), "too many taps (%zu)", n_eff);
    comms_fir* h = new (std::nothrow) comms_fir;
    COMMS_ARG(h != nullptr, "out of host memory");
    comms_status_t st = h->init(device);
    if (st != COMMS_OK) {
        delete h;
        return st;
    }
    h->n_eff = static_cast<int>(n_eff);
    h->taps.assign(taps, taps + n_eff);
    h->real_taps = true;
    for (size_t k = 0; k < n_eff; ++k)
        if (taps[k].im != 0.0f) h->real_taps = false;
    for (int i = 0; i < 2; ++i) {
        hipError_t e = hipMalloc(&h->d_hist[i], n_eff * sizeof(float2));
        if (e == hipSuccess) e = zero_device(h->d_hist[i], n_eff * sizeof(float2));
        if (e != hipSuccess) {
            free_fir(h);
            return fail(COMMS_ERR_DEVICE, "FIR history alloc: %s", hipGetErrorString(e));
        }
    }
    if (state) {
        st = fir_upload_state(h, state, n_state);
        if (st != COMMS_OK) {
            free_fir(h);
            return st;
        }
    }
    *out = h;
    return COMMS_OK;
}

comms_status_t comms_fir_set_algo(comms_fir_t* h, int32_t algo) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG(algo >= COMMS_FIR_AUTO && algo <= COMMS_FIR_OS1024_FIXED, "unknown algo %d", algo);
    h->os1024_fixed = algo == COMMS_FIR_OS1024_FIXED;
    if (h->os1024_fixed) algo = COMMS_FIR_OS1024;
    COMMS_ARG(algo != COMMS_FIR_OS1024 || h->n_eff <= 257,
              "the 1024-point overlap-save kernel supports at most 257 taps");
    COMMS_ARG(algo != COMMS_FIR_DIRECT || h->n_eff <= DIRECT_MAX_TAPS,
              "direct-form FIR supports at most %d taps", DIRECT_MAX_TAPS);

    h->algo = algo;
    return COMMS_OK;
}

// How comms_fir_run_dev launches the 1024-point overlap-save FIR for n samples.
struct Os1024Plan {
    int hr;       // halo rows of 64 samples: 1 / 2 / 3 / 4 for <= 65 / 129 / 193 / 257 taps, 1024 - 64 hr new samples per segment
    bool dyn;     // ticketed segments (fir_os1024_dyn_kernel) rather than fixed runs (fir_os1024_kernel)
    int wpb;      // waves per workgroup of the fixed-run kernel
    size_t min_run, nseg;
};
static Os1024Plan os1024_plan(const comms_fir* h, size_t n) {
    static const bool short_halo_ok = tune_int("COMMS_OS1024_SHORT_HALO", 1) != 0;
    static const int wpb_env = tune_int("COMMS_OS1024_WPB", 0);
    static const int min_run = tune_int("COMMS_OS1024_MINRUN", 1);
    // ticketed segments pay once there are a few segments per wave slot (they trim the tail of the launch):
    // measured -5 % at 2^22 samples, -11 % at 2^24, -9 % at 2^26, +3 % at 2^21 (scripts/ab_fir.py)
    static const size_t dyn_minseg = static_cast<size_t>(tune_int("COMMS_OS1024_DYN_MINSEG", 4096));
    Os1024Plan p{};
    p.hr = !short_halo_ok ? 4 : h->n_eff <= 65 ? 1 : h->n_eff <= 129 ? 2 : h->n_eff <= 193 ? 3 : 4;
    p.min_run = static_cast<size_t>(min_run);
    const size_t wv = 1024 - 64 * static_cast<size_t>(p.hr);
    p.nseg = (n + wv - 1) / wv;
    // up to one segment per SIMD of the chip, 4-wave workgroups spread the batch over four times as many
    // CUs, one wave per SIMD (8.6-9.3 us instead of 9.8-10.4 up to 2^19 samples; slower from 2^20 on)
    const int wpb = wpb_env ? wpb_env : p.nseg <= 4u * kNumCU ? 4 : 16;
    p.wpb = wpb;
    const int mode = os1024_dynamic_mode();
    p.dyn = mode != 0 && !h->os1024_fixed && wpb == 16 && p.nseg >= dyn_minseg;
    return p;
}

comms_status_t comms_fir_get_kernel(const comms_fir_t* h, size_t n, char* name, size_t name_len) {
    COMMS_ARG(h && name && name_len, "NULL argument");
    const char* k = "fir_direct_kernel";
    switch (fir_pick(h, n)) {
        case COMMS_FIR_OS1024: k = os1024_plan(h, n).dyn ? "fir_os1024_dyn_kernel" : "fir_os1024_kernel"; break;
        case COMMS_FIR_OS4096: k = "fir_os4096_kernel"; break;
        case COMMS_FIR_OS16K: k = "fir_os16k_kernel"; break;
        default: break;
    }
    std::snprintf(name, name_len, "%s", k);
    return COMMS_OK;
}

comms_status_t comms_fir_get_algo(const comms_fir_t* h, size_t n, int32_t* out_algo) {
    COMMS_ARG(h && out_algo, "NULL argument");
    *out_algo = fir_pick(h, n);
    return COMMS_OK;
}

}  // extern "C"

// Wire-format input for the kernels that read Complex<f32> only: one conversion pass (iqformat.hip's
// kernels) into a handle-owned buffer on the same stream.  The direct, ticketed 1024-point and
// decimating-chain kernels convert in their load stage instead and never come here.
static comms_status_t fir_converted_input(comms_fir* h, const void* d_in, size_t n, hipStream_t s, const float2** out) {
    if (h->in_fmt == COMMS_IQ_C32) {
        *out = static_cast<const float2*>(d_in);
        return COMMS_OK;
    }
    COMMS_TRY(h->conv.reserve(n * sizeof(float2)));
    comms_c32* tmp = static_cast<comms_c32*>(h->conv.p);
    if (h->in_fmt == COMMS_IQ_I16)
        COMMS_TRY(comms_iq_i16_to_c32_dev(static_cast<const int16_t*>(d_in), n, h->in_scale, tmp, h->device, s));
    else
        COMMS_TRY(comms_iq_u8_to_c32_dev(static_cast<const uint8_t*>(d_in), n, tmp, h->device, s));
    *out = static_cast<const float2*>(h->conv.p);
    return COMMS_OK;
}

template <int HR, class In>
static comms_status_t launch_dyn_in(hipStream_t s, In in, comms_fir* h, float2* o, size_t n, const WTables& tb, float2* nh,
                                    hipEvent_t ea, hipEvent_t eb, KStamp ks) {
    return launch_os1024_dyn<HR, false, In>(s, in, h->d_hist[h->cur], h->n_eff, o, n, tb, nh, nullptr, ea, eb, ks);
}
template <class In>
static comms_status_t launch_dyn_hr(int hr, hipStream_t s, In in, comms_fir* h, float2* o, size_t n, const WTables& tb,
                                    float2* nh, hipEvent_t ea, hipEvent_t eb, KStamp ks) {
    switch (hr) {
        case 1: return launch_dyn_in<1>(s, in, h, o, n, tb, nh, ea, eb, ks);
        case 2: return launch_dyn_in<2>(s, in, h, o, n, tb, nh, ea, eb, ks);
        case 3: return launch_dyn_in<3>(s, in, h, o, n, tb, nh, ea, eb, ks);
        default: return launch_dyn_in<4>(s, in, h, o, n, tb, nh, ea, eb, ks);
    }
}
template <class In>
static comms_status_t launch_fixed_hr(int hr, int wpb, size_t runs, hipStream_t s, In in, const float2* hist, int n_eff, float2* o,
                                      size_t n, size_t nseg, const WTables& tb, float2* nh, hipEvent_t ea, hipEvent_t eb) {
    switch (hr) {
        case 1: return launch_os1024<0, 1, In>(wpb, runs, s, in, hist, n_eff, o, n, nseg, tb, nh, ChainArgs{}, ea, eb);
        case 2: return launch_os1024<0, 2, In>(wpb, runs, s, in, hist, n_eff, o, n, nseg, tb, nh, ChainArgs{}, ea, eb);
        case 3: return launch_os1024<0, 3, In>(wpb, runs, s, in, hist, n_eff, o, n, nseg, tb, nh, ChainArgs{}, ea, eb);
        default: return launch_os1024<0, 4, In>(wpb, runs, s, in, hist, n_eff, o, n, nseg, tb, nh, ChainArgs{}, ea, eb);
    }
}
// diagnostic build: comms_debug_os16k_fault(1) makes the next 16384-point launches withhold one LDS signal
static std::atomic<int> g_os16k_fault{0};
static int os16k_fault() { return g_os16k_fault.load(std::memory_order_relaxed); }
#ifdef COMMS_DIAG
// bit 0: withhold a signal; bits 1...: wave priority scheme (trial, see the kernel)
extern "C" void comms_debug_os16k_fault(int on) { g_os16k_fault.store(on, std::memory_order_relaxed); }
// buf: device memory, [workgroups][16][X_TRACE_SEGS][16] u64 (null: off); returns X_TRACE_SEGS
extern "C" unsigned comms_debug_os16k_trace(void* buf) {
    unsigned long long* p = static_cast<unsigned long long*>(buf);
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_x_trace), &p, sizeof(p)) != hipSuccess) return 0;
    return X_TRACE_SEGS;
}
#endif
template <class In>
static void launch_os16k_hr(int hr, unsigned blocks, size_t lds, hipStream_t s, In in, const float2* hist, int n_eff, float2* o, size_t n,
                            size_t nseg, const XTables& tb, float2* nh, int dl, int acc, KStamp ks, unsigned* err) {
    const int fault = os16k_fault();
    switch (hr) {
        case 1: fir_os16k_kernel<1, In><<<dim3(blocks), dim3(1024), lds, s>>>(in, hist, n_eff, o, n, nseg, tb, nh, dl, acc, ks, err, fault); break;
        case 2: fir_os16k_kernel<2, In><<<dim3(blocks), dim3(1024), lds, s>>>(in, hist, n_eff, o, n, nseg, tb, nh, dl, acc, ks, err, fault); break;
        case 3: fir_os16k_kernel<3, In><<<dim3(blocks), dim3(1024), lds, s>>>(in, hist, n_eff, o, n, nseg, tb, nh, dl, acc, ks, err, fault); break;
        default: fir_os16k_kernel<4, In><<<dim3(blocks), dim3(1024), lds, s>>>(in, hist, n_eff, o, n, nseg, tb, nh, dl, acc, ks, err, fault); break;
    }
}
template <class In>
static void launch_direct_in(comms_fir* h, In in, const float2* hist, float2* o, size_t n, float2* nh, unsigned blocks,
                             size_t lds, hipStream_t s) {
    if (h->real_taps)
        fir_direct_kernel<true, In><<<dim3(blocks), dim3(256), lds, s>>>(in, hist, h->n_eff, h->d_taps_pad, h->NP, o, n, nh);
    else
        fir_direct_kernel<false, In><<<dim3(blocks), dim3(256), lds, s>>>(in, hist, h->n_eff, h->d_taps_pad, h->NP, o, n, nh);
}

extern "C" {

comms_status_t comms_fir_run_dev(comms_fir_t* h, const comms_c32* d_in_any, size_t n,
                                 comms_c32* d_out, void* stream) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG((d_in_any && d_out) || !n, "NULL device pointer");
    COMMS_TRY(use_device(h->device));
    if (!n) return COMMS_OK;
    const void* d_in = d_in_any;  // n samples in the handle's input format
    const size_t in_elem = h->in_fmt == COMMS_IQ_I16 ? 4 : h->in_fmt == COMMS_IQ_U8 ? 2 : 8;
    COMMS_ARG(!ranges_overlap(d_in, n * in_elem, d_out, n * 8), "FIR cannot run in place");
    COMMS_ARG((reinterpret_cast<uintptr_t>(d_in) & (in_elem - 1)) == 0 && (reinterpret_cast<uintptr_t>(d_out) & 7) == 0,
              "device pointers must be aligned to one sample");
    COMMS_TRY(fir_check_sticky(h));
    hipStream_t s = nullptr;
    COMMS_TRY(h->enter(stream, &s));
    float2* o = reinterpret_cast<float2*>(d_out);
    const float2* hist = h->d_hist[h->cur];
    float2* nh = h->d_hist[h->cur ^ 1];  // the kernel's workgroup 0 advances the history into it
    const int algo = fir_pick(h, n);
    // kernels whose load stages read raw IQ (the 4096-point one in its default three-workgroups-per-CU build)
    static const int os4096_wps = tune_int("COMMS_OS4096_WPS", 3);
    const bool fused_fmt = algo == COMMS_FIR_DIRECT || algo == COMMS_FIR_OS1024 || algo == COMMS_FIR_OS16K ||
                           (algo == COMMS_FIR_OS4096 && os4096_wps == 3);
    const float2* in = nullptr;  // Complex<f32> view of the input (the conversion pass, where the kernel needs one)
    if (!fused_fmt || h->in_fmt == COMMS_IQ_C32) COMMS_TRY(fir_converted_input(h, d_in, n, s, &in));
    if (algo == COMMS_FIR_DIRECT) {
        COMMS_TRY(fir_prepare_direct(h));
        const unsigned blocks = static_cast<unsigned>((n + DTILE - 1) / DTILE);
        const int nrows = (DTILE + h->NP) / 8;
        const size_t lds = static_cast<size_t>(nrows) * DROW * sizeof(float2) + static_cast<size_t>(h->NP) * sizeof(float2);
        h->tic(s);
        if (h->in_fmt == COMMS_IQ_I16)
            launch_direct_in(h, InI16{static_cast<const short2*>(d_in), h->in_scale}, hist, o, n, nh, blocks, lds, s);
        else if (h->in_fmt == COMMS_IQ_U8)
            launch_direct_in(h, InU8{static_cast<const uchar2*>(d_in)}, hist, o, n, nh, blocks, lds, s);
        else
            launch_direct_in(h, in, hist, o, n, nh, blocks, lds, s);
        h->toc(s);
        COMMS_TRY(launch_ok("fir_direct_kernel"));
    } else if (algo == COMMS_FIR_OS1024) {
        COMMS_TRY(fir_prepare_os1024(h));
        const Os1024Plan pl = os1024_plan(h, n);
        const size_t nseg = pl.nseg;
        const size_t runs = os1024_runs(pl.wpb, nseg, pl.min_run);
        WTables tb{reinterpret_cast<const cf*>(h->d_wtw1), reinterpret_cast<const cf*>(h->d_wtw2), reinterpret_cast<const cf*>(h->d_whdev)};
        hipEvent_t ea = nullptr, eb = nullptr;
        const bool own_stamps = pl.wpb == 16;  // (16-wave launches report the kernel's own begin / end through the pair)
        if (own_stamps)
            (void)h->take_events(ea, eb);
        else
            h->tic(s);
        const KStamp ks = pl.dyn ? h->next_stamp() : KStamp{nullptr, nullptr};
        if (pl.dyn) {
            if (h->in_fmt == COMMS_IQ_I16)
                COMMS_TRY(launch_dyn_hr(pl.hr, s, InI16{static_cast<const short2*>(d_in), h->in_scale}, h, o, n, tb, nh, ea, eb, ks));
            else if (h->in_fmt == COMMS_IQ_U8)
                COMMS_TRY(launch_dyn_hr(pl.hr, s, InU8{static_cast<const uchar2*>(d_in)}, h, o, n, tb, nh, ea, eb, ks));
            else
                COMMS_TRY(launch_dyn_hr(pl.hr, s, in, h, o, n, tb, nh, ea, eb, ks));
        } else if (h->in_fmt == COMMS_IQ_I16) {
            COMMS_TRY(launch_fixed_hr(pl.hr, pl.wpb, runs, s, InI16{static_cast<const short2*>(d_in), h->in_scale}, hist, h->n_eff, o, n, nseg, tb, nh, ea, eb));
        } else if (h->in_fmt == COMMS_IQ_U8) {
            COMMS_TRY(launch_fixed_hr(pl.hr, pl.wpb, runs, s, InU8{static_cast<const uchar2*>(d_in)}, hist, h->n_eff, o, n, nseg, tb, nh, ea, eb));
        } else {
            COMMS_TRY(launch_fixed_hr(pl.hr, pl.wpb, runs, s, in, hist, h->n_eff, o, n, nseg, tb, nh, ea, eb));
        }
        if (!own_stamps) h->toc(s);
        COMMS_TRY(launch_ok("fir_os1024_kernel"));
    } else if (algo == COMMS_FIR_OS16K) {
        COMMS_TRY(fir_prepare_os16k(h));
        // halo rows: what the taps need (one pass), the full four for the 4097-tap partitions of longer filters
        const int hr = h->x_part > 1 ? 4 : h->n_eff <= 1025 ? 1 : h->n_eff <= 2049 ? 2 : h->n_eff <= 3073 ? 3 : 4;
        const size_t xv = static_cast<size_t>(16 - hr) * 1024;
        const size_t nseg = (n + xv - 1) / xv;
        const unsigned blocks = static_cast<unsigned>(nseg < static_cast<size_t>(kNumCU) ? nseg : kNumCU);
        const size_t lds = X_LDS_BYTES;
        h->tic(s);
        const KStamp ks = h->next_stamp();  // (the passes of a partitioned filter stamp the same slots: the whole call)
        for (int pt = 0; pt < h->x_part; ++pt) {
            XTables tb{reinterpret_cast<const cf*>(h->d_xt[0]), reinterpret_cast<const cf*>(h->d_xt[1]),
                       reinterpret_cast<const cf*>(h->d_xt[2]), reinterpret_cast<const cf*>(h->d_xt[3]),
                       reinterpret_cast<const cf*>(h->d_xh[pt])};
            const int dl = pt * X_PART, acc = pt ? 1 : 0;
            if (h->in_fmt == COMMS_IQ_I16)
                launch_os16k_hr(hr, blocks, lds, s, InI16{static_cast<const short2*>(d_in), h->in_scale}, hist, h->n_eff, o, n, nseg, tb, nh, dl, acc, ks, h->d_err);
            else if (h->in_fmt == COMMS_IQ_U8)
                launch_os16k_hr(hr, blocks, lds, s, InU8{static_cast<const uchar2*>(d_in)}, hist, h->n_eff, o, n, nseg, tb, nh, dl, acc, ks, h->d_err);
            else
                launch_os16k_hr(hr, blocks, lds, s, in, hist, h->n_eff, o, n, nseg, tb, nh, dl, acc, ks, h->d_err);
        }
        h->toc(s);
        COMMS_TRY(launch_ok("fir_os16k_kernel"));
    } else {
        COMMS_TRY(fir_prepare_os(h));
        const size_t V = OSF - 256 * static_cast<size_t>(h->hblk);
        const size_t nseg = (n + V - 1) / V;
        // persistent grid: WPS workgroups per CU (one wave of each per SIMD), segments dealt round-robin.
        // Three workgroups per CU with the samples read as 4-byte loads (InC32Split, or the raw i16 / u8 views): with
        // 8-byte loads the kernel's load stage needs ~200 VGPRs, which at three workgroups (budget 170) spilled 24
        // registers and cost 4-21 %, while two workgroups leave the CU short of waves; with 4-byte loads it fits in 130
        // (511 taps at 2^24: 86.8 us spilling, 69.7 at two workgroups, 66.6 now; 2049 taps: 107.5 / 105.7 / 97.5).
        // COMMS_OS4096_WPS=2 / 4 select the 8-byte-load builds (raw input then takes a conversion pass).
        const int wps = os4096_wps;
        // segments b, b + G, ... per workgroup (the chip sweeps the stream as one window: 2-4 % faster at 2^24 ...
        // 2^26 than a contiguous run per workgroup, 1 % at 2^28); 0 restores the runs
        static const int il = tune_int("COMMS_OS4096_INTERLEAVE", 1);
        const size_t slots = static_cast<size_t>(wps) * kNumCU;
        const unsigned blocks = static_cast<unsigned>(nseg < slots ? nseg : slots);
        h->tic(s);
        for (int pt = 0; pt < h->n_part; ++pt) {
            OsTables tb{reinterpret_cast<const cf*>(h->d_tw1), reinterpret_cast<const cf*>(h->d_tw2),
                        reinterpret_cast<const cf*>(h->d_hparts[pt])};
            const int dl = pt * OS_PART, acc = pt ? 1 : 0;
            if (wps == 4)
                fir_os4096_kernel<4><<<dim3(blocks), dim3(256), 0, s>>>(in, hist, h->n_eff, o, n, h->hblk, nseg, tb, nh, dl, acc, il, OsDec{});
            else if (wps == 2)
                fir_os4096_kernel<2><<<dim3(blocks), dim3(256), 0, s>>>(in, hist, h->n_eff, o, n, h->hblk, nseg, tb, nh, dl, acc, il, OsDec{});
            else if (h->in_fmt == COMMS_IQ_I16)
                fir_os4096_kernel<3, InI16><<<dim3(blocks), dim3(256), 0, s>>>(InI16{static_cast<const short2*>(d_in), h->in_scale}, hist, h->n_eff, o, n,
                                                                              h->hblk, nseg, tb, nh, dl, acc, il, OsDec{});
            else if (h->in_fmt == COMMS_IQ_U8)
                fir_os4096_kernel<3, InU8><<<dim3(blocks), dim3(256), 0, s>>>(InU8{static_cast<const uchar2*>(d_in)}, hist, h->n_eff, o, n, h->hblk, nseg,
                                                                             tb, nh, dl, acc, il, OsDec{});
            else
                fir_os4096_kernel<3, InC32Split><<<dim3(blocks), dim3(256), 0, s>>>(InC32Split{reinterpret_cast<const float*>(in)}, hist, h->n_eff, o, n,
                                                                                   h->hblk, nseg, tb, nh, dl, acc, il, OsDec{});
        }
        h->toc(s);
        COMMS_TRY(launch_ok("fir_os4096_kernel"));
    }
    h->cur ^= 1;
    return COMMS_OK;
}

comms_status_t comms_fir_run(comms_fir_t* h, const comms_c32* in, size_t n, comms_c32* out) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG((in && out) || !n, "NULL host pointer");
    COMMS_TRY(use_device(h->device));
    if (!n) return COMMS_OK;
    const size_t in_elem = h->in_fmt == COMMS_IQ_I16 ? 4 : h->in_fmt == COMMS_IQ_U8 ? 2 : 8;
    // (long batches go through the chunked host pipeline, common.hpp: the history streams across the chunks as across calls)
    COMMS_TRY(h->run_host_units(in, n * in_elem, in_elem, out, n * sizeof(comms_c32), sizeof(comms_c32), [&](void* d_in, void* d_out, size_t ib, size_t) {
        return comms_fir_run_dev(h, static_cast<const comms_c32*>(d_in), ib / in_elem, static_cast<comms_c32*>(d_out), COMMS_STREAM_HANDLE);
    }));
    // run_host has waited for this call's launches: a wait of the 16384-point kernel that ran out in THIS call is
    // reported by this call (its samples are in `out`, and they are wrong)
    return fir_check_sticky(h);
}

comms_status_t comms_fir_get_state(comms_fir_t* h, comms_c32* state, size_t n_state) {
    COMMS_ARG(h && state, "NULL argument");
    COMMS_ARG(n_state <= static_cast<size_t>(h->n_eff), "n_state %zu exceeds the %d effective taps", n_state, h->n_eff);
    COMMS_TRY(use_device(h->device));
    COMMS_TRY(h->quiesce());  // the history is advanced by the launches, on whatever stream they ran
    COMMS_TRY(fir_check_sticky(h));
    std::vector<float2> ring(h->n_eff);
    COMMS_HIP_TRY(hipMemcpy(ring.data(), h->d_hist[h->cur], ring.size() * sizeof(float2), hipMemcpyDeviceToHost));
    for (size_t k = 0; k < n_state; ++k) {
        state[k].re = ring[h->n_eff - 1 - k].x;
        state[k].im = ring[h->n_eff - 1 - k].y;
    }
    return COMMS_OK;
}

comms_status_t comms_fir_set_state(comms_fir_t* h, const comms_c32* state, size_t n_state) {
    COMMS_ARG(h && state, "NULL argument");
    COMMS_ARG(n_state == static_cast<size_t>(h->n_eff), "state must hold exactly the %d effective taps", h->n_eff);
    COMMS_TRY(use_device(h->device));
    COMMS_TRY(h->quiesce());  // no pending launch may still read the buffer that is overwritten
    COMMS_TRY(fir_check_sticky(h));
    return fir_upload_state(h, state, n_state);
}

comms_status_t comms_fir_set_timer(comms_fir_t* h, comms_timer_t* t) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    h->timer = t;
    return COMMS_OK;
}

comms_status_t comms_fir_set_input_format(comms_fir_t* h, int32_t format, float scale) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG(format == COMMS_IQ_C32 || format == COMMS_IQ_I16 || format == COMMS_IQ_U8, "unknown sample format %d", format);
    COMMS_ARG(format != COMMS_IQ_I16 || std::isfinite(scale), "scale must be finite");
    h->in_fmt = format;
    h->in_scale = format == COMMS_IQ_I16 ? scale : 1.0f;
    return COMMS_OK;
}

// ---- fused chain entry (internal; used by chain.hip).  mode = CH_* bits.
comms_status_t comms_fir_run_fused_dev(comms_fir_t* h, const comms_c32* d_in, size_t n, void* d_out,
                                       int32_t mode, uint64_t turns0, uint64_t frac, uint32_t rate,
                                       const void* fm_prev, void* fm_prev_new, void* stream) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG((d_in && d_out) || !n, "NULL device pointer");
    COMMS_ARG(h->n_eff <= 257, "the fused chain kernel supports at most 257 taps");
    COMMS_ARG(rate >= 1 && n % rate == 0, "n must be a multiple of the decimation rate");
    COMMS_ARG(!(mode & CH_FM) || (rate <= 64 && h->n_eff + static_cast<int>(rate) <= 257),
              "fused FM demod needs rate <= 64 and taps + rate <= 257");
    COMMS_TRY(fir_check_sticky(h));
    COMMS_TRY(use_device(h->device));
    if (!n) return COMMS_OK;
    COMMS_ARG(!ranges_overlap(d_in, n * 8, d_out, (n / rate) * ((mode & CH_FM) ? 4 : 8)), "the fused chain cannot run in place");
    COMMS_TRY(fir_prepare_os1024(h));
    hipStream_t s = nullptr;
    COMMS_TRY(h->enter(stream, &s));
    const float2* in = reinterpret_cast<const float2*>(d_in);
    float2* o = reinterpret_cast<float2*>(d_out);
    const float2* hist = h->d_hist[h->cur];
    float2* nh = h->d_hist[h->cur ^ 1];
    ChainArgs ch{};
    ch.turns0 = turns0;
    ch.frac = frac;
    mix_host_rotor(static_cast<uint64_t>(WV) * frac, ch.seg_c, ch.seg_s);
    for (int a = 0; a < 16; ++a) {
        double c, sn;
        mix_host_rotor(static_cast<uint64_t>(64 * a) * frac, c, sn);
        ch.step_a[a] = make_float2(static_cast<float>(c), static_cast<float>(sn));
        const unsigned off = a >= 4 ? 64u * (a - 4) : 0u;
        ch.q_a[a] = off / rate;
        ch.r_a[a] = off % rate;
    }
    ch.rate = rate;
    ch.q_seg = WV / rate;
    ch.r_seg = WV % rate;
    ch.fm_prev = static_cast<const float2*>(fm_prev);
    ch.fm_prev_new = static_cast<float2*>(fm_prev_new);
    const size_t nseg = (n + WV - 1) / WV;
    static const int wpb = tune_int("COMMS_OS1024_WPB", 16);
    const size_t runs = os1024_runs(wpb, nseg, 1);
    WTables tb{reinterpret_cast<const cf*>(h->d_wtw1), reinterpret_cast<const cf*>(h->d_wtw2), reinterpret_cast<const cf*>(h->d_whdev)};
    h->tic(s);
    switch (mode) {
        case CH_PRE | CH_DEC:
            COMMS_TRY(launch_os1024<CH_PRE | CH_DEC>(wpb, runs, s, in, hist, h->n_eff, o, n, nseg, tb, nh, ch));
            break;
        case CH_PRE | CH_DEC | CH_FM:
            COMMS_TRY(launch_os1024<CH_PRE | CH_DEC | CH_FM>(wpb, runs, s, in, hist, h->n_eff, o, n, nseg, tb, nh, ch));
            break;
        case CH_POST | CH_DEC:
            COMMS_TRY(launch_os1024<CH_POST | CH_DEC>(wpb, runs, s, in, hist, h->n_eff, o, n, nseg, tb, nh, ch));
            break;
        case CH_POST | CH_DEC | CH_FM:
            COMMS_TRY(launch_os1024<CH_POST | CH_DEC | CH_FM>(wpb, runs, s, in, hist, h->n_eff, o, n, nseg, tb, nh, ch));
            break;
        case CH_TRACE:  // diagnostic build: production geometry, per-wave times into fm_prev_new
            if (os1024_dynamic_mode() != 0)
                COMMS_TRY((launch_os1024_dyn<4, true>(s, in, hist, h->n_eff, o, n, tb, nh, fm_prev_new)));
            else
                COMMS_TRY(launch_os1024<CH_TRACE>(16, os1024_runs(16, nseg, 1), s, in, hist, h->n_eff, o, n, nseg, tb, nh, ch));
            break;
        case CH_STAMP:  // diagnostic build: 4-wave workgroups, stamps into fm_prev_new
            COMMS_TRY(launch_os1024<CH_STAMP>(4, os1024_runs(4, nseg, 4), s, in, hist, h->n_eff, o, n, nseg, tb, nh, ch));
            break;
        default:
            return fail(COMMS_ERR_ARG, "unsupported fused mode %d", mode);
    }
    h->toc(s);
    COMMS_TRY(launch_ok("fir_os1024_kernel (fused)"));
    h->cur ^= 1;
    return COMMS_OK;
}

// ---- long-filter decimating chain entry (internal; used by chain.hip): FIR (4096-point overlap-save) -> mixer -> keep every
// rate-th output, one launch.  d_in: n samples in the handle's input format; d_out: n / rate Complex<f32>.
int32_t comms_fir_os4096_decim_supported(const comms_fir_t* h, uint32_t rate) {
    return h && h->n_eff > 257 && h->n_eff <= 1537 && rate >= 2 && rate <= (1u << 20) ? 1 : 0;
}
// ... and on the 16384-point kernel (1538 ... 4097 taps; Complex<f32> input)
int32_t comms_fir_os16k_decim_supported(const comms_fir_t* h, uint32_t rate) {
    return h && h->n_eff > 1537 && h->n_eff <= 4097 && rate >= 2 && rate <= (1u << 20) ? 1 : 0;
}

comms_status_t comms_fir_run_os16k_decim_dev(comms_fir_t* h, const comms_c32* d_in, size_t n, void* d_out, uint64_t turns0,
                                             uint64_t frac, uint32_t rate, void* stream) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG((d_in && d_out) || !n, "NULL device pointer");
    COMMS_ARG(comms_fir_os16k_decim_supported(h, rate), "the 16384-point decimating chain kernel takes 1538 ... 4097 taps and rates 2 ... 2^20");
    COMMS_ARG(n % rate == 0, "n must be a multiple of the decimation rate");
    COMMS_TRY(fir_check_sticky(h));
    COMMS_TRY(use_device(h->device));
    if (!n) return COMMS_OK;
    COMMS_ARG(!ranges_overlap(d_in, n * 8, d_out, (n / rate) * 8), "the decimating chain cannot run in place");
    COMMS_ARG((reinterpret_cast<uintptr_t>(d_in) & 7) == 0 && (reinterpret_cast<uintptr_t>(d_out) & 7) == 0,
              "device pointers must be aligned to one sample");
    COMMS_TRY(fir_prepare_os16k(h));
    COMMS_ARG(h->x_part == 1, "the filter does not fit one 16384-point pass");
    hipStream_t s = nullptr;
    COMMS_TRY(h->enter(stream, &s));
    const float2* in = reinterpret_cast<const float2*>(d_in);
    float2* o = reinterpret_cast<float2*>(d_out);
    const float2* hist = h->d_hist[h->cur];
    float2* nh = h->d_hist[h->cur ^ 1];
    const int hr = h->n_eff <= 2049 ? 2 : h->n_eff <= 3073 ? 3 : 4;
    const size_t xv = static_cast<size_t>(16 - hr) * 1024;
    const size_t nseg = (n + xv - 1) / xv;
    const unsigned blocks = static_cast<unsigned>(nseg < static_cast<size_t>(kNumCU) ? nseg : kNumCU);
    OsDec dc{};
    dc.turns0 = turns0;
    dc.frac = frac;
    dc.rate = rate;
    dc.c1 = 1024u % rate;
    dc.d1 = 1024u / rate;
    dc.dq = xv / rate;  // (a workgroup walks consecutive segments)
    dc.dr = static_cast<unsigned>(xv % rate);
    for (int j = 0; j < 16; ++j) {
        double c, sn;
        mix_host_rotor(static_cast<uint64_t>(1024 * j) * frac, c, sn);
        dc.step[j] = make_float2(static_cast<float>(c), static_cast<float>(sn));
    }
    XTables tb{reinterpret_cast<const cf*>(h->d_xt[0]), reinterpret_cast<const cf*>(h->d_xt[1]), reinterpret_cast<const cf*>(h->d_xt[2]),
               reinterpret_cast<const cf*>(h->d_xt[3]), reinterpret_cast<const cf*>(h->d_xh[0])};
    const size_t lds = X_LDS_BYTES;
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        COMMS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fir_os16k_kernel<2, const float2*, true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        COMMS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fir_os16k_kernel<3, const float2*, true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        COMMS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fir_os16k_kernel<4, const float2*, true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    }
    h->tic(s);
    const KStamp ks = h->next_stamp();
    const int fault = os16k_fault();
    switch (hr) {
        case 2: fir_os16k_kernel<2, const float2*, true><<<dim3(blocks), dim3(1024), lds, s>>>(in, hist, h->n_eff, o, n, nseg, tb, nh, 0, 0, ks, h->d_err, fault, dc); break;
        case 3: fir_os16k_kernel<3, const float2*, true><<<dim3(blocks), dim3(1024), lds, s>>>(in, hist, h->n_eff, o, n, nseg, tb, nh, 0, 0, ks, h->d_err, fault, dc); break;
        default: fir_os16k_kernel<4, const float2*, true><<<dim3(blocks), dim3(1024), lds, s>>>(in, hist, h->n_eff, o, n, nseg, tb, nh, 0, 0, ks, h->d_err, fault, dc); break;
    }
    h->toc(s);
    COMMS_TRY(launch_ok("fir_os16k_kernel (decimating)"));
    h->cur ^= 1;
    h->last_poly8 = false;
    return COMMS_OK;
}

comms_status_t comms_fir_run_os4096_decim_dev(comms_fir_t* h, const void* d_in, size_t n, void* d_out, uint64_t turns0,
                                              uint64_t frac, uint32_t rate, void* stream) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG((d_in && d_out) || !n, "NULL device pointer");
    COMMS_ARG(comms_fir_os4096_decim_supported(h, rate), "the 4096-point decimating chain kernel takes 258 ... 1537 taps and rates 2 ... 2^20");
    COMMS_ARG(n % rate == 0, "n must be a multiple of the decimation rate");
    COMMS_TRY(fir_check_sticky(h));
    COMMS_TRY(use_device(h->device));
    if (!n) return COMMS_OK;
    const size_t in_elem = h->in_fmt == COMMS_IQ_I16 ? 4 : h->in_fmt == COMMS_IQ_U8 ? 2 : 8;
    COMMS_ARG(!ranges_overlap(d_in, n * in_elem, d_out, (n / rate) * 8), "the decimating chain cannot run in place");
    COMMS_ARG((reinterpret_cast<uintptr_t>(d_in) & (in_elem - 1)) == 0 && (reinterpret_cast<uintptr_t>(d_out) & 7) == 0,
              "device pointers must be aligned to one sample");
    COMMS_TRY(fir_prepare_os(h));
    hipStream_t s = nullptr;
    COMMS_TRY(h->enter(stream, &s));
    float2* o = reinterpret_cast<float2*>(d_out);
    const float2* hist = h->d_hist[h->cur];
    float2* nh = h->d_hist[h->cur ^ 1];
    const size_t V = OSF - 256 * static_cast<size_t>(h->hblk);
    const size_t nseg = (n + V - 1) / V;
    const size_t slots = static_cast<size_t>(3) * kNumCU;
    const unsigned blocks = static_cast<unsigned>(nseg < slots ? nseg : slots);
    OsDec dc{};
    dc.turns0 = turns0;
    dc.frac = frac;
    dc.rate = rate;
    dc.c1 = 256u % rate;
    dc.d1 = 256u / rate;
    const unsigned long long hop = static_cast<unsigned long long>(blocks) * V;  // (segments b, b + G, ... per workgroup)
    dc.dq = hop / rate;
    dc.dr = static_cast<unsigned>(hop % rate);
    for (int j = 0; j < 16; ++j) {
        double c, sn;
        mix_host_rotor(static_cast<uint64_t>(256 * j) * frac, c, sn);
        dc.step[j] = make_float2(static_cast<float>(c), static_cast<float>(sn));
    }
    OsTables tb{reinterpret_cast<const cf*>(h->d_tw1), reinterpret_cast<const cf*>(h->d_tw2), reinterpret_cast<const cf*>(h->d_hparts[0])};
    h->tic(s);
    if (h->in_fmt == COMMS_IQ_I16)
        fir_os4096_kernel<3, InI16, true><<<dim3(blocks), dim3(256), 0, s>>>(InI16{static_cast<const short2*>(d_in), h->in_scale}, hist, h->n_eff, o, n,
                                                                            h->hblk, nseg, tb, nh, 0, 0, 1, dc);
    else if (h->in_fmt == COMMS_IQ_U8)
        fir_os4096_kernel<3, InU8, true><<<dim3(blocks), dim3(256), 0, s>>>(InU8{static_cast<const uchar2*>(d_in)}, hist, h->n_eff, o, n, h->hblk, nseg,
                                                                           tb, nh, 0, 0, 1, dc);
    else
        fir_os4096_kernel<3, InC32Split, true><<<dim3(blocks), dim3(256), 0, s>>>(InC32Split{static_cast<const float*>(d_in)}, hist, h->n_eff, o, n,
                                                                                 h->hblk, nseg, tb, nh, 0, 0, 1, dc);
    h->toc(s);
    COMMS_TRY(launch_ok("fir_os4096_kernel (decimating)"));
    h->cur ^= 1;
    h->last_poly8 = false;
    return COMMS_OK;
}

comms_status_t comms_fir_destroy(comms_fir_t* h) {
    if (!h) return COMMS_OK;
    free_fir(h);
    return COMMS_OK;
}

}  // extern "C"

// ================================================================= pulse handle
struct comms_pulse : Handle {
    int n_taps = 0;
    int sps = 1;
    int hist_len = 0;  // symbols of history kept: ceil(n_taps / sps)
    float2* d_taps = nullptr;
    float2* d_hist[2] = {nullptr, nullptr};
    int cur = 0;
    std::vector<comms_c32> taps;  // host copy (kernel-argument taps of the polyphase kernel)
    bool real_taps = false;
    // fused output mixer (comms_pulse_set_mixer): phase of the next output, step per output
    bool mix = false;
    uint64_t turns = 0, frac = 0;
    bool out_i16 = false;  // comms_pulse_set_output_format
    float out_scale = 1.0f;
};

// Launches pulse_poly_kernel if (sps, taps) fit it; false -> the caller runs the generic kernel.
template <int SPS>
static bool pulse_poly_try(comms_pulse* h, const float2* sym, size_t n_sym, float2* out, hipStream_t s) {
    constexpr int SPSP = SPS + (SPS & 1);
    int J = (h->n_taps + SPS - 1) / SPS;
    J = (J + comms::PP_JB - 1) / comms::PP_JB * comms::PP_JB;
    if (J > comms::PP_JMAX || J * SPSP > comms::PP_AMAX) return false;
    comms::PulseArgs a{};
    a.sym = sym;
    a.hist = h->d_hist[h->cur];
    a.new_hist = h->d_hist[h->cur ^ 1];
    a.out = out;
    a.n_sym = n_sym;
    a.hist_len = h->hist_len;
    a.J = J;
    for (int j = 0; j < J; ++j)
        for (int p = 0; p < SPS; ++p) {
            const int k = p + j * SPS;
            if (k < h->n_taps) {
                a.are[j * SPSP + p] = h->taps[k].re;
                a.aim[j * SPSP + p] = h->taps[k].im;
            }
        }
    const size_t ntiles = (n_sym + 255) / 256;
    const unsigned blocks = static_cast<unsigned>(ntiles < 8u * comms::kNumCU ? ntiles : 8u * comms::kNumCU);
    a.mx.out_i16 = h->out_i16 ? 1 : 0;
    a.mx.out_scale = h->out_scale;
    if (h->mix) {
        a.mx.on = 1;
        a.mx.turns0 = h->turns;
        a.mx.frac = h->frac;
        mix_host_rotor(static_cast<uint64_t>(blocks) * 256u * SPS * h->frac, a.mx.sweep_c, a.mx.sweep_s);
        for (int p = 0; p < SPS; ++p) {
            double c, sn;
            mix_host_rotor(static_cast<uint64_t>(p) * h->frac, c, sn);
            a.step[p] = make_float2(static_cast<float>(c), static_cast<float>(sn));
        }
    }
    // with a kernel timer attached: the kernel's own begin / end timestamps (events recorded around a launch of config 1's
    // size -- 5 us -- would mostly time the dispatch gap)
    hipEvent_t ea = nullptr, eb = nullptr;
    (void)h->take_events(ea, eb);
    a.ks = h->next_stamp();
#define COMMS_PULSE_GO(REAL, MIX)                                                                                            \
    do {                                                                                                                     \
        if (ea)                                                                                                              \
            hipExtLaunchKernelGGL((comms::pulse_poly_kernel<SPS, REAL, MIX>), dim3(blocks), dim3(256), 0u, s, ea, eb, 0u, a); \
        else                                                                                                                 \
            comms::pulse_poly_kernel<SPS, REAL, MIX><<<dim3(blocks), dim3(256), 0, s>>>(a);                                  \
    } while (0)
    if (h->mix) {
        if (h->real_taps) COMMS_PULSE_GO(true, true); else COMMS_PULSE_GO(false, true);
    } else {
        if (h->real_taps) COMMS_PULSE_GO(true, false); else COMMS_PULSE_GO(false, false);
    }
#undef COMMS_PULSE_GO
    return true;
}
static bool pulse_poly_launch(comms_pulse* h, const float2* sym, size_t n_sym, float2* out, hipStream_t s) {
    static const bool off = diag_knob("COMMS_PULSE_GENERIC", 0) != 0;
    if (off) return false;
    switch (h->sps) {
        case 2: return pulse_poly_try<2>(h, sym, n_sym, out, s);
        case 3: return pulse_poly_try<3>(h, sym, n_sym, out, s);
        case 4: return pulse_poly_try<4>(h, sym, n_sym, out, s);
        case 5: return pulse_poly_try<5>(h, sym, n_sym, out, s);
        case 6: return pulse_poly_try<6>(h, sym, n_sym, out, s);
        case 8: return pulse_poly_try<8>(h, sym, n_sym, out, s);
        case 10: return pulse_poly_try<10>(h, sym, n_sym, out, s);
        case 12: return pulse_poly_try<12>(h, sym, n_sym, out, s);
        case 16: return pulse_poly_try<16>(h, sym, n_sym, out, s);
        case 20: return pulse_poly_try<20>(h, sym, n_sym, out, s);
        case 32: return pulse_poly_try<32>(h, sym, n_sym, out, s);
        default: return false;
    }
}

static void free_pulse(comms_pulse* h) {
    (void)use_device(h->device);
    if (h->d_taps) (void)hipFree(h->d_taps);
    if (h->d_hist[0]) (void)hipFree(h->d_hist[0]);
    if (h->d_hist[1]) (void)hipFree(h->d_hist[1]);
    h->fini();
    delete h;
}

extern "C" {

comms_status_t comms_pulse_create(const comms_c32* taps, size_t n_taps, size_t sam_per_sym,
                                  int32_t device, comms_pulse_t** out) {
    COMMS_ARG(out != nullptr, "out is NULL");
    *out = nullptr;
    COMMS_ARG(taps != nullptr && n_taps > 0, "taps must hold at least one tap");
    COMMS_ARG(sam_per_sym >= 1, "sam_per_sym must be >= 1 (0 underflows in the reference, pulse.rs:88)");
    COMMS_ARG(n_taps <= 8192, "pulse shaping supports at most 8192 taps (got %zu)", n_taps);
    COMMS_ARG(sam_per_sym <= (1u << 20), "sam_per_sym too large");
    comms_pulse* h = new (std::nothrow) comms_pulse;
    COMMS_ARG(h != nullptr, "out of host memory");
    comms_status_t st = h->init(device);
    if (st != COMMS_OK) {
        delete h;
        return st;
    }
    h->n_taps = static_cast<int>(n_taps);
    h->sps = static_cast<int>(sam_per_sym);
    h->hist_len = static_cast<int>((n_taps + sam_per_sym - 1) / sam_per_sym);
    h->taps.assign(taps, taps + n_taps);
    h->real_taps = true;
    for (size_t k = 0; k < n_taps; ++k)
        if (taps[k].im != 0.0f) h->real_taps = false;
    hipError_t e = hipMalloc(&h->d_taps, n_taps * sizeof(float2));
    if (e == hipSuccess) e = hipMemcpy(h->d_taps, taps, n_taps * sizeof(float2), hipMemcpyHostToDevice);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        e = hipMalloc(&h->d_hist[i], h->hist_len * sizeof(float2));
        if (e == hipSuccess) e = zero_device(h->d_hist[i], h->hist_len * sizeof(float2));
    }
    if (e != hipSuccess) {
        free_pulse(h);
        return fail(COMMS_ERR_DEVICE, "pulse alloc: %s", hipGetErrorString(e));
    }
    *out = h;
    return COMMS_OK;
}

comms_status_t comms_pulse_run_dev(comms_pulse_t* h, const comms_c32* d_sym, size_t n_sym,
                                   comms_c32* d_out, void* stream) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG((d_sym && d_out) || !n_sym, "NULL device pointer");
    COMMS_TRY(use_device(h->device));
    if (!n_sym) return COMMS_OK;
    COMMS_ARG(n_sym <= SIZE_MAX / 8 / h->sps, "n_sym * sam_per_sym overflows");
    const size_t n_out = n_sym * h->sps;
    COMMS_ARG(!ranges_overlap(d_sym, n_sym * 8, d_out, n_out * (h->out_i16 ? 4 : 8)), "pulse shaping cannot run in place");
    hipStream_t s = nullptr;
    COMMS_TRY(h->enter(stream, &s));
    const float2* sym = reinterpret_cast<const float2*>(d_sym);
    if (!pulse_poly_launch(h, sym, n_sym, reinterpret_cast<float2*>(d_out), s)) {
        h->tic(s);
        size_t blocks = (n_out + 255) / 256;
        if (blocks > 8u * kNumCU) blocks = 8u * kNumCU;
        PulseMix mx{};
        mx.out_i16 = h->out_i16 ? 1 : 0;
        mx.out_scale = h->out_scale;
        if (h->mix) {
            mx.on = 1;
            mx.turns0 = h->turns;
            mx.frac = h->frac;
            mix_host_rotor(static_cast<uint64_t>(blocks) * 256u * h->frac, mx.sweep_c, mx.sweep_s);
        }
        pulse_kernel<<<dim3(static_cast<unsigned>(blocks)), dim3(256), h->n_taps * sizeof(float2), s>>>(
            sym, h->d_hist[h->cur], h->hist_len, h->d_taps, h->n_taps, h->sps,
            reinterpret_cast<float2*>(d_out), n_sym, h->d_hist[h->cur ^ 1], mx);
        h->toc(s);
    }
    COMMS_TRY(launch_ok("pulse kernel"));  // (workgroup 0 of the same launch advanced the history)
    h->cur ^= 1;
    if (h->mix) h->turns += static_cast<uint64_t>(n_out) * h->frac;
    return COMMS_OK;
}

// Fuses the MixerNode that follows a PulseNode into the same launch: every later run() returns
// Mixer::new(phase, dphase) applied to the shaped samples, the phase carried across calls.
comms_status_t comms_pulse_set_mixer(comms_pulse_t* h, double dphase, double phase) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG(std::isfinite(dphase) && std::isfinite(phase), "dphase/phase must be finite");
    h->mix = true;
    h->frac = mix_to_turns(mix_wrap_dphase(dphase));  // Mixer::new wraps dphase (src/mixer.rs:43-51)
    h->turns = mix_to_turns(phase);
    return COMMS_OK;
}

// The transmit chain straight into the wire format IQOutput writes (src/io/raw_iq.rs:173-178):
// every later run stores `(scale * y) as i16` pairs (examples/single_thread_bpsk.rs:40-44) instead of
// Complex<f32> -- 4 B per output sample instead of 8 written + 8 read + 4 written by a conversion pass.
comms_status_t comms_pulse_set_output_format(comms_pulse_t* h, int32_t format, float scale) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG(format == COMMS_IQ_C32 || format == COMMS_IQ_I16, "the pulse node writes Complex<f32> or i16 (got format %d)", format);
    COMMS_ARG(format != COMMS_IQ_I16 || std::isfinite(scale), "scale must be finite");
    h->out_i16 = format == COMMS_IQ_I16;
    h->out_scale = h->out_i16 ? scale : 1.0f;
    return COMMS_OK;
}

comms_status_t comms_pulse_get_phase(const comms_pulse_t* h, double* out_phase) {
    COMMS_ARG(h && out_phase, "NULL argument");
    COMMS_ARG(h->mix, "no mixer is fused into this pulse node");
    *out_phase = static_cast<double>(h->turns >> 11) * (comms::kMixT * 0x1.0p-53);
    return COMMS_OK;
}

comms_status_t comms_pulse_run(comms_pulse_t* h, const comms_c32* sym, size_t n_sym,
                               comms_c32* out) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG((sym && out) || !n_sym, "NULL host pointer");
    COMMS_TRY(use_device(h->device));
    if (!n_sym) return COMMS_OK;
    const size_t out_sym = static_cast<size_t>(h->sps) * (h->out_i16 ? 4 : sizeof(comms_c32));  // output bytes per symbol
    return h->run_host_units(sym, n_sym * sizeof(comms_c32), sizeof(comms_c32), out, n_sym * out_sym, out_sym, [&](void* d_in, void* d_out, size_t ib, size_t) {
        return comms_pulse_run_dev(h, static_cast<const comms_c32*>(d_in), ib / sizeof(comms_c32), static_cast<comms_c32*>(d_out), COMMS_STREAM_HANDLE);
    });
}

comms_status_t comms_pulse_set_timer(comms_pulse_t* h, comms_timer_t* t) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    h->timer = t;
    return COMMS_OK;
}

comms_status_t comms_pulse_destroy(comms_pulse_t* h) {
    if (!h) return COMMS_OK;
    free_pulse(h);
    return COMMS_OK;
}

}  // extern "C"
