// TRIAL, NOT PART OF THE BUILD (round 5).  It was wired into fir.hip's 1024-point dispatch in place of fir_os1024_dyn_kernel
// (fir_os1024_r32_run below; comms_fir had a d_r32 table pointer), passed the FIR parity tests against the oracle, and measured:
//   sixteen waves per CU (128 VGPRs): 180-220 B of scratch per lane -- the transposes hold 96 data registers + twiddles + addresses --
//     and 2 x SLOWER than fir_os1024_dyn_kernel (2^24 samples: 119 against 54 us);
//   twelve waves per CU (158 VGPRs, no scratch):  2^22 17.8 against 16.3 us, 2^24 54.7 against 47.5, 2^26 231 against 233.
// Fewer vector instructions (1060 packed operations per pair of segments, no lane swaps, against 2 x 664), but the thirty-two
// points per lane cost the fourth wave per SIMD, and the kernel it would replace needs that wave to keep its vector pipe 70 % busy.
// Kept for the record of the idea and its numbers (NOTES.md round 5).
// fir_r32.hip -- the 1024-point overlap-save FIR (batch_fir, src/filter/fir.rs:87-102; up to 257 taps) with the transform split
// 32 x 32: a HALF-wave per segment, thirty-two points per lane.
//
// fir_os1024_dyn_kernel (fir.hip) splits 1024 = 16 x 16 x 4: sixteen points per lane, two radix-16 layers in registers and a radix-4
// ACROSS lanes (v_permlane swaps), two twiddle layers.  At 2^24 samples -- a footprint the size of the Infinity Cache, where a plain
// copy runs at 6.8 TB/s -- that kernel is bound by its own vector work: 536 packed operations + 128 lane-swap instructions per
// segment, ~3800 cycles over four SIMDs (NOTES.md round 5).  Here a lane holds 32 points of ITS half-wave's segment (64 VGPRs; the
// kernel has them to spare at sixteen waves per CU), so both layers are radix-32 IN the lane (two radix-16 and sixteen W32
// butterflies: 220 packed operations per 32 points), there is ONE twiddle layer and no cross-lane arithmetic at all:
//     n = 32a + t:   DFT32 over a  ->  x W1024^{t k0}  ->  transpose (LDS)  ->  DFT32 over t   = X[k0 + 32 k1] in lane k0
// ~1070 packed operations per wave and PAIR of segments = 535 per segment, no swaps.  The transpose goes through the wave's 8.5-KB
// buffer in two rounds of sixteen registers (a half-wave's 1024 points would need twice the LDS a wave can have at sixteen per CU).
// Same ticket dealing, guarded first / last segments, history and raw-format load stage as fir_os1024_dyn_kernel.
#include <hip/hip_ext.h>

#include <cmath>
#include <vector>

#include "common.hpp"
#include "fft_radix.hpp"
#include "fir_handle.hpp"

namespace comms {

constexpr int R32_S = 66;             // row stride of the transpose buffer (2 mod 32: sixteen readers of a half on distinct banks)
constexpr int R32_BUF = 16 * R32_S;   // per-wave buffer, in cf
#ifndef COMMS_R32_WPB
#define COMMS_R32_WPB 12
#endif
constexpr int R32_WPB = COMMS_R32_WPB;  // waves per workgroup (one workgroup per CU)
constexpr size_t R32_LDS_BYTES = (2048 + R32_WPB * R32_BUF) * sizeof(float2) + 16;

__device__ __forceinline__ void r32_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// forward W32^K = (cos(2 pi K / 32), -sin(2 pi K / 32)), K = 1 ... 15
template <int K>
__device__ __forceinline__ cf w32() {
    constexpr float C[16] = {1.f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f, 0.70710678118654752440f,
                             0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f, 0.f, -0.19509032201612826785f,
                             -0.38268343236508977173f, -0.55557023301960222474f, -0.70710678118654752440f, -0.83146961230254523708f,
                             -0.92387953251128675613f, -0.98078528040323044913f};
    constexpr float S[16] = {0.f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f, 0.70710678118654752440f,
                             0.83146961230254523708f, 0.92387953251128675613f, 0.98078528040323044913f, 1.f, 0.98078528040323044913f,
                             0.92387953251128675613f, 0.83146961230254523708f, 0.70710678118654752440f, 0.55557023301960222474f,
                             0.38268343236508977173f, 0.19509032201612826785f};
    return cf{C[K], -S[K]};
}

template <int DIR, int K>
__device__ __forceinline__ void r32_step(cf& E, cf& O) {
    if constexpr (K == 0) {
        const cf t = O;
        O = csub(E, t);
        E = cadd(E, t);
    } else if constexpr (K == 8) {  // W32^8 = -+i
        const cf t = O;
        O = csub_di<DIR>(E, t);
        E = cadd_di<DIR>(E, t);
    } else {
        const cf t = tw_mul_s<DIR>(O, w32<K>());
        O = csub(E, t);
        E = cadd(E, t);
    }
}
// 32-point DFT in the lane.  In: e[j] = x[2j], o[j] = x[2j + 1].  Out: X[k] in e[R16_POS(k)], X[k + 16] in o[R16_POS(k)], k = 0 ... 15.
template <int DIR>
__device__ __forceinline__ void radix32(cf (&e)[16], cf (&o)[16]) {
    radix16<DIR>(e);
    radix16<DIR>(o);
#define COMMS_R32(K) r32_step<DIR, K>(e[R16_POS(K)], o[R16_POS(K)]);
    COMMS_R32(0) COMMS_R32(1) COMMS_R32(2) COMMS_R32(3) COMMS_R32(4) COMMS_R32(5) COMMS_R32(6) COMMS_R32(7)
    COMMS_R32(8) COMMS_R32(9) COMMS_R32(10) COMMS_R32(11) COMMS_R32(12) COMMS_R32(13) COMMS_R32(14) COMMS_R32(15)
#undef COMMS_R32
}
// the value of index k (0 ... 31) after radix32
#define R32_AT(e, o, k) ((k) < 16 ? (e)[R16_POS((k) & 15)] : (o)[R16_POS((k) & 15)])

// Transpose within each half-wave: lane t holds u(k, t) for k = 0 ... 31 at R32_AT(e, o, k) (times the twiddle tw[k][t], conjugated
// for DIR = +1); afterwards lane k holds u(k, t) for t = 0 ... 31 in natural order: ne[j] = u(k, 2j), no[j] = u(k, 2j + 1).
template <int DIR>
__device__ __forceinline__ void r32_transpose(const cf (&e)[16], const cf (&o)[16], cf (&ne)[16], cf (&no)[16], cf* lds, const cf* tw, int l) {
    const int t = l & 31, hoff = l & 32;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        // (four twiddles in flight at a time: in the second round the lane also holds the first round's thirty-two results)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int kk = 4 * g; kk < 4 * g + 4; ++kk) {
                const int k = 16 * r + kk;
                cf x = r ? o[R16_POS(kk)] : e[R16_POS(kk)];
                if (k) x = tw_mul<DIR>(x, tw[k * 32 + t]);
                lds[kk * R32_S + l] = x;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        r32_sync();
        if ((t >> 4) == r) {
            const cf* row = lds + (t & 15) * R32_S + hoff;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                ne[j] = row[2 * j];
                no[j] = row[2 * j + 1];
            }
        }
        r32_sync();
    }
}

// One pair of segments (one per half-wave): e[j] = row 2j, o[j] = row 2j + 1 of the lane's segment (sample 32 a + t of it) ->
// the filtered rows, row a at R32_AT(e, o, a).
__device__ __forceinline__ void os1024_r32_core(cf (&e)[16], cf (&o)[16], cf* lds, const cf* tw, const cf* hsp, int l) {
    const int t = l & 31;
    cf e2[16], o2[16];
    radix32<-1>(e, o);
    r32_transpose<-1>(e, o, e2, o2, lds, tw, l);
    radix32<-1>(e2, o2);                               // lane k0: X[k0 + 32 k1] at R32_AT(e2, o2, k1)
    // x H (eight at a time: all thirty-two table values in flight at once would not fit beside the data), then into natural
    // order over k1 for the inverse layer (a renaming: every index is a constant)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int k = 8 * g; k < 8 * g + 8; ++k) e2[R16_POS(k)] = cmulf(e2[R16_POS(k)], hsp[k * 32 + t]);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int k = 8 * g; k < 8 * g + 8; ++k) o2[R16_POS(k)] = cmulf(o2[R16_POS(k)], hsp[(16 + k) * 32 + t]);
        __builtin_amdgcn_sched_barrier(0);
    }
    cf ye[16], yo[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        ye[j] = R32_AT(e2, o2, 2 * j);
        yo[j] = R32_AT(e2, o2, 2 * j + 1);
    }
    radix32<1>(ye, yo);                                // lane k0: index t' at R32_AT(ye, yo, t')
    r32_transpose<1>(ye, yo, e, o, lds, tw, l);        // lane t: values over k0, natural order
    radix32<1>(e, o);                                  // row a at R32_AT(e, o, a)
}

template <int HR, class In>
__global__ __launch_bounds__(64 * R32_WPB, (R32_WPB + 3) / 4) void fir_os1024_r32_kernel(In in, const float2* __restrict__ hist, int hist_len,
                                                                 float2* __restrict__ out, size_t n, const cf* __restrict__ twg,
                                                                 const cf* __restrict__ hg, float2* __restrict__ new_hist,
                                                                 unsigned chunk_log2, KStamp ks) {
    constexpr int WVK = 1024 - 64 * HR, HALO = 64 * HR, R0 = 2 * HR;  // new samples per segment; first valid row of 32
    extern __shared__ __attribute__((aligned(16))) char smem[];
    kstamp_begin(ks);
    hist_advance(hist, in, n, new_hist, hist_len);
    cf* tw = reinterpret_cast<cf*>(smem);   // [32][32]  W1024^{t k}
    cf* hsp = tw + 1024;                    // [32][32]  H[k0 + 32 k1] / 1024 at [k1][k0]
    const int l = threadIdx.x & 63, t = l & 31, half = l >> 5;
    const int wave = threadIdx.x >> 6;
    cf* lds = hsp + 1024 + wave * R32_BUF;
    unsigned* ticket = reinterpret_cast<unsigned*>(hsp + 1024 + R32_WPB * R32_BUF);
    for (int i = threadIdx.x; i < 1024; i += 64 * R32_WPB) {
        tw[i] = twg[i];
        hsp[i] = hg[i];
    }
    if (threadIdx.x == 0) *ticket = 0;
    __syncthreads();

    // interior segments 1 ... nfull - 1 in PAIRS: pair p = segments 1 + 2p, 2 + 2p (the second may fall off the end)
    const size_t nfull = n / WVK;
    const size_t inner = nfull > 1 ? nfull - 1 : 0;
    const size_t npairs = (inner + 1) / 2;
    const bool contiguous = chunk_log2 >= 32u;
    const unsigned G = gridDim.x;
    const unsigned wg = (G % 8u == 0u) ? (blockIdx.x % 8u) * (G / 8u) + blockIdx.x / 8u : blockIdx.x;
    const size_t lo = blockIdx.x * npairs / gridDim.x;
    const size_t hi = contiguous ? (blockIdx.x + 1) * npairs / gridDim.x : npairs;
    auto draw = [&]() -> size_t {
        unsigned tk0 = 0;
        if (l == 0) tk0 = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const unsigned tk = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(tk0)));
        if (contiguous) return lo + tk;
        const size_t c = static_cast<size_t>(tk >> chunk_log2) * G + wg;
        return (c << chunk_log2) + (tk & ((1u << chunk_log2) - 1u));
    };

    cf e[16], o[16];
    // The stream's first segment (halo from the history) and its partial last one: the two halves of wave 0 of the first workgroup
    if (wave == 0 && blockIdx.x == 0) {
        const size_t nseg = (n + WVK - 1) / WVK;
        const bool have_last = nseg >= 2 && nseg != nfull;
        const size_t sg = half ? nseg - 1 : 0;
        const bool on = half ? have_last : true;
        const long long b0 = static_cast<long long>(sg * WVK) - HALO + t;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            e[j] = on ? to_cf(stream_at(in, hist, hist_len, b0 + 32 * (2 * j), n)) : cf{0.f, 0.f};
            o[j] = on ? to_cf(stream_at(in, hist, hist_len, b0 + 32 * (2 * j + 1), n)) : cf{0.f, 0.f};
        }
        os1024_r32_core(e, o, lds, tw, hsp, l);
        if (on) {
#pragma unroll
            for (int a = R0; a < 32; ++a) {
                const size_t i = sg * WVK + 32 * (a - R0) + t;
                if (i < n) out[i] = to_f2(R32_AT(e, o, a));
            }
        }
    }
    size_t pr = draw();
    while (pr < hi) {
        const size_t sg = 1 + 2 * pr + half;
        const bool on = sg < nfull;  // (an odd count of interior segments: the last pair's second half idles)
        const size_t p = (on ? sg : 1) * WVK - HALO + t;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            e[j] = to_cf(in[p + 32 * (2 * j)]);
            o[j] = to_cf(in[p + 32 * (2 * j + 1)]);
        }
        const size_t pr_next = draw();
        os1024_r32_core(e, o, lds, tw, hsp, l);
        if (on) {
            float2* dst = out + sg * WVK + t;
#pragma unroll
            for (int a = R0; a < 32; ++a) dst[32 * (a - R0)] = to_f2(R32_AT(e, o, a));
        }
        pr = pr_next;
    }
    kstamp_end(ks);
}

}  // namespace comms

using namespace comms;

namespace {

comms_status_t r32_prepare(comms_fir* h) {
    if (h->d_r32) return COMMS_OK;
    const double kPi = 3.14159265358979323846264338327950288;
    std::vector<float2> tb(2048);
    for (int k = 0; k < 32; ++k)
        for (int t = 0; t < 32; ++t) {
            const double a = -2.0 * kPi * static_cast<double>((k * t) % 1024) / 1024.0;
            tb[k * 32 + t] = make_float2(static_cast<float>(std::cos(a)), static_cast<float>(std::sin(a)));
        }
    // H[k] = sum_j h[j] W1024^{jk}, / 1024 (the unnormalised inverse transform)
    std::vector<double> cs(1024), sn(1024);
    for (int i = 0; i < 1024; ++i) {
        cs[i] = std::cos(2.0 * kPi * i / 1024.0);
        sn[i] = -std::sin(2.0 * kPi * i / 1024.0);
    }
    for (int k = 0; k < 1024; ++k) {
        double re = 0.0, im = 0.0;
        for (int j = 0; j < h->n_eff; ++j) {
            const int ex = (j * k) & 1023;
            const double tr = h->taps[j].re, ti = h->taps[j].im;
            re += tr * cs[ex] - ti * sn[ex];
            im += tr * sn[ex] + ti * cs[ex];
        }
        tb[1024 + (k >> 5) * 32 + (k & 31)] = make_float2(static_cast<float>(re / 1024.0), static_cast<float>(im / 1024.0));
    }
    COMMS_HIP_TRY(hipMalloc(&h->d_r32, tb.size() * sizeof(float2)));
    COMMS_HIP_TRY(hipMemcpy(h->d_r32, tb.data(), tb.size() * sizeof(float2), hipMemcpyHostToDevice));
    return COMMS_OK;
}

template <int HR, class In>
comms_status_t r32_launch(comms_fir* h, hipStream_t s, In in, float2* o, size_t n, float2* nh, unsigned chunk_log2, hipEvent_t ea,
                          hipEvent_t eb, KStamp ks) {
    static DeviceOnce attr_once;
    if (attr_once.need())
        COMMS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fir_os1024_r32_kernel<HR, In>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(R32_LDS_BYTES)));
    const size_t nseg = (n + (1024 - 64 * HR) - 1) / (1024 - 64 * HR);
    const size_t want = (nseg / 2 + R32_WPB - 1) / R32_WPB;
    const dim3 grid(static_cast<unsigned>(want < 1 ? 1 : want < static_cast<size_t>(kNumCU) ? want : kNumCU));
    const cf* tw = reinterpret_cast<const cf*>(h->d_r32);
    if (ea)
        hipExtLaunchKernelGGL((fir_os1024_r32_kernel<HR, In>), grid, dim3(64 * R32_WPB), static_cast<uint32_t>(R32_LDS_BYTES), s, ea, eb, 0u, in,
                              h->d_hist[h->cur], h->n_eff, o, n, tw, tw + 1024, nh, chunk_log2, ks);
    else
        fir_os1024_r32_kernel<HR, In><<<grid, dim3(64 * R32_WPB), R32_LDS_BYTES, s>>>(in, h->d_hist[h->cur], h->n_eff, o, n, tw, tw + 1024, nh,
                                                                             chunk_log2, ks);
    return launch_ok("fir_os1024_r32_kernel");
}

template <class In>
comms_status_t r32_launch_hr(int hr, comms_fir* h, hipStream_t s, In in, float2* o, size_t n, float2* nh, unsigned chunk_log2,
                             hipEvent_t ea, hipEvent_t eb, KStamp ks) {
    switch (hr) {
        case 1: return r32_launch<1>(h, s, in, o, n, nh, chunk_log2, ea, eb, ks);
        case 2: return r32_launch<2>(h, s, in, o, n, nh, chunk_log2, ea, eb, ks);
        case 3: return r32_launch<3>(h, s, in, o, n, nh, chunk_log2, ea, eb, ks);
        default: return r32_launch<4>(h, s, in, o, n, nh, chunk_log2, ea, eb, ks);
    }
}

}  // namespace

namespace comms {

// The launch fir.hip makes in place of fir_os1024_dyn_kernel (same arguments; the caller has entered the stream and flips h->cur)
comms_status_t fir_os1024_r32_run(comms_fir* h, hipStream_t s, const void* d_in, float2* o, size_t n, float2* nh, int hr,
                                  unsigned chunk_log2, hipEvent_t ea, hipEvent_t eb, KStamp ks) {
    COMMS_TRY(r32_prepare(h));
    if (h->in_fmt == COMMS_IQ_I16)
        return r32_launch_hr(hr, h, s, InI16{static_cast<const short2*>(d_in), h->in_scale}, o, n, nh, chunk_log2, ea, eb, ks);
    if (h->in_fmt == COMMS_IQ_U8) return r32_launch_hr(hr, h, s, InU8{static_cast<const uchar2*>(d_in)}, o, n, nh, chunk_log2, ea, eb, ks);
    return r32_launch_hr(hr, h, s, static_cast<const float2*>(d_in), o, n, nh, chunk_log2, ea, eb, ks);
}

}  // namespace comms
