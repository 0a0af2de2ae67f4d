// fir_decim_any.hip -- the fused chain in time domain at ANY decimation rate.
//
//   [mixer ->] FIR -> [mixer ->] keep every R-th [-> FM demod]       (DecimateNode::new(dec_rate) takes any rate:
//   src/util/resample_node.rs:23, :53-65; chain order as examples/fm_radio.rs:146-148)
// fir_decim_kernel (fir_decim.hip) is built per rate (2 ... 16: its LDS phase arrays and tap chunks are compile-
// time shapes); every other rate used to run the overlap-save launch, which filters at the full rate and throws
// R - 1 of R outputs away (~60 us at 2^24 samples whatever R is).  This kernel takes the rate as an argument.
//
// Shape: HALF A WAVE PER OUTPUT.  The 32 lanes of a half-wave split the taps of one output between them -- lane i
// holds taps i, i + 32, i + 64, ... in registers for the whole launch -- so for every step the half-wave reads 32
// CONSECUTIVE samples (newest first): one coalesced 256-byte run from global memory, or a conflict-free LDS read
// whatever R is (a lane-per-output layout strides by R and collides for even R).  The partial sums meet in a
// five-step cross-lane reduction (four DPP row steps and one v_permlane16_swap).  A first form of this kernel
// (round 2, one lane per output) lost to the overlap-save launch: one wave walked every tap of a tile while the
// other three idled, and small tiles at large R re-staged their halo rows again and again.
//   STAGED  (rate < ~48): the tile's samples go through LDS once (8 B per input sample from HBM, each reused by
//           taps / rate outputs);
//   DIRECT  (larger rates): windows barely overlap, every lane loads its samples itself -- at rate >= taps the
//           samples between the windows are never read at all.
// Taps may be complex; a mixer in FRONT of the FIR is folded into them by the caller
// (sum_k h[k] x[n-k] e^{i phi(n-k)} = e^{i phi(n)} sum_k (h[k] e^{-i k dphi}) x[n-k]), so the kernel only knows the
// mixer-behind-the-FIR form: one rotor per kept output.  FM demod runs over the tile's outputs in LDS (tiles
// overlap by one output, as in fir_decim_kernel).
#include <cmath>
#include <vector>

#include "common.hpp"
#include "fft_radix.hpp"
#include "fir_handle.hpp"

namespace comms {

struct AnyArgs {
    const void* in;      // n samples in format `fmt`
    const float2* hist;  // last hist_len RAW input samples before this call, time order
    float2* new_hist;
    void* out;           // float2 per kept output, or float with FM demod
    const float2* fm_prev;
    float2* fm_prev_new;
    const float2* taps;  // [32 * NT] complex taps, k ascending, zero beyond N
    size_t n, n_out, n_tiles;
    int hist_len, N, R, T;   // T = outputs per tile (a multiple of 8)
    int mode, fmt;
    float in_scale;
    uint64_t turns0, frac;
    double tile_c, tile_s;   // rotor of the step from one tile of a workgroup to its next
};

__device__ __forceinline__ void any_rotor_at(uint64_t turns, double& c, double& s) {
    sincos(static_cast<double>(turns >> 11) * (2.0 * 3.14159265358979323846264338327950288 * 0x1.0p-53), &s, &c);
}

// Sum over the 32 lanes of each half-wave, result in every lane of the half.  DPP: quad_perm [1,0,3,2] (0xB1),
// quad_perm [2,3,0,1] (0x4E), row_half_mirror (0x141), row_mirror (0x140) leave the 16-lane row sum in every lane of
// a row; v_permlane16_swap of the value with a copy of itself puts row 0 / row 2 in both rows of one register's half
// and row 1 / row 3 in the other's.
__device__ __forceinline__ float row_sum16(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false));
    return v;
}
__device__ __forceinline__ cf half_sum32(cf v) {
    const float x = row_sum16(v.x), y = row_sum16(v.y);
    const auto sx = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    const auto sy = __builtin_amdgcn_permlane16_swap(__float_as_uint(y), __float_as_uint(y), false, false);
    return cf{__uint_as_float(sx[0]) + __uint_as_float(sx[1]), __uint_as_float(sy[0]) + __uint_as_float(sy[1])};
}

// The same sums for TWO values per lane (the two outputs a half-wave works on per step) as a reduce-scatter: every exchange
// halves what a lane still carries -- lanes with bit 0 set keep output b, the others a (xor 1); lanes with bit 1 set keep the
// imaginary part (xor 2) -- so the remaining steps (the other three quads of the row: row_ror 4 and 8; the other row) move
// ONE float.  15 vector instructions for both outputs instead of 2 x 12; afterwards lane 0 of a half holds a, lane 1 holds b.
template <int CTRL>
__device__ __forceinline__ float any_dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ cf half_sum32_pair(cf a, cf b, int lane) {
    const bool odd = (lane & 1) != 0, hi = (lane & 2) != 0;
    float kx = odd ? b.x : a.x, ky = odd ? b.y : a.y;
    const float sx = odd ? a.x : b.x, sy = odd ? a.y : b.y;
    kx += any_dpp_mov<0xB1>(sx);          // the neighbour's value of the output this lane keeps
    ky += any_dpp_mov<0xB1>(sy);
    float k = hi ? ky : kx;
    const float s = hi ? kx : ky;
    k += any_dpp_mov<0x4E>(s);            // lane (bit 0 = output, bit 1 = component): the quad's sum
    k += any_dpp_mov<0x124>(k);           // row_ror 4: + the next quad
    k += any_dpp_mov<0x128>(k);           // row_ror 8: + the other two
    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(k), __float_as_uint(k), false, false);
    k = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);  // + the half-wave's other row
    return cf{k, any_dpp_mov<0x4E>(k)};   // lanes 0 / 1: {re, im} of output a / b
}

// acc += h * x for a complex tap h (num-complex form: re = hr xr - hi xi, im = hr xi + hi xr), or a real one
template <bool REAL>
__device__ __forceinline__ void any_mac(cf& acc, cf h, cf x) {
    // acc += h.re * x
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(h), "v"(x));
    // acc += (i * h.im) * x : acc.re -= h.im * x.im, acc.im += h.im * x.re
    if (!REAL) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(acc) : "v"(h), "v"(x));
}

constexpr int ANY_WG = 256;

template <int NT, bool REAL, bool STAGED>
__global__ __launch_bounds__(ANY_WG, 4) void fir_decim_any_kernel(const AnyArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* ys = reinterpret_cast<cf*>(smem);  // [T]  the tile's outputs (mixed), in order
    cf* rt = ys + a.T;                     // [T]  e^{i R t dphi}: the mixer's rotor of output t relative to the tile's first
    cf* xs = rt + a.T;                     // STAGED: the tile's samples, [PAD + (T - 1) R + N]
    switch (a.fmt) {
        case COMMS_IQ_I16: hist_advance(a.hist, InI16{static_cast<const short2*>(a.in), a.in_scale}, a.n, a.new_hist, a.hist_len); break;
        case COMMS_IQ_U8: hist_advance(a.hist, InU8{static_cast<const uchar2*>(a.in)}, a.n, a.new_hist, a.hist_len); break;
        default: hist_advance(a.hist, static_cast<const float2*>(a.in), a.n, a.new_hist, a.hist_len); break;
    }
    const int tid = threadIdx.x, i = tid & 31, hw = tid >> 5;  // lane of the half-wave, half-wave of the workgroup (0 .. 7)
    const int N = a.N, R = a.R, T = a.T;
    const bool post = (a.mode & COMMS_CHAIN_POST) != 0;
    const bool fm = (a.mode & COMMS_CHAIN_FM) != 0;
    const int ovl = fm ? 1 : 0;
    const long long ts = T - ovl;  // outputs a tile stores
    constexpr int PAD_TAPS = 32 * NT;
    const int pad = PAD_TAPS - N;  // front padding: the taps beyond N are zero, their samples only have to exist

    cf tp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) tp[t] = to_cf(a.taps[i + 32 * t]);
    if (post)
        for (int t = tid; t < T; t += ANY_WG) {
            double c, s;
            any_rotor_at(static_cast<uint64_t>(R) * static_cast<uint64_t>(t) * a.frac, c, s);
            rt[t] = cf{static_cast<float>(c), static_cast<float>(s)};
        }
    const size_t t0 = blockIdx.x;
    if (t0 >= a.n_tiles) return;
    double tt_c = 1.0, tt_s = 0.0;  // rotor of the tile's first output (f64, stepped per tile)
    if (post) any_rotor_at(a.turns0 + static_cast<uint64_t>(R * (static_cast<long long>(t0) * ts - ovl)) * a.frac, tt_c, tt_s);

    auto sample = [&](long long g) -> cf {  // raw sample g of [history | input | zeros], converted
        switch (a.fmt) {
            case COMMS_IQ_I16: return to_cf(stream_at(InI16{static_cast<const short2*>(a.in), a.in_scale}, a.hist, a.hist_len, g, a.n));
            case COMMS_IQ_U8: return to_cf(stream_at(InU8{static_cast<const uchar2*>(a.in)}, a.hist, a.hist_len, g, a.n));
            default: return to_cf(stream_at(static_cast<const float2*>(a.in), a.hist, a.hist_len, g, a.n));
        }
    };
    for (size_t tile = t0; tile < a.n_tiles; tile += gridDim.x) {
        const long long jb = static_cast<long long>(tile) * ts - ovl;  // first output this tile computes
        if (STAGED) {
            // samples R jb - (N - 1) ... R (jb + T - 1), at xs[pad + m]; at most 16 per lane, all requested before
            // the first is stored (one at a time every load's HBM latency showed: 73 -> ... us at rate 20)
            const long long s0 = static_cast<long long>(R) * jb - (N - 1);
            const int count = (T - 1) * R + N;
            cf tmp[16];
            if (s0 >= 0 && static_cast<size_t>(s0) + static_cast<size_t>(count) <= a.n) {
                // (buffer-addressed: one offset register for the sixteen loads; reads past `count` return zero)
                auto grab = [&](auto in) {
                    const BufRows<decltype(in)> br(in, static_cast<size_t>(s0), static_cast<size_t>(count));
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        tmp[u] = to_cf(br.get(tid * BufRows<decltype(in)>::E, u * ANY_WG * BufRows<decltype(in)>::E));
                };
                switch (a.fmt) {
                    case COMMS_IQ_I16: grab(InI16{static_cast<const short2*>(a.in), a.in_scale}); break;
                    case COMMS_IQ_U8: grab(InU8{static_cast<const uchar2*>(a.in)}); break;
                    default: grab(static_cast<const float2*>(a.in)); break;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int m = tid + ANY_WG * u;
                    if (m < count) xs[pad + m] = tmp[u];
                }
            } else {  // a tile at the stream's edge (history in front, or the end): one sample at a time
#pragma unroll 1
                for (int m = tid; m < count; m += ANY_WG) xs[pad + m] = sample(s0 + m);
            }
            if (tid < pad) xs[tid] = cf{0.f, 0.f};
        }
        __syncthreads();  // samples staged (and rt written; the previous tile's ys fully stored)
        const cf tt = cf{static_cast<float>(tt_c), static_cast<float>(tt_s)};
        // two outputs per half-wave and step (to, to + 8): their loads are in flight together (one at 16 taps per lane:
        // the registers)
        constexpr int U = NT <= 8 ? 2 : 1;
        for (int to = hw; to < T; to += 8 * U) {
            cf acc[U];
            cf xv[U][NT];
            bool live[U];
#pragma unroll
            for (int q = 0; q < U; ++q) {
                acc[q] = cf{0.f, 0.f};
                const int tq = to + 8 * q;
                const long long j = jb + tq;
                live[q] = tq < T && j >= 0 && j < static_cast<long long>(a.n_out);
                if (!live[q]) continue;
                // tap k = i + 32 t meets sample R j - k
                if (STAGED) {
                    const cf* p = xs + (tq * R + (N - 1) - i - 32 * (NT - 1) + pad);  // sample of tap i + 32 (NT - 1): the lowest address
#pragma unroll
                    for (int t = 0; t < NT; ++t) xv[q][t] = p[32 * (NT - 1 - t)];
                } else {
                    const long long g0 = static_cast<long long>(R) * j - i;
                    if (g0 - 32 * (NT - 1) - 31 + i >= 0) {  // (half-wave-uniform) the whole window is input
                        switch (a.fmt) {
                            case COMMS_IQ_I16: {
                                const InI16 in{static_cast<const short2*>(a.in), a.in_scale};
#pragma unroll
                                for (int t = 0; t < NT; ++t) xv[q][t] = to_cf(in[static_cast<size_t>(g0 - 32 * t)]);
                            } break;
                            case COMMS_IQ_U8: {
                                const InU8 in{static_cast<const uchar2*>(a.in)};
#pragma unroll
                                for (int t = 0; t < NT; ++t) xv[q][t] = to_cf(in[static_cast<size_t>(g0 - 32 * t)]);
                            } break;
                            default: {
                                const float2* in = static_cast<const float2*>(a.in);
#pragma unroll
                                for (int t = 0; t < NT; ++t) xv[q][t] = to_cf(in[static_cast<size_t>(g0 - 32 * t)]);
                            } break;
                        }
                    } else {
#pragma unroll
                        for (int t = 0; t < NT; ++t) xv[q][t] = sample(g0 - 32 * t);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < U; ++q) {
                if (live[q]) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) any_mac<REAL>(acc[q], tp[t], xv[q][t]);
                }
            }
            // (the rotor comes in the store pass: once per output, not per half-wave step)
            if constexpr (U == 2) {
                const cf r2 = half_sum32_pair(acc[0], acc[1], i);
                const int tq = to + 8 * i;   // lane 0: output `to`, lane 1: output `to + 8`
                if (i < 2 && tq < T) ys[tq] = r2;
            } else {
                const cf r1 = half_sum32(acc[0]);
                if (i == 0) ys[to] = r1;
            }
        }
        __syncthreads();  // ys complete; every read of xs is done
        auto mixed = [&](int to) -> float2 {  // output `to` of the tile behind the mixer (FM.prev of the previous call for y[-1])
            if (fm && jb + to < 0) return a.fm_prev[0];
            return post ? to_f2(cmulf(ys[to], cmulf(tt, rt[to]))) : to_f2(ys[to]);
        };
        for (int to = tid + ovl; to < T; to += ANY_WG) {
            const long long j = jb + to;
            if (j >= static_cast<long long>(a.n_out)) break;
            const float2 y = mixed(to);
            if (fm) {
                static_cast<float*>(a.out)[j] = fm_step(y, mixed(to - 1));
                if (j == static_cast<long long>(a.n_out) - 1) a.fm_prev_new[0] = y;
            } else {
                static_cast<float2*>(a.out)[j] = y;
            }
        }
        if (post) {  // the next tile of this workgroup
            const double nc = tt_c * a.tile_c - tt_s * a.tile_s;
            tt_s = tt_c * a.tile_s + tt_s * a.tile_c;
            tt_c = nc;
        }
    }
}

}  // namespace comms

using namespace comms;

namespace {

template <int NT>
comms_status_t launch_any(const AnyArgs& a, bool real, bool staged, unsigned blocks, size_t lds, hipStream_t s) {
#define COMMS_ANY_GO(REALV, STG)                                                                                           \
    do {                                                                                                                   \
        static DeviceOnce once;                                                                                            \
        if (once.need())                                                                                                   \
            COMMS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fir_decim_any_kernel<NT, REALV, STG>),         \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));                     \
        fir_decim_any_kernel<NT, REALV, STG><<<dim3(blocks), dim3(ANY_WG), lds, s>>>(a);                                    \
    } while (0)
    if (real) {
        if (staged) COMMS_ANY_GO(true, true); else COMMS_ANY_GO(true, false);
    } else {
        if (staged) COMMS_ANY_GO(false, true); else COMMS_ANY_GO(false, false);
    }
#undef COMMS_ANY_GO
    return launch_ok("fir_decim_any_kernel");
}

}  // namespace

extern "C" {

// Taps the any-rate kernel takes (lanes hold up to 16 taps each)
int32_t comms_fir_decim_any_supported(const comms_fir_t* h, uint32_t rate) {
    return h && h->n_eff >= 1 && h->n_eff <= 512 && rate >= 2 && rate <= (1u << 20) ? 1 : 0;
}

// The chain (mixer behind the FIR, or none: mode = COMMS_CHAIN_DEC [| POST] [| FM]) on the any-rate kernel.
comms_status_t comms_fir_run_decim_any_dev(comms_fir_t* h, const void* d_in, size_t n, void* d_out, int32_t mode,
                                           uint64_t turns0, uint64_t frac, uint32_t rate, const void* fm_prev,
                                           void* fm_prev_new, void* stream) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG((d_in && d_out) || !n, "NULL device pointer");
    COMMS_ARG(comms_fir_decim_any_supported(h, rate), "the any-rate chain kernel takes up to 512 taps and rates 2 ... 2^20");
    COMMS_ARG(n % rate == 0, "n must be a multiple of the decimation rate");
    COMMS_ARG((mode & COMMS_CHAIN_DEC) && !(mode & COMMS_CHAIN_PRE), "the any-rate kernel knows the mixer behind the FIR only");
    COMMS_TRY(fir_check_sticky(h));
    COMMS_TRY(use_device(h->device));
    if (!n) return COMMS_OK;
    // multiples of 8 up to 64: the polyphase frequency-domain kernel, every (rate / 8)-th output kept (fir_poly8.hip)
    if (comms_fir_poly8_supported(h, rate, mode, n) == 2)
        return comms_fir_run_poly8_dev(h, d_in, n, d_out, mode, turns0, frac, rate, fm_prev, fm_prev_new, stream);
    h->last_poly8 = false;
    const size_t in_elem = h->in_fmt == COMMS_IQ_I16 ? 4 : h->in_fmt == COMMS_IQ_U8 ? 2 : 8;
    COMMS_ARG(!ranges_overlap(d_in, n * in_elem, d_out, (n / rate) * ((mode & COMMS_CHAIN_FM) ? 4 : 8)),
              "the decimating chain cannot run in place");
    COMMS_ARG((reinterpret_cast<uintptr_t>(d_in) & (in_elem - 1)) == 0, "input must be aligned to one IQ sample");
    hipStream_t s = nullptr;
    COMMS_TRY(h->enter(stream, &s));
    const int N = h->n_eff, R = static_cast<int>(rate);
    const int NT = N <= 64 ? 2 : N <= 128 ? 4 : N <= 256 ? 8 : 16;
    // padded device copy of the taps, built once per handle
    if (!h->d_any_taps || h->any_nt != NT) {
        std::vector<float2> tp(32 * NT, make_float2(0.f, 0.f));
        for (int k = 0; k < N; ++k) tp[k] = make_float2(h->taps[k].re, h->taps[k].im);
        COMMS_TRY(h->quiesce());
        h->last_stream = s;  // (quiesce forgot the stream `enter` just recorded: the launch below must stay tracked)
        h->launched = true;
        if (h->d_any_taps) (void)hipFree(h->d_any_taps);
        h->d_any_taps = nullptr;
        COMMS_HIP_TRY(hipMalloc(&h->d_any_taps, tp.size() * sizeof(float2)));
        COMMS_HIP_TRY(hipMemcpy(h->d_any_taps, tp.data(), tp.size() * sizeof(float2), hipMemcpyHostToDevice));
        h->any_nt = NT;
    }
    const bool fm = (mode & COMMS_CHAIN_FM) != 0;
    // STAGED while neighbouring windows share most of their samples: below rate max(48, 0.4 taps) (255 taps: staged 33 us
    // against 42 direct at rate 64, a tie at 100; 127 and 63 taps: direct ahead from rate 64, a tie at 48 --
    // scripts/bench_chain_any.py, profiles/r03_bench_chain_rates.txt); COMMS_ANY_STAGED=0/1 forces one form
    static const int forced = diag_knob("COMMS_ANY_STAGED", -1);
    const int staged_below = 2 * N / 5 > 48 ? 2 * N / 5 : 48;
    bool staged = forced >= 0 ? forced != 0 : R < staged_below;
    AnyArgs a{};
    a.in = d_in;
    a.fmt = h->in_fmt;
    a.in_scale = h->in_scale;
    a.hist = h->d_hist[h->cur];
    a.new_hist = h->d_hist[h->cur ^ 1];
    a.out = d_out;
    a.fm_prev = static_cast<const float2*>(fm_prev);
    a.fm_prev_new = static_cast<float2*>(fm_prev_new);
    a.taps = h->d_any_taps;
    a.n = n;
    a.n_out = n / rate;
    a.hist_len = h->n_eff;
    a.N = N;
    a.R = R;
    a.mode = mode;
    a.turns0 = turns0;
    a.frac = frac;
    // outputs per tile: STAGED: what ~30 KiB of samples hold (at least 8); DIRECT: 256
    int T;
    if (staged) {
        const long long room = 3840 - 32 * NT;
        T = static_cast<int>(room / R) / 8 * 8;
        if (T < 8) staged = false;  // a tile of eight outputs does not fit: the windows are long past sharing
        if (T > 512) T = 512;
    }
    if (!staged) {
        // enough tiles for every CU to hold several workgroups (a workgroup walks its tile's outputs 16 at a time,
        // each step one HBM round trip deep): 92 -> ... us at rate 1000 with 256-output tiles
        const size_t want = (n / rate) / (16 * static_cast<size_t>(kNumCU));
        T = want >= 256 ? 256 : want >= 128 ? 128 : want >= 64 ? 64 : want >= 32 ? 32 : 16;
    }
    a.T = T;
    const size_t ts = static_cast<size_t>(T) - (fm ? 1 : 0);
    a.n_tiles = (a.n_out + ts - 1) / ts;
    const size_t lds = (2 * static_cast<size_t>(T) + (staged ? static_cast<size_t>(32 * NT) + static_cast<size_t>(T - 1) * R + 32 * NT : 0)) * sizeof(float2);
    const size_t per_cu = staged ? (lds > 40 * 1024 ? 3 : 4) : 8;
    const size_t slots = per_cu * kNumCU;
    const unsigned blocks = static_cast<unsigned>(a.n_tiles < slots ? a.n_tiles : slots);
    mix_host_rotor(static_cast<uint64_t>(R) * ts * blocks * frac, a.tile_c, a.tile_s);
    h->tic(s);
    comms_status_t st;
    switch (NT) {
        case 2: st = launch_any<2>(a, h->real_taps, staged, blocks, lds, s); break;
        case 4: st = launch_any<4>(a, h->real_taps, staged, blocks, lds, s); break;
        case 8: st = launch_any<8>(a, h->real_taps, staged, blocks, lds, s); break;
        default: st = launch_any<16>(a, h->real_taps, staged, blocks, lds, s); break;
    }
    h->toc(s);
    COMMS_TRY(st);
    h->cur ^= 1;
    return COMMS_OK;
}

}  // extern "C"
