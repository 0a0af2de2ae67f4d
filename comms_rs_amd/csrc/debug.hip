// debug.hip -- memory-pattern probes used while tuning (not part of the public ABI).
// Compiled only into the diagnostic build (`make diag` -> lib/libcomms_hip_diag.so, -DCOMMS_DIAG);
// the product library exports exactly include/comms_hip.h.
#ifdef COMMS_DIAG
#include "common.hpp"

namespace comms {

// mode 0: float4 grid-stride copy; mode 1: float2 grid-stride copy
template <typename V>
__global__ __launch_bounds__(256) void probe_copy_kernel(const V* __restrict__ in, V* __restrict__ out, size_t n) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = in[i];
}

// modes 9 / 10: pure fill (no reads) with 8 / 16 bytes per lane -- what a write-only kernel (upsample's zeros) can reach
template <typename V>
__global__ __launch_bounds__(256) void probe_fill_kernel(V* __restrict__ out, size_t n, V val) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = val;
}

// mode 2: the os1024 access pattern without the math: a wave owns a run of 768-sample
// tiles; per tile 12 x (64 lanes x 8 B) loads, then 12 x 512-B stores.
typedef float nt_f2 __attribute__((ext_vector_type(2)));
// NT bit 0: nontemporal loads, bit 1: nontemporal stores (modes 6 / 7 / 8)
template <int TPL, int NT = 0>
__global__ __launch_bounds__(256) void probe_tile_kernel(const float2* __restrict__ in, float2* __restrict__ out,
                                                         size_t ntiles, size_t n_runs) {
    const int l = threadIdx.x & 63;
    const size_t run = static_cast<size_t>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (run >= n_runs) return;
    const size_t t0 = run * ntiles / n_runs, t1 = (run + 1) * ntiles / n_runs;
    for (size_t t = t0; t < t1; ++t) {
        float2 v[TPL];
        const size_t base = t * (64 * TPL);
#pragma unroll
        for (int a = 0; a < TPL; ++a) {
            if (NT & 1) {
                const nt_f2 q = __builtin_nontemporal_load(reinterpret_cast<const nt_f2*>(&in[base + 64 * a + l]));
                v[a] = make_float2(q.x, q.y);
            }
            else v[a] = in[base + 64 * a + l];
        }
#pragma unroll
        for (int a = 0; a < TPL; ++a) {
            if (NT & 2) {
                nt_f2 q;
                q.x = v[a].x;
                q.y = v[a].y;
                __builtin_nontemporal_store(q, reinterpret_cast<nt_f2*>(&out[base + 64 * a + l]));
            }
            else out[base + 64 * a + l] = v[a];
        }
    }
}

// modes 100 + K: tiles of K x 512 B dealt round-robin over the waves (wave g: tiles g, g + W, ...) instead of a
// contiguous run of tiles per wave
template <int TPL>
__global__ __launch_bounds__(256) void probe_tile_rr_kernel(const float2* __restrict__ in, float2* __restrict__ out,
                                                            size_t ntiles, size_t n_waves) {
    const int l = threadIdx.x & 63;
    const size_t g = static_cast<size_t>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    for (size_t t = g; t < ntiles; t += n_waves) {
        float2 v[TPL];
        const size_t base = t * (64 * TPL);
#pragma unroll
        for (int a = 0; a < TPL; ++a) v[a] = in[base + 64 * a + l];
#pragma unroll
        for (int a = 0; a < TPL; ++a) out[base + 64 * a + l] = v[a];
    }
}

// mode 3: same tiles but 16 B per lane (float4): 6 x 1-KiB loads / stores per tile
__global__ __launch_bounds__(256) void probe_tile4_kernel(const float4* __restrict__ in, float4* __restrict__ out,
                                                          size_t ntiles, size_t n_runs) {
    const int l = threadIdx.x & 63;
    const size_t run = static_cast<size_t>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (run >= n_runs) return;
    const size_t t0 = run * ntiles / n_runs, t1 = (run + 1) * ntiles / n_runs;
    for (size_t t = t0; t < t1; ++t) {
        float4 v[6];
        const size_t base = t * (64 * 6);
#pragma unroll
        for (int a = 0; a < 6; ++a) v[a] = in[base + 64 * a + l];
#pragma unroll
        for (int a = 0; a < 6; ++a) out[base + 64 * a + l] = v[a];
    }
}

// modes 4/5: the four-step FFT passes' access patterns for N = 2^20 (1024 x 1024) without the
// math: one 16-wave workgroup per CU owns 16-transform tiles, the next tile's 16 loads per lane
// are requested before this tile's 16 stores.  ROWIN 0: 16 adjacent columns in, same places
// out (128-B chunks at 8-KiB stride both ways); ROWIN 1: 16 adjacent rows in (256-B runs),
// transposed out (128-B chunks).
template <int ROWIN>
__global__ __launch_bounds__(1024) void probe_fft_tile_kernel(const float2* __restrict__ in, float2* __restrict__ out,
                                                              size_t n_tiles) {
    const unsigned tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    unsigned in_off, out_off;
    if (ROWIN) {
        const unsigned c = 2 * (w & 7) + (l >> 5), r0 = 32 * (w >> 3) + (l & 31);
        in_off = c * 1024 + r0;                      // + 64 a
        out_off = (w + 16 * (l >> 4)) * 1024 + (l & 15);  // + (64 i + 256 k2) * 1024
    } else {
        const unsigned c = tid & 15, r0 = tid >> 4;
        in_off = r0 * 1024 + c;  // + 64 a * 1024
        out_off = (w + 16 * (l >> 4)) * 1024 + (l & 15);
    }
    auto src = [&](size_t t) { return in + (t >> 6) * (1u << 20) + (t & 63) * (ROWIN ? 16 * 1024 : 16); };
    float2 x[16];
    if (blockIdx.x < n_tiles) {
        const float2* sp = src(blockIdx.x);
#pragma unroll
        for (int a = 0; a < 16; ++a) x[a] = sp[in_off + a * (ROWIN ? 64u : 65536u)];
    }
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        float2 v[16];
#pragma unroll
        for (int a = 0; a < 16; ++a) v[a] = x[a];
        if (t + gridDim.x < n_tiles) {
            const float2* sp = src(t + gridDim.x);
#pragma unroll
            for (int a = 0; a < 16; ++a) x[a] = sp[in_off + a * (ROWIN ? 64u : 65536u)];
        }
        float2* dp = out + (t >> 6) * (1u << 20) + (t & 63) * 16;
#pragma unroll
        for (int a = 0; a < 16; ++a) dp[out_off + a * 65536u] = v[a];
    }
}


// ---- read-only probes (round 5): what the decimating chain's INPUT side can reach without its arithmetic.
// Grid-stride reads (V = float2 / float4), summed; the sum leaves only if it hits a value it never has.
template <typename V>
__global__ __launch_bounds__(256) void probe_read_kernel(const V* __restrict__ in, float* __restrict__ sink, size_t n) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    float acc = 0.f;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) acc += in[i].x;
    if (acc == 12345.678f) sink[0] = acc;
}
// fir_decim_kernel<8, 2>'s tile traffic: a 256-lane workgroup reads tiles of 16 rows x 256 samples (+ half a halo row)
// dealt round-robin over a persistent grid.  FLAGS bit 0: the next tile is requested before this one is consumed;
// bit 1: the tile goes through LDS as 16 phase arrays with the kernel's two barriers; bit 2: nontemporal loads;
// bit 3: 16 bytes per lane (8 rows of 4 KiB); bit 4: one 4-byte output per two lanes' worth of tile is stored
// (the FM chain's 1/16 of the input bytes).
template <int FLAGS>
__global__ __launch_bounds__(256, 4) void probe_read_tile_kernel(const float2* __restrict__ in, float* __restrict__ sink,
                                                                 float* __restrict__ out, size_t n_tiles) {
    constexpr bool PF = FLAGS & 1, LDSST = (FLAGS & 2) != 0, NT = (FLAGS & 4) != 0, WIDE = (FLAGS & 8) != 0, ST = (FLAGS & 16) != 0;
    constexpr int S = 273;
    __shared__ float2 sh[LDSST ? 16 * S : 1];
    const int tid = threadIdx.x;
    float acc = 0.f;
    constexpr int NR = WIDE ? 8 : 16;
    typedef float nt_f4 __attribute__((ext_vector_type(4)));
    float4 xa[WIDE ? NR : 1], xb[WIDE ? NR : 1];
    float2 ya[WIDE ? 1 : NR + 1], yb[WIDE ? 1 : NR + 1];
    auto fetch = [&](size_t t, auto& xv, auto& yv) {
        const float2* base = in + t * 4096;
        if constexpr (WIDE) {
#pragma unroll
            for (int m = 0; m < NR; ++m) {
                if (NT) {
                    const nt_f4 q = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(base) + tid + 256 * m);
                    xv[m] = make_float4(q.x, q.y, q.z, q.w);
                } else
                    xv[m] = reinterpret_cast<const float4*>(base)[tid + 256 * m];
            }
        } else {
#pragma unroll
            for (int m = 0; m < NR; ++m) {
                if (NT) {
                    const nt_f2 q = __builtin_nontemporal_load(reinterpret_cast<const nt_f2*>(base) + tid + 256 * m);
                    yv[m] = make_float2(q.x, q.y);
                } else
                    yv[m] = base[tid + 256 * m];
            }
            yv[NR] = (t > 0 && tid < 128) ? base[tid - 128] : make_float2(0.f, 0.f);  // halo: the last 128 samples of the tile before
        }
    };
    auto consume = [&](size_t t, auto& xv, auto& yv) {
        if constexpr (LDSST && !WIDE) {
#pragma unroll
            for (int m = 0; m < NR; ++m) {
                const unsigned s = static_cast<unsigned>(tid + 256 * m);
                sh[(s & 15) * S + 8 + (s >> 4)] = yv[m];
            }
            if (tid < 128) sh[(tid & 15) * S + (tid >> 4)] = yv[NR];
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            acc += sh[(tid & 15) * S + 8 + (tid >> 4)].x;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else if constexpr (WIDE) {
#pragma unroll
            for (int m = 0; m < NR; ++m) acc += xv[m].x + xv[m].z;
        } else {
#pragma unroll
            for (int m = 0; m <= NR; ++m) acc += yv[m].x;
        }
        if (ST) {
            out[t * 512 + tid] = acc;
            out[t * 512 + 256 + tid] = acc + 1.f;
        }
    };
    size_t t = blockIdx.x;
    if (PF) {
        if (t < n_tiles) fetch(t, xa, ya);
        for (; t < n_tiles; t += 2 * static_cast<size_t>(gridDim.x)) {
            const size_t t2 = t + gridDim.x, t3 = t2 + gridDim.x;
            if (t2 < n_tiles) fetch(t2, xb, yb);
            consume(t, xa, ya);
            if (t3 < n_tiles) fetch(t3, xa, ya);
            if (t2 < n_tiles) consume(t2, xb, yb);
        }
    } else {
        for (; t < n_tiles; t += gridDim.x) {
            fetch(t, xa, ya);
            consume(t, xa, ya);
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}


// Wave-private streaming (the shape of a barrier-free decimating chain): a wave reads tiles of 16 rows x 64 samples
// (512 B per instruction, 8 KiB per tile), `nt_chunk` consecutive tiles per chunk, chunks dealt round-robin over
// all the waves of a persistent grid; the next tile is requested before this one is consumed.  FLAGS bit 0:
// nontemporal loads; bit 1: 512 B stored per wave and tile (an FM chain's output at rate 8); bit 2: 16 B per lane.
template <int FLAGS>
__global__ __launch_bounds__(256, 4) void probe_read_wave_kernel(const float2* __restrict__ in, float* __restrict__ sink,
                                                                 float2* __restrict__ out, size_t n_tiles, int nt_chunk, int st_mode) {
    constexpr bool NT = FLAGS & 1, ST = (FLAGS & 2) != 0, WIDE = (FLAGS & 4) != 0;
    const int l = threadIdx.x & 63;
    const size_t wave = static_cast<size_t>(blockIdx.x) * 4 + (threadIdx.x >> 6), n_waves = static_cast<size_t>(gridDim.x) * 4;
    const size_t n_chunks = (n_tiles + nt_chunk - 1) / nt_chunk;
    float acc = 0.f;
    float2 xa[16], xb[16];
    auto fetch = [&](size_t t, float2 (&x)[16]) __attribute__((always_inline)) {
        const float2* base = in + t * 1024;
        if constexpr (WIDE) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                typedef float nt_f4 __attribute__((ext_vector_type(4)));
                nt_f4 q;
                if (NT) q = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(base) + l + 64 * m);
                else { const float4 v = reinterpret_cast<const float4*>(base)[l + 64 * m]; q = nt_f4{v.x, v.y, v.z, v.w}; }
                x[2 * m] = make_float2(q.x, q.y);
                x[2 * m + 1] = make_float2(q.z, q.w);
            }
        } else {
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                if (NT) {
                    const nt_f2 q = __builtin_nontemporal_load(reinterpret_cast<const nt_f2*>(base) + l + 64 * m);
                    x[m] = make_float2(q.x, q.y);
                } else
                    x[m] = base[l + 64 * m];
            }
        }
    };
    auto consume = [&](size_t t, const float2 (&x)[16]) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 16; ++m) acc += x[m].x;
        if (ST) {
            const float2 v = make_float2(acc, acc + 1.f);
            float2* q = out + t * 64 + l;
            if (st_mode == 0) *q = v;
            else if (st_mode == 1) {
                nt_f2 w; w.x = v.x; w.y = v.y;
                __builtin_nontemporal_store(w, reinterpret_cast<nt_f2*>(q));
            } else if (st_mode == 2) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" : : "v"(q), "v"(v) : "memory");
            else if (st_mode == 3) asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(q), "v"(v) : "memory");
            else if (st_mode == 4) asm volatile("global_store_dwordx2 %0, %1, off sc0" : : "v"(q), "v"(v) : "memory");
            else if (st_mode == 5) out[(t & 1023) * 64 + l] = v;   // a 512-KiB window: what the store path costs without HBM writes
            else if (st_mode == 6) { if ((t & 7) == 7) { for (int i = 0; i < 8; ++i) out[(t - 7 + i) * 64 + l] = v; } }  // eight tiles' outputs in one burst
        }
    };
    if (st_mode >= 16) {  // staggered: the wave walks its chunk from tile (wave * mul) % nt_chunk on and wraps (no two waves' phases alike)
        const int mul = st_mode >> 4;
        st_mode &= 15;
        for (size_t c = wave; c < n_chunks; c += n_waves) {
            const size_t t0 = c * nt_chunk;
            const size_t s0 = (wave * mul) % nt_chunk;
            fetch(t0 + s0, xa);
            for (int i = 0; i < nt_chunk; i += 2) {
                const size_t ta = t0 + (s0 + i) % nt_chunk, tb = t0 + (s0 + i + 1) % nt_chunk, tc = t0 + (s0 + i + 2) % nt_chunk;
                if (i + 1 < nt_chunk) fetch(tb, xb);
                consume(ta, xa);
                if (i + 2 < nt_chunk) fetch(tc, xa);
                if (i + 1 < nt_chunk) consume(tb, xb);
            }
        }
    } else
    for (size_t c = wave; c < n_chunks; c += n_waves) {
        const size_t t0 = c * nt_chunk, t1 = t0 + nt_chunk < n_tiles ? t0 + nt_chunk : n_tiles;
        fetch(t0, xa);
        for (size_t t = t0; t < t1; t += 2) {
            if (t + 1 < t1) fetch(t + 1, xb);
            consume(t, xa);
            if (t + 2 < t1) fetch(t + 2, xa);
            if (t + 1 < t1) consume(t + 1, xb);
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

}  // namespace comms

using namespace comms;

extern "C" comms_status_t comms_debug_copy(const void* d_in, void* d_out, size_t n_c32, int mode, int waves_per_cu,
                                           void* stream) {
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (mode == 0) {
        probe_copy_kernel<float4><<<dim3(8 * kNumCU), dim3(256), 0, s>>>(static_cast<const float4*>(d_in), static_cast<float4*>(d_out), n_c32 / 2);
    } else if (mode == 1) {
        probe_copy_kernel<float2><<<dim3(8 * kNumCU), dim3(256), 0, s>>>(static_cast<const float2*>(d_in), static_cast<float2*>(d_out), n_c32);
    } else if (mode == 2) {
        const size_t ntiles = n_c32 / 768, runs = static_cast<size_t>(waves_per_cu) * kNumCU;
        probe_tile_kernel<12><<<dim3((runs + 3) / 4), dim3(256), 0, s>>>(static_cast<const float2*>(d_in), static_cast<float2*>(d_out), ntiles, runs);
    } else if (mode == 9) {
        probe_fill_kernel<float2><<<dim3(8 * kNumCU), dim3(256), 0, s>>>(static_cast<float2*>(d_out), n_c32, make_float2(1.f, 2.f));
    } else if (mode == 10) {
        probe_fill_kernel<float4><<<dim3(8 * kNumCU), dim3(256), 0, s>>>(static_cast<float4*>(d_out), n_c32 / 2, make_float4(1.f, 2.f, 3.f, 4.f));
    } else if (mode >= 100) {
        const int K = mode - 100;
        const size_t n_waves = static_cast<size_t>(waves_per_cu) * kNumCU, ntiles = n_c32 / (64 * static_cast<size_t>(K));
        const dim3 g(static_cast<unsigned>(n_waves / 4)), b(256);
        const float2* in = static_cast<const float2*>(d_in);
        float2* out = static_cast<float2*>(d_out);
        switch (K) {
            case 1: probe_tile_rr_kernel<1><<<g, b, 0, s>>>(in, out, ntiles, n_waves); break;
            case 2: probe_tile_rr_kernel<2><<<g, b, 0, s>>>(in, out, ntiles, n_waves); break;
            case 4: probe_tile_rr_kernel<4><<<g, b, 0, s>>>(in, out, ntiles, n_waves); break;
            case 12: probe_tile_rr_kernel<12><<<g, b, 0, s>>>(in, out, ntiles, n_waves); break;
            case 16: probe_tile_rr_kernel<16><<<g, b, 0, s>>>(in, out, ntiles, n_waves); break;
            default: return COMMS_ERR_ARG;
        }
    } else if (mode >= 6 && mode <= 8) {
        const size_t ntiles = n_c32 / 768, runs = static_cast<size_t>(waves_per_cu) * kNumCU;
        const dim3 g((runs + 3) / 4), b(256);
        const float2* in = static_cast<const float2*>(d_in);
        float2* out = static_cast<float2*>(d_out);
        if (mode == 6) probe_tile_kernel<12, 1><<<g, b, 0, s>>>(in, out, ntiles, runs);
        else if (mode == 7) probe_tile_kernel<12, 2><<<g, b, 0, s>>>(in, out, ntiles, runs);
        else probe_tile_kernel<12, 3><<<g, b, 0, s>>>(in, out, ntiles, runs);
    } else if (mode == 4 || mode == 5) {
        const size_t n_tiles = (n_c32 >> 20) * 64;  // 64 tiles of 16 transforms per 2^20-point matrix
        const unsigned blocks = static_cast<unsigned>(waves_per_cu > 0 ? waves_per_cu : kNumCU);
        if (mode == 4)
            probe_fft_tile_kernel<0><<<dim3(blocks), dim3(1024), 0, s>>>(static_cast<const float2*>(d_in), static_cast<float2*>(d_out), n_tiles);
        else
            probe_fft_tile_kernel<1><<<dim3(blocks), dim3(1024), 0, s>>>(static_cast<const float2*>(d_in), static_cast<float2*>(d_out), n_tiles);
    } else {
        const size_t ntiles = n_c32 / 768, runs = static_cast<size_t>(waves_per_cu) * kNumCU;
        probe_tile4_kernel<<<dim3((runs + 3) / 4), dim3(256), 0, s>>>(static_cast<const float4*>(d_in), static_cast<float4*>(d_out), ntiles, runs);
    }
    return launch_ok("probe kernel");
}


// read-only probes: mode 0 / 1 grid-stride float4 / float2; 100 + FLAGS the decimating chain's tile pattern
// (probe_read_tile_kernel), wgs_per_cu persistent workgroups per CU
extern "C" comms_status_t comms_debug_read(const void* d_in, size_t n_c32, int mode, int wgs_per_cu, float* d_sink,
                                           float* d_out, void* stream) {
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (mode == 0)
        probe_read_kernel<float4><<<dim3(8 * kNumCU), dim3(256), 0, s>>>(static_cast<const float4*>(d_in), d_sink, n_c32 / 2);
    else if (mode == 1)
        probe_read_kernel<float2><<<dim3(8 * kNumCU), dim3(256), 0, s>>>(static_cast<const float2*>(d_in), d_sink, n_c32);
    else if (mode >= 100 && mode < 132) {
        const size_t n_tiles = n_c32 / 4096;
        const dim3 g(static_cast<unsigned>(wgs_per_cu) * kNumCU), b(256);
        const float2* in = static_cast<const float2*>(d_in);
        switch (mode - 100) {
#define COMMS_RT(F) case F: probe_read_tile_kernel<F><<<g, b, 0, s>>>(in, d_sink, d_out, n_tiles); break;
            COMMS_RT(0) COMMS_RT(1) COMMS_RT(2) COMMS_RT(3) COMMS_RT(4) COMMS_RT(5) COMMS_RT(6) COMMS_RT(7)
            COMMS_RT(8) COMMS_RT(9) COMMS_RT(12) COMMS_RT(13) COMMS_RT(16) COMMS_RT(17) COMMS_RT(18) COMMS_RT(19) COMMS_RT(20) COMMS_RT(21) COMMS_RT(22) COMMS_RT(23)
#undef COMMS_RT
            default: return COMMS_ERR_ARG;
        }
    } else if (mode >= 200 && mode < 208) {
        // wave-private streaming: wgs_per_cu = nt_chunk * 16 + workgroups per CU (4 waves each)
        const size_t n_tiles = n_c32 / 1024;
        const int wg = wgs_per_cu & 15, ntc = (wgs_per_cu >> 4) & 255, stm = wgs_per_cu >> 12;
        const dim3 g(static_cast<unsigned>(wg) * kNumCU), b(256);
        const float2* in = static_cast<const float2*>(d_in);
        float2* o2 = reinterpret_cast<float2*>(d_out);
        switch (mode - 200) {
#define COMMS_RW(F) case F: probe_read_wave_kernel<F><<<g, b, 0, s>>>(in, d_sink, o2, n_tiles, ntc, stm); break;
            COMMS_RW(0) COMMS_RW(1) COMMS_RW(2) COMMS_RW(3) COMMS_RW(4) COMMS_RW(5) COMMS_RW(6) COMMS_RW(7)
#undef COMMS_RW
        }
    } else
        return COMMS_ERR_ARG;
    return launch_ok("probe read kernel");
}

// ---- VALU issue-rate probe: ITER x 16 independent ops per lane
namespace comms {
template <int KIND>
__global__ __launch_bounds__(256) void probe_valu_kernel(float* out, int iters, float a, float b) {
    float r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == 0) r[i] = __builtin_fmaf(r[i], a, b);
            if (KIND == 1) r[i] = r[i] + a;
            if (KIND == 2) r[i] = r[i] * a;
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// f64 issue rates: KIND 40 v_fma_f64 (16 independent chains), 41 v_add_f64, 42 v_mul_f64, 43 v_fma_f64 with an SGPR factor
template <int KIND>
__global__ __launch_bounds__(256) void probe_f64_kernel(float* out, int iters, double a, double b) {
    double r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = threadIdx.x * 0.001 + i;
    double bv = b;
    asm volatile("" : "+v"(bv));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == 40) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(bv), "v"(bv));
            if (KIND == 41) asm volatile("v_add_f64 %0, %0, %1" : "+v"(r[i]) : "v"(bv));
            if (KIND == 42) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(r[i]) : "v"(bv));
            if (KIND == 43) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(r[i]) : "s"(a), "v"(bv));
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i];
    out[blockIdx.x * 256 + threadIdx.x] = static_cast<float>(s);
}

typedef float v2f __attribute__((ext_vector_type(2)));
// KIND 3: v_pk_fma_f32; KIND 4: complex multiply as 2 packed ops with op_sel/neg modifiers
template <int KIND>
__global__ __launch_bounds__(256) void probe_pk_kernel(float* out, int iters, float a, float b) {
    v2f r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = v2f{threadIdx.x * 0.001f + i, 1.0f + i};
    const v2f va{a, a * 0.5f}, vb{b, -b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == 3) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r[i]) : "v"(r[i]), "v"(va), "v"(vb));
            } else {
                v2f p;
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(p) : "v"(r[i]), "v"(va));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
                             : "=v"(r[i]) : "v"(r[i]), "v"(va), "v"(p));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i].x + r[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// The decimating chain's MAC: acc += u * tap with the real tap in one half of a 64-bit pair that is either an
// SGPR pair (kernel argument, as fir_decim.hip does) or a VGPR pair; NA accumulators = NA dependent chains.
struct MacProbeArgs {
    float t[32];
};
template <bool SGPR, int NA>
__global__ __launch_bounds__(256) void probe_mac_kernel(float* out, int iters, const MacProbeArgs a) {
    v2f acc[NA], u[8];
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = v2f{threadIdx.x * 0.001f + i, 1.0f + i};
#pragma unroll
    for (int i = 0; i < 8; ++i) u[i] = v2f{threadIdx.x * 0.002f + i, 2.0f - i};
    v2f tv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) tv[i] = v2f{a.t[2 * i], a.t[2 * i + 1]};
    if (!SGPR) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(tv[i]));  // pin the pairs in VGPRs
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            v2f& ac = acc[i % NA];
            if (SGPR) {
                if (i & 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(ac) : "v"(u[i & 7]), "s"(tv[i >> 1]));
                else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(ac) : "v"(u[i & 7]), "s"(tv[i >> 1]));
            } else {
                if (i & 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(ac) : "v"(u[i & 7]), "v"(tv[i >> 1]));
                else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(ac) : "v"(u[i & 7]), "v"(tv[i >> 1]));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NA; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// Operand-form matrix for the packed MAC (16 chains): which part of `acc += u * tap` costs the second pass?
template <int V>
__global__ __launch_bounds__(256) void probe_mac_form_kernel(float* out, int iters, const MacProbeArgs a) {
    v2f acc[16], u[8], tv[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = v2f{threadIdx.x * 0.001f + i, 1.0f + i};
#pragma unroll
    for (int i = 0; i < 8; ++i) u[i] = v2f{threadIdx.x * 0.002f + i, 2.0f - i};
#pragma unroll
    for (int i = 0; i < 8; ++i) tv[i] = v2f{a.t[2 * i], a.t[2 * i + 1]};
    if (V != 5 && V != 6 && V != 13) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(tv[i]));
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            v2f& ac = acc[i];
            const v2f& uu = u[i & 7];
            const v2f& tt = tv[i >> 1];
            if (V == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(ac) : "v"(uu), "v"(tt));
            if (V == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(ac) : "v"(uu), "v"(tt));
            if (V == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(ac) : "v"(uu), "v"(tt));
            if (V == 3) asm volatile("v_pk_fma_f32 %0, %0, %2, %1" : "+v"(ac) : "v"(uu), "v"(tt));
            if (V == 4) asm volatile("v_pk_fma_f32 %0, %0, %2, %1 op_sel_hi:[1,0,1]" : "+v"(ac) : "v"(uu), "v"(tt));
            if (V == 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(ac) : "v"(uu), "s"(tt));
            if (V == 6) {
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(ac.x) : "v"(uu.x), "s"(tt.x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(ac.y) : "v"(uu.y), "s"(tt.x));
            }
            if (V == 7) {
                v2f p;
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(p) : "v"(uu), "v"(tt));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(ac) : "v"(p));
            }
            if (V == 8) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(ac) : "v"(uu));        // two distinct VGPR pairs only
            if (V == 9) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(ac) : "v"(uu), "v"(tt));  // no accumulate
            if (V == 10 || V == 11 || V == 13) {  // lo / hi alternating: 10 same pair twice, 11 a new pair each time, 13 SGPR pair
                const v2f& t2 = V == 11 ? tv[i & 7] : tt;
                if (V == 13) {
                    if (i & 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(ac) : "v"(uu), "s"(t2));
                    else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(ac) : "v"(uu), "s"(t2));
                } else {
                    if (i & 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(ac) : "v"(uu), "v"(t2));
                    else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(ac) : "v"(uu), "v"(t2));
                }
            }
            if (V == 12) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(ac) : "v"(uu), "v"(tt));  // lo, lo on the same pair
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// KIND 5: v_permlane32_swap_b32, KIND 6: v_permlane16_swap_b32 (8 swaps of register pairs per 16 "ops")
template <int KIND>
__global__ __launch_bounds__(256) void probe_swap_kernel(float* out, int iters) {
    unsigned r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = threadIdx.x * 16 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            // the builtins, not inline asm: the compiler then inserts the wait states these need
            // after a VALU write of their operands (raw asm right after a v_mov read stale lanes)
            const auto q = KIND == 5 ? __builtin_amdgcn_permlane32_swap(r[i], r[i + 1], false, false)
                                     : __builtin_amdgcn_permlane16_swap(r[i], r[i + 1], false, false);
            r[i] = q[0];
            r[i + 1] = q[1];
        }
    }
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i];
    out[blockIdx.x * 256 + threadIdx.x] = static_cast<float>(s & 0xffff);
}
// what the swaps do: out[0..63] / [64..127] = (a, b) after v_permlane32_swap of a = lane, b = 100 + lane;
// out[128..255] the same for v_permlane16_swap
__global__ void probe_swap_semantics_kernel(unsigned* out) {
    const unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    const auto p = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[threadIdx.x] = p[0];
    out[64 + threadIdx.x] = p[1];
    const auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[128 + threadIdx.x] = q[0];
    out[192 + threadIdx.x] = q[1];
}
}  // namespace comms
extern "C" comms_status_t comms_debug_swap_semantics(unsigned* d_out, void* stream) {
    comms::probe_swap_semantics_kernel<<<dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream)>>>(d_out);
    return comms::launch_ok("probe_swap_semantics");
}
extern "C" comms_status_t comms_debug_valu(float* d_out, int kind, int iters, int blocks, void* stream) {
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (kind == 0) comms::probe_valu_kernel<0><<<dim3(blocks), dim3(256), 0, s>>>(d_out, iters, 1.0001f, 0.5f);
    else if (kind == 1) comms::probe_valu_kernel<1><<<dim3(blocks), dim3(256), 0, s>>>(d_out, iters, 1.0001f, 0.5f);
    else if (kind == 2) comms::probe_valu_kernel<2><<<dim3(blocks), dim3(256), 0, s>>>(d_out, iters, 1.0001f, 0.5f);
    else if (kind == 5) comms::probe_swap_kernel<5><<<dim3(blocks), dim3(256), 0, s>>>(d_out, iters);
    else if (kind == 6) comms::probe_swap_kernel<6><<<dim3(blocks), dim3(256), 0, s>>>(d_out, iters);
    else if (kind == 40) comms::probe_f64_kernel<40><<<dim3(blocks), dim3(256), 0, s>>>(d_out, iters, 0.9999, 1e-3);
    else if (kind == 41) comms::probe_f64_kernel<41><<<dim3(blocks), dim3(256), 0, s>>>(d_out, iters, 0.9999, 1e-3);
    else if (kind == 42) comms::probe_f64_kernel<42><<<dim3(blocks), dim3(256), 0, s>>>(d_out, iters, 0.9999, 1.0000001);
    else if (kind == 43) comms::probe_f64_kernel<43><<<dim3(blocks), dim3(256), 0, s>>>(d_out, iters, 0.9999, 1e-3);
    else if (kind >= 20 && kind <= 33) {
        comms::MacProbeArgs a;
        for (int i = 0; i < 32; ++i) a.t[i] = 1e-3f * (i + 1);
#define COMMS_PF(V) if (kind == 20 + V) comms::probe_mac_form_kernel<V><<<dim3(blocks), dim3(256), 0, s>>>(d_out, iters, a);
        COMMS_PF(0) COMMS_PF(1) COMMS_PF(2) COMMS_PF(3) COMMS_PF(4) COMMS_PF(5) COMMS_PF(6) COMMS_PF(7) COMMS_PF(8) COMMS_PF(9) COMMS_PF(10) COMMS_PF(11) COMMS_PF(12) COMMS_PF(13)
#undef COMMS_PF
    } else if (kind >= 7 && kind <= 12) {
        comms::MacProbeArgs a;
        for (int i = 0; i < 32; ++i) a.t[i] = 1e-3f * (i + 1);
        if (kind == 7) comms::probe_mac_kernel<true, 2><<<dim3(blocks), dim3(256), 0, s>>>(d_out, iters, a);
        if (kind == 8) comms::probe_mac_kernel<false, 2><<<dim3(blocks), dim3(256), 0, s>>>(d_out, iters, a);
        if (kind == 9) comms::probe_mac_kernel<true, 16><<<dim3(blocks), dim3(256), 0, s>>>(d_out, iters, a);
        if (kind == 10) comms::probe_mac_kernel<false, 16><<<dim3(blocks), dim3(256), 0, s>>>(d_out, iters, a);
        if (kind == 11) comms::probe_mac_kernel<true, 4><<<dim3(blocks), dim3(256), 0, s>>>(d_out, iters, a);
        if (kind == 12) comms::probe_mac_kernel<false, 4><<<dim3(blocks), dim3(256), 0, s>>>(d_out, iters, a);
    } else if (kind == 3) comms::probe_pk_kernel<3><<<dim3(blocks), dim3(256), 0, s>>>(d_out, iters, 0.9999f, 0.01f);
    else comms::probe_pk_kernel<4><<<dim3(blocks), dim3(256), 0, s>>>(d_out, iters, 0.9999f, 0.01f);
    return comms::launch_ok("probe_valu");
}
#endif  // COMMS_DIAG
