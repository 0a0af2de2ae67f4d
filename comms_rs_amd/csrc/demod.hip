// demod.hip -- TimingEstimator and NCO.
//
// TimingEstimator::push  src/demodulation/timing_estimator.rs:85-112 (Mengali 8.4, f64):
//     r[i]   = exp(-i*pi*i/n)                       (:90)
//     qin[i] = conj(s[i])*r[i],  din[i] = s[i]*r[i] (:93-94)
//     qout   = batch_fir(qin, q(t) taps, zeros)     (:102, fresh state every push)
//     dout   = batch_fir(din, delay of n*d, zeros)  (:103)
//     est    = -n * arg( sum qout*dout ) / (2*pi)   (:108-111)
//   One kernel: a workgroup stages the mixed window of its 256 outputs in LDS, every lane
//   runs the q filter for its output in the reference's tap order (k = 0.. ascending, so
//   qout is the same f64 value), multiplies by the delayed sample and the products are
//   tree-reduced (the reference folds them sequentially: rounding-level difference only).
//
// Nco::push  src/demodulation/nco.rs:71-77: phase += dphase + perr; single wrap above 2*pi;
//   out = exp(i*phase).  The recurrence is a prefix sum, so a block of phase errors is a
//   scan: increments are converted to 64-bit fixed-point turns (wrap-around of the integer
//   is the mod-2*pi), scanned exactly, and one f64 sincos per sample produces the output.
//   Three launches: tile sums, scan of tile sums (one workgroup), apply.  Differences to
//   the reference's sequentially rounded f64 phase stay below ~n * 2^-63 turns.
#include <cmath>
#include <vector>

#include "common.hpp"

namespace comms {

constexpr double kPi = 3.14159265358979323846264338327950288;

// ---------------------------------------------------------------- timing estimator
constexpr int TE_T = 256;   // outputs per tile (= workgroup size)
constexpr int TE_KC = 256;  // taps staged per pass

__device__ __forceinline__ double2 te_rotor(long long i, double n_sps) {
    // Complex::new(0.0, -PI * i as f64 / n as f64).exp()  -> (cos, sin) of that angle
    const double th = (-kPi * static_cast<double>(i)) / n_sps;
    double s, c;
    sincos(th, &s, &c);
    return make_double2(c, s);
}

__global__ __launch_bounds__(TE_T) void timing_kernel(const double2* __restrict__ x, size_t len,
                                                      const double* __restrict__ qtaps, uint32_t n_q, uint32_t nd,
                                                      double n_sps, double2* __restrict__ partials) {
    __shared__ double2 sh_q[TE_T + TE_KC - 1];
    __shared__ double sh_t[TE_KC];
    __shared__ double2 wsum[TE_T / 64];
    const int tid = threadIdx.x;
    const size_t ntiles = (len + TE_T - 1) / TE_T;
    double2 total = make_double2(0.0, 0.0);
    for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long i0 = static_cast<long long>(tile) * TE_T;
        const long long i = i0 + tid;
        double2 q = make_double2(0.0, 0.0);
        for (uint32_t k0 = 0; k0 < n_q; k0 += TE_KC) {
            const int kc = static_cast<int>(n_q - k0 < static_cast<uint32_t>(TE_KC) ? n_q - k0 : TE_KC);
            const long long base = i0 - static_cast<long long>(k0) - (kc - 1);
            __syncthreads();
            for (int j = tid; j < TE_T - 1 + kc; j += TE_T) {
                const long long idx = base + j;
                double2 v = make_double2(0.0, 0.0);
                if (idx >= 0 && idx < static_cast<long long>(len)) {
                    const double2 s = x[idx];
                    const double2 r = te_rotor(idx, n_sps);
                    const double ci = -s.y;  // conj
                    v = make_double2(s.x * r.x - ci * r.y, s.x * r.y + ci * r.x);
                }
                sh_q[j] = v;
            }
            if (tid < kc) sh_t[tid] = qtaps[k0 + tid];
            __syncthreads();
            for (int kk = 0; kk < kc; ++kk) {
                const double t = sh_t[kk];
                const double2 v = sh_q[tid + kc - 1 - kk];
                q.x += t * v.x;
                q.y += t * v.y;
            }
        }
        if (i < static_cast<long long>(len) && i >= static_cast<long long>(nd)) {
            const double2 s = x[i - nd];
            const double2 r = te_rotor(i - nd, n_sps);
            const double2 d = make_double2(s.x * r.x - s.y * r.y, s.x * r.y + s.y * r.x);
            total.x += q.x * d.x - q.y * d.y;
            total.y += q.x * d.y + q.y * d.x;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        total.x += __shfl_down(total.x, off);
        total.y += __shfl_down(total.y, off);
    }
    if ((tid & 63) == 0) wsum[tid >> 6] = total;
    __syncthreads();
    if (tid == 0) {
        double2 s = wsum[0];
        for (int w = 1; w < TE_T / 64; ++w) {
            s.x += wsum[w].x;
            s.y += wsum[w].y;
        }
        partials[blockIdx.x] = s;
    }
}

// ---------------------------------------------------------------- NCO
constexpr int NCO_WG = 256;
constexpr int NCO_PER = 8;                    // samples per lane
constexpr int NCO_TILE = NCO_WG * NCO_PER;    // 2048 samples per workgroup
constexpr double kTwoPi = 2.0 * kPi;          // the reference's 2.0 * PI

// (dphase + perr) -> fixed-point turns (two's complement, modulo one turn).  1/(2*pi) is
// carried as hi + lo so that the conversion error stays near 2^-64 turns for |x| <= 2*pi.
__device__ __forceinline__ uint64_t nco_turns(double x) {
    constexpr double inv_hi = 0x1.45f306dc9c883p-3;   // fl(1/(2*pi))
    constexpr double inv_lo = -0x1.6b01ec5417056p-57;  // 1/(2*pi) - inv_hi
    const double hi = x * inv_hi;
    const double lo = __fma_rn(x, inv_hi, -hi) + x * inv_lo;
    // hi = whole + frac; whole turns vanish modulo 2^64
    const double fr = hi - rint(hi);  // exact, in [-0.5, 0.5]
    const uint64_t a = static_cast<uint64_t>(static_cast<long long>(fr * 0x1.0p63)) << 1;  // fr * 2^64 mod 2^64
    const uint64_t b = static_cast<uint64_t>(static_cast<long long>(rint(lo * 0x1.0p64)));
    return a + b;
}

__device__ __forceinline__ uint64_t wave_incl_scan(uint64_t v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint64_t o = __shfl_up(v, off);
        if (lane >= off) v += o;
    }
    return v;
}

// tile sums
__global__ __launch_bounds__(NCO_WG) void nco_sum_kernel(const double* __restrict__ perr, size_t n, double dphase,
                                                         uint64_t* __restrict__ tile_sum) {
    __shared__ uint64_t wsum[NCO_WG / 64];
    const size_t base = static_cast<size_t>(blockIdx.x) * NCO_TILE + static_cast<size_t>(threadIdx.x) * NCO_PER;
    uint64_t acc = 0;
#pragma unroll
    for (int j = 0; j < NCO_PER; ++j)
        if (base + j < n) acc += nco_turns(dphase + perr[base + j]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t s = 0;
        for (int w = 0; w < NCO_WG / 64; ++w) s += wsum[w];
        tile_sum[blockIdx.x] = s;
    }
}

// exclusive scan of the tile sums in place (one workgroup), seeded with the node's phase;
// the total (phase after the block) goes to *phase_io.
__global__ __launch_bounds__(1024) void nco_scan_kernel(uint64_t* __restrict__ tile_sum, size_t ntiles,
                                                        uint64_t* __restrict__ phase_io) {
    __shared__ uint64_t wtot[16];
    __shared__ uint64_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) carry_s = *phase_io;
    __syncthreads();
    for (size_t b0 = 0; b0 < ntiles; b0 += 1024) {
        const size_t i = b0 + tid;
        const uint64_t v = i < ntiles ? tile_sum[i] : 0;
        const uint64_t inc = wave_incl_scan(v, lane);
        if (lane == 63) wtot[w] = inc;
        __syncthreads();
        uint64_t off = carry_s;
        for (int k = 0; k < w; ++k) off += wtot[k];
        if (i < ntiles) tile_sum[i] = off + inc - v;
        __syncthreads();
        if (tid == 1023) carry_s = off + inc;
        __syncthreads();
    }
    if (tid == 0) *phase_io = carry_s;
}

// out[i] = exp(i * phase_i),  phase_i = phase_before + sum_{j<=i} (dphase + perr[j])
__global__ __launch_bounds__(NCO_WG) void nco_apply_kernel(const double* __restrict__ perr, size_t n, double dphase,
                                                           const uint64_t* __restrict__ tile_off,
                                                           double2* __restrict__ out) {
    __shared__ uint64_t wtot[NCO_WG / 64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const size_t base = static_cast<size_t>(blockIdx.x) * NCO_TILE + static_cast<size_t>(tid) * NCO_PER;
    uint64_t t[NCO_PER];
    uint64_t acc = 0;
#pragma unroll
    for (int j = 0; j < NCO_PER; ++j) {
        acc += base + j < n ? nco_turns(dphase + perr[base + j]) : 0;
        t[j] = acc;
    }
    const uint64_t inc = wave_incl_scan(acc, lane);
    if (lane == 63) wtot[w] = inc;
    __syncthreads();
    uint64_t off = tile_off[blockIdx.x] + inc - acc;
    for (int k = 0; k < w; ++k) off += wtot[k];
#pragma unroll
    for (int j = 0; j < NCO_PER; ++j) {
        if (base + j < n) {
            const uint64_t ph = off + t[j];
            // 53 significant bits of the turn fraction -> radians in [0, 2*pi)
            const double ang = static_cast<double>(ph >> 11) * (kTwoPi * 0x1.0p-53);
            double s, c;
            sincos(ang, &s, &c);
            out[base + j] = make_double2(c, s);
        }
    }
}

}  // namespace comms

using namespace comms;

struct comms_timing : Handle {
    uint32_t n = 0, d = 0, n_q = 0;
    double* d_taps = nullptr;
    double2* d_part = nullptr;
    unsigned max_blocks = 8 * kNumCU;
};

struct comms_nco : Handle {
    double dphase = 0.0;
    uint64_t* d_phase = nullptr;  // fixed-point turns, device resident
    Scratch tiles;
};

static comms_status_t qfilt_host(uint32_t n_taps, double alpha, uint32_t sam_per_sym, std::vector<double>& out) {
    COMMS_ARG(alpha >= 0.0 && alpha <= 1.0, "InvalidRolloffError: alpha=%g outside [0,1]", alpha);
    COMMS_ARG(sam_per_sym >= 1, "sam_per_sym must be >= 1");
    COMMS_ARG(n_taps < (1u << 30), "n_taps too large");
    const uint32_t real_n = n_taps % 2 == 0 ? n_taps + 1 : n_taps;  // util/math.rs:317-320
    const int32_t half = static_cast<int32_t>(std::floor(static_cast<double>(real_n) / 2.0));
    out.resize(real_n);
    for (uint32_t i = 0; i < real_n; ++i) {
        const double tt = static_cast<double>(static_cast<int32_t>(i) - half) / static_cast<double>(sam_per_sym);
        const double two_alpha_tt = 2.0 * alpha * tt;
        if (std::fabs(two_alpha_tt) == 1.0) {  // l'Hospital branch, util/math.rs:331-333
            out[i] = std::sin(kPi * alpha * tt) / (8.0 * tt);
        } else {
            const double num = alpha * std::cos(kPi * alpha * tt);
            const double den = kPi * (1.0 - (two_alpha_tt * two_alpha_tt));
            out[i] = num / den;
        }
    }
    return COMMS_OK;
}

extern "C" {

size_t comms_qfilt_len(uint32_t n_taps) { return n_taps % 2 == 0 ? static_cast<size_t>(n_taps) + 1 : n_taps; }

comms_status_t comms_qfilt_taps(uint32_t n_taps, double alpha, uint32_t sam_per_sym, double* out) {
    COMMS_ARG(out != nullptr, "out is NULL");
    std::vector<double> t;
    COMMS_TRY(qfilt_host(n_taps, alpha, sam_per_sym, t));
    std::memcpy(out, t.data(), t.size() * sizeof(double));
    return COMMS_OK;
}

comms_status_t comms_timing_create(uint32_t n, uint32_t d, double alpha, int32_t device, comms_timing_t** out) {
    COMMS_ARG(out != nullptr, "out is NULL");
    *out = nullptr;
    COMMS_ARG(n >= 1, "samples per symbol must be >= 1");
    COMMS_ARG(static_cast<uint64_t>(n) * d < (1u << 24), "filter delay n*d too large");
    std::vector<double> taps;
    COMMS_TRY(qfilt_host(2 * n * d + 1, alpha, n, taps));
    comms_timing* h = new (std::nothrow) comms_timing;
    COMMS_ARG(h != nullptr, "out of host memory");
    comms_status_t st = h->init(device);
    if (st != COMMS_OK) {
        delete h;
        return st;
    }
    h->n = n;
    h->d = d;
    h->n_q = static_cast<uint32_t>(taps.size());
    hipError_t e = hipMalloc(&h->d_taps, taps.size() * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(h->d_taps, taps.data(), taps.size() * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&h->d_part, h->max_blocks * sizeof(double2));
    if (e != hipSuccess) {
        if (h->d_taps) (void)hipFree(h->d_taps);
        h->fini();
        delete h;
        return fail(COMMS_ERR_DEVICE, "timing estimator alloc: %s", hipGetErrorString(e));
    }
    *out = h;
    return COMMS_OK;
}

comms_status_t comms_timing_push_dev(comms_timing_t* h, const double* d_samples, size_t len, double* estimate,
                                     void* stream) {
    COMMS_ARG(h != nullptr && estimate != nullptr, "NULL argument");
    COMMS_ARG(d_samples || !len, "NULL device pointer");
    COMMS_ARG((reinterpret_cast<uintptr_t>(d_samples) & 15) == 0, "samples must be 16-byte aligned");
    COMMS_TRY(use_device(h->device));
    hipStream_t s = nullptr;
    COMMS_TRY(h->enter(stream, &s));
    double re = 0.0, im = 0.0;
    if (len) {
        size_t blocks = (len + TE_T - 1) / TE_T;
        if (blocks > h->max_blocks) blocks = h->max_blocks;
        h->tic(s);
        timing_kernel<<<dim3(static_cast<unsigned>(blocks)), dim3(TE_T), 0, s>>>(
            reinterpret_cast<const double2*>(d_samples), len, h->d_taps, h->n_q, h->n * h->d,
            static_cast<double>(h->n), h->d_part);
        h->toc(s);
        COMMS_TRY(launch_ok("timing_kernel"));
        std::vector<double2> part(blocks);
        COMMS_HIP_TRY(hipMemcpyAsync(part.data(), h->d_part, blocks * sizeof(double2), hipMemcpyDeviceToHost, s));
        COMMS_HIP_TRY(hipStreamSynchronize(s));
        for (size_t b = 0; b < blocks; ++b) {
            re += part[b].x;
            im += part[b].y;
        }
    }
    // -(self.n as f64) * sum_value.arg() / (2.0 * PI)
    *estimate = (-static_cast<double>(h->n) * std::atan2(im, re)) / (2.0 * kPi);
    return COMMS_OK;
}

comms_status_t comms_timing_push(comms_timing_t* h, const double* samples, size_t len, double* estimate) {
    COMMS_ARG(h != nullptr && estimate != nullptr, "NULL argument");
    COMMS_ARG(samples || !len, "NULL host pointer");
    COMMS_TRY(use_device(h->device));
    if (len) {
        COMMS_TRY(h->in_scratch.reserve(len * 16));
        COMMS_HIP_TRY(hipMemcpyAsync(h->in_scratch.p, samples, len * 16, hipMemcpyHostToDevice, h->stream));
    }
    return comms_timing_push_dev(h, static_cast<const double*>(h->in_scratch.p), len, estimate, COMMS_STREAM_HANDLE);
}

comms_status_t comms_timing_destroy(comms_timing_t* h) {
    if (!h) return COMMS_OK;
    (void)use_device(h->device);
    if (h->d_taps) (void)hipFree(h->d_taps);
    if (h->d_part) (void)hipFree(h->d_part);
    h->fini();
    delete h;
    return COMMS_OK;
}

// ---- NCO -------------------------------------------------------------------------
comms_status_t comms_nco_create(double dphase, double phase, int32_t device, comms_nco_t** out) {
    COMMS_ARG(out != nullptr, "out is NULL");
    *out = nullptr;
    COMMS_ARG(std::isfinite(dphase) && std::isfinite(phase), "dphase and phase must be finite");
    comms_nco* h = new (std::nothrow) comms_nco;
    COMMS_ARG(h != nullptr, "out of host memory");
    comms_status_t st = h->init(device);
    if (st != COMMS_OK) {
        delete h;
        return st;
    }
    h->dphase = mix_wrap_dphase(dphase);  // Nco::new, nco.rs:41-49 (same wrap as Mixer::new)
    const uint64_t turns = mix_to_turns(phase);
    hipError_t e = hipMalloc(&h->d_phase, sizeof(uint64_t));
    if (e == hipSuccess) e = hipMemcpy(h->d_phase, &turns, sizeof(uint64_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        h->fini();
        delete h;
        return fail(COMMS_ERR_DEVICE, "nco state alloc: %s", hipGetErrorString(e));
    }
    *out = h;
    return COMMS_OK;
}

comms_status_t comms_nco_run_dev(comms_nco_t* h, const double* d_perr, size_t n, double* d_out, void* stream) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG((d_perr && d_out) || !n, "NULL device pointer");
    COMMS_ARG((reinterpret_cast<uintptr_t>(d_out) & 15) == 0, "output must be 16-byte aligned");
    COMMS_TRY(use_device(h->device));
    if (!n) return COMMS_OK;
    COMMS_ARG(!ranges_overlap(d_perr, n * 8, d_out, n * 16), "nco cannot run in place");
    hipStream_t s = nullptr;
    COMMS_TRY(h->enter(stream, &s));
    const size_t ntiles = (n + NCO_TILE - 1) / NCO_TILE;
    COMMS_ARG(ntiles <= 0x7fffffffu, "block too long");
    COMMS_TRY(h->tiles.reserve(ntiles * sizeof(uint64_t)));
    uint64_t* tiles = static_cast<uint64_t*>(h->tiles.p);
    h->tic(s);
    nco_sum_kernel<<<dim3(static_cast<unsigned>(ntiles)), dim3(NCO_WG), 0, s>>>(d_perr, n, h->dphase, tiles);
    nco_scan_kernel<<<dim3(1), dim3(1024), 0, s>>>(tiles, ntiles, h->d_phase);
    nco_apply_kernel<<<dim3(static_cast<unsigned>(ntiles)), dim3(NCO_WG), 0, s>>>(d_perr, n, h->dphase, tiles,
                                                                                 reinterpret_cast<double2*>(d_out));
    h->toc(s);
    return launch_ok("nco kernels");
}

comms_status_t comms_nco_run(comms_nco_t* h, const double* perr, size_t n, double* out) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG((perr && out) || !n, "NULL host pointer");
    COMMS_TRY(use_device(h->device));
    if (!n) return COMMS_OK;
    COMMS_TRY(h->in_scratch.reserve(n * 8));
    COMMS_TRY(h->out_scratch.reserve(n * 16));
    COMMS_HIP_TRY(hipMemcpyAsync(h->in_scratch.p, perr, n * 8, hipMemcpyHostToDevice, h->stream));
    COMMS_TRY(comms_nco_run_dev(h, static_cast<const double*>(h->in_scratch.p), n,
                                static_cast<double*>(h->out_scratch.p), COMMS_STREAM_HANDLE));
    COMMS_HIP_TRY(hipMemcpyAsync(out, h->out_scratch.p, n * 16, hipMemcpyDeviceToHost, h->stream));
    COMMS_HIP_TRY(hipStreamSynchronize(h->stream));
    return COMMS_OK;
}

comms_status_t comms_nco_get_phase(comms_nco_t* h, double* phase) {
    COMMS_ARG(h != nullptr && phase != nullptr, "NULL argument");
    COMMS_TRY(use_device(h->device));
    uint64_t turns = 0;
    COMMS_HIP_TRY(hipDeviceSynchronize());
    COMMS_HIP_TRY(hipMemcpy(&turns, h->d_phase, sizeof(uint64_t), hipMemcpyDeviceToHost));
    *phase = static_cast<double>(turns >> 11) * (kMixT * 0x1.0p-53);
    return COMMS_OK;
}

comms_status_t comms_nco_destroy(comms_nco_t* h) {
    if (!h) return COMMS_OK;
    (void)use_device(h->device);
    if (h->d_phase) (void)hipFree(h->d_phase);
    h->tiles.release();
    h->fini();
    delete h;
    return COMMS_OK;
}

}  // extern "C"
