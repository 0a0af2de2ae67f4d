// estimators.hip -- block estimators as device reductions.
//
// Replaces (f64 in, one f64 scalar out, as in the reference):
//   frequency_offset_estimate  src/demodulation/frequency_estimator.rs:27-42
//       arg( sum_n x[n+1] * conj(x[n]) )
//   psk_phase_estimate(m)      src/demodulation/phase_estimator.rs:26-33    arg( sum x^m ) / m
//   qam_phase_estimate         src/demodulation/phase_estimator.rs:58-65    arg( sum -x^4 ) / 4
// Each is map -> complex f64 sum -> arg: HBM-bound (16 B per Complex<f64> sample).  Lanes
// accumulate a grid-stride slice, waves reduce by shuffles, one partial per workgroup goes
// to HBM; the (fixed-order) sum of the <= 2048 partials, atan2 and the division run on the
// host, so results are reproducible run to run.  The summation order differs from the
// reference's sequential fold (rounding-level differences only, ~1e-16 relative).
#include <cmath>
#include <vector>

#include "common.hpp"

namespace comms {

__device__ __forceinline__ double2 zmul(double2 a, double2 b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// Complex::powi(exp >= 0): exponentiation by squaring in num_traits::pow's operation order
__device__ __forceinline__ double2 zpowi(double2 base, unsigned exp) {
    if (exp == 0) return make_double2(1.0, 0.0);
    while ((exp & 1) == 0) {
        base = zmul(base, base);
        exp >>= 1;
    }
    if (exp == 1) return base;
    double2 acc = base;
    while (exp > 1) {
        exp >>= 1;
        base = zmul(base, base);
        if (exp & 1) acc = zmul(acc, base);
    }
    return acc;
}

// KIND 0: x[i+1]*conj(x[i]) (i < n-1);  1: x^m;  2: -x^4
template <int KIND>
__global__ __launch_bounds__(256) void estimator_kernel(const double2* __restrict__ x, size_t n, unsigned m,
                                                        double2* __restrict__ partials) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    const size_t count = KIND == 0 ? (n ? n - 1 : 0) : n;
    auto term = [&](size_t i) {
        if (KIND == 0) {
            const double2 a = x[i + 1], b = x[i];
            return zmul(a, make_double2(b.x, -b.y));
        } else if (KIND == 1) {
            return zpowi(x[i], m);
        } else {
            const double2 p = zpowi(x[i], 4);
            return make_double2(-1.0 * p.x, -1.0 * p.y);
        }
    };
    // four sweeps of the grid per iteration, four accumulators: the loads of a lane are in flight together
    // (one load per iteration left the kernel waiting on memory latency: 70 us for 2^24 samples)
    double2 a4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a4[u] = make_double2(0.0, 0.0);
    size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < count; i += 4 * stride) {
        double2 p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u] = term(i + u * stride);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a4[u].x += p[u].x;
            a4[u].y += p[u].y;
        }
    }
    for (; i < count; i += stride) {
        const double2 p = term(i);
        a4[0].x += p.x;
        a4[0].y += p.y;
    }
    double2 acc = make_double2((a4[0].x + a4[1].x) + (a4[2].x + a4[3].x), (a4[0].y + a4[1].y) + (a4[2].y + a4[3].y));
    // wave reduction by shuffles, then one LDS hop across the 4 waves
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        acc.x += __shfl_down(acc.x, off);
        acc.y += __shfl_down(acc.y, off);
    }
    __shared__ double2 wsum[4];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double2 s = wsum[0];
        for (int w = 1; w < 4; ++w) {
            s.x += wsum[w].x;
            s.y += wsum[w].y;
        }
        partials[blockIdx.x] = s;
    }
}

static comms_status_t estimate(int kind, const double* d_x, size_t n, unsigned m, double* out, int32_t device,
                               void* stream) {
    COMMS_ARG(out != nullptr && (d_x || !n), "NULL argument");
    COMMS_ARG((reinterpret_cast<uintptr_t>(d_x) & 15) == 0, "samples must be 16-byte aligned");
    COMMS_TRY(use_device(device));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    size_t blocks = (n + 255) / 256;
    if (blocks > 8u * kNumCU) blocks = 8u * kNumCU;
    if (blocks < 1) blocks = 1;
    // per-thread, per-device partials buffer, allocated once (a node lives on one thread and every
    // call ends with a stream sync, so the buffer is never shared between launches in flight;
    // hipMalloc / hipFree per call would also synchronise the whole device)
    static thread_local double2* tl_part[64] = {};
    COMMS_ARG(device >= 0 && device < 64, "device index out of range");
    if (!tl_part[device]) COMMS_HIP_TRY(hipMalloc(&tl_part[device], 8u * kNumCU * sizeof(double2)));
    double2* d_part = tl_part[device];
    const double2* x = reinterpret_cast<const double2*>(d_x);
    if (kind == 0)
        estimator_kernel<0><<<dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s>>>(x, n, m, d_part);
    else if (kind == 1)
        estimator_kernel<1><<<dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s>>>(x, n, m, d_part);
    else
        estimator_kernel<2><<<dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s>>>(x, n, m, d_part);
    comms_status_t st = launch_ok("estimator_kernel");
    std::vector<double2> part(blocks);
    hipError_t e = hipSuccess;
    if (st == COMMS_OK) e = hipMemcpyAsync(part.data(), d_part, blocks * sizeof(double2), hipMemcpyDeviceToHost, s);
    if (st == COMMS_OK && e == hipSuccess) e = hipStreamSynchronize(s);
    if (st != COMMS_OK) return st;
    if (e != hipSuccess) return fail(COMMS_ERR_DEVICE, "estimator copy-back: %s", hipGetErrorString(e));
    double re = 0.0, im = 0.0;
    for (size_t b = 0; b < blocks; ++b) {
        re += part[b].x;
        im += part[b].y;
    }
    const double ang = std::atan2(im, re);  // Complex::arg
    *out = kind == 0 ? ang : kind == 1 ? ang / static_cast<double>(m) : ang / 4.0;
    return COMMS_OK;
}

static comms_status_t estimate_host(int kind, const double* x, size_t n, unsigned m, double* out, int32_t device) {
    COMMS_ARG(out != nullptr && (x || !n), "NULL argument");
    COMMS_TRY(use_device(device));
    Handle* h = nullptr;
    COMMS_TRY(thread_handle(device, &h));
    if (!n) return estimate(kind, nullptr, 0, m, out, device, h->stream);
    // input only: short blocks are read straight from pinned host memory, long ones uploaded
    const void* d = nullptr;
    if (n * 16 <= zero_copy_limit()) {
        COMMS_TRY(h->pin_in.reserve(n * 16));
        std::memcpy(h->pin_in.h, x, n * 16);
        d = h->pin_in.d;
    } else {
        COMMS_TRY(h->in_scratch.reserve(n * 16));
        COMMS_HIP_TRY(hipMemcpyAsync(h->in_scratch.p, x, n * 16, hipMemcpyHostToDevice, h->stream));
        d = h->in_scratch.p;
    }
    return estimate(kind, static_cast<const double*>(d), n, m, out, device, h->stream);
}

}  // namespace comms

using namespace comms;

extern "C" {

comms_status_t comms_frequency_offset_estimate_dev(const double* d_samples, size_t n, double* out, int32_t device,
                                                   void* stream) {
    return estimate(0, d_samples, n, 0, out, device, stream);
}
comms_status_t comms_psk_phase_estimate_dev(const double* d_symbols, size_t n, uint32_t m, double* out,
                                            int32_t device, void* stream) {
    COMMS_ARG(m >= 1 && m <= (1u << 20), "PSK order m out of range");
    return estimate(1, d_symbols, n, m, out, device, stream);
}
comms_status_t comms_qam_phase_estimate_dev(const double* d_symbols, size_t n, double* out, int32_t device,
                                            void* stream) {
    return estimate(2, d_symbols, n, 4, out, device, stream);
}
comms_status_t comms_frequency_offset_estimate(const double* samples, size_t n, double* out, int32_t device) {
    return estimate_host(0, samples, n, 0, out, device);
}
comms_status_t comms_psk_phase_estimate(const double* symbols, size_t n, uint32_t m, double* out, int32_t device) {
    COMMS_ARG(m >= 1 && m <= (1u << 20), "PSK order m out of range");
    return estimate_host(1, symbols, n, m, out, device);
}
comms_status_t comms_qam_phase_estimate(const double* symbols, size_t n, double* out, int32_t device) {
    return estimate_host(2, symbols, n, 4, out, device);
}

}  // extern "C"
