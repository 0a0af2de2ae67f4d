// fir_f64.hip -- FirNode / BatchFirNode / PulseNode over Complex<f64>.
//
// The reference's fir(), batch_fir() and PulseNode are generic over T: Num + Copy (src/filter/fir.rs:43-54, :87-102;
// src/pulse.rs:38-93); its own doc example of batch_fir (fir.rs:68-86) and its timing estimator (src/demodulation/
// timing_estimator.rs:102-103) instantiate them on Complex<f64>.  Every BASELINE config is f32 -- those are the tuned kernels
// of fir.hip -- but a graph that carries f64 samples must find its nodes too (round 5; until then such graphs stayed on the CPU).
//
// Arithmetic: exactly the reference's, so that the result is BIT-IDENTICAL to it (the tests compare for equality): per output the taps
// are walked k ascending, each product is num::Complex's (ar*br - ai*bi, ar*bi + ai*br) -- four multiplications, one
// subtraction, one addition, no FMA (-ffp-contract=off, Makefile) -- and the sum folds from zero in that order
// (`taps.iter().zip(state).map(|(x, y)| x * y).sum()`, fir.rs:53,99).  f64 has no tolerance to hide behind.
//
// One kernel serves all three nodes: out[m * sps + p] = sum_j taps[p + j * sps] * x[m - j] (sps = 1: the FIR; the pulse
// shaper's zero-stuffed samples contribute products by zero: +0 terms that leave every sum as it is, so they are skipped).
// A plain tiled form -- taps in LDS, inputs through the cache, one output per thread: FP64 work (8 flops per tap and output at
// the chip's 78 TFLOP/s vector rate), not a roofline claim.
#include <vector>

#include "common.hpp"

namespace comms {

__device__ __forceinline__ double2 stream_at_f64(const double2* __restrict__ in, const double2* __restrict__ hist,
                                                 int hist_len, long long g, size_t n) {
    if (g >= 0) return static_cast<size_t>(g) < n ? in[g] : make_double2(0.0, 0.0);
    return g >= -static_cast<long long>(hist_len) ? hist[hist_len + g] : make_double2(0.0, 0.0);
}

constexpr int FD_TAPS_LDS = 2048;  // taps staged per pass (32 KiB)

// PULSE: the zero-stuffed form (PulseNode): out[m * sps + p] walks taps p, p + sps, ...; the reference multiplies the
// stuffed zeros as well, which adds +0.0 (or -0.0) products to a sum that started from +0.0: the value is unchanged
// except for the sign of an exactly-zero result, which is +0.0 either way (a sum of +0.0 and -0.0 terms folded from +0.0).
__global__ __launch_bounds__(256) void fir_f64_kernel(const double2* __restrict__ in, const double2* __restrict__ hist,
                                                      int hist_len, const double2* __restrict__ taps, int n_taps, int sps,
                                                      double2* __restrict__ out, size_t n_in, double2* __restrict__ new_hist) {
    __shared__ double2 tp[FD_TAPS_LDS];
    if (blockIdx.x == 0)  // new_hist = last hist_len samples of concat(old_hist, in)
        for (int j = threadIdx.x; j < hist_len; j += blockDim.x) {
            const size_t p = n_in + static_cast<size_t>(j);
            new_hist[j] = p < static_cast<size_t>(hist_len) ? hist[p] : in[p - hist_len];
        }
    const size_t n_out = n_in * static_cast<size_t>(sps);
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    const size_t rounds = (n_out + stride - 1) / stride;
    for (size_t r = 0; r < rounds; ++r) {
        const size_t i = r * stride + static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
        const size_t m = i / sps;
        const int p = static_cast<int>(i - m * sps);
        double ar = 0.0, ai = 0.0;
        for (int k0 = 0; k0 < n_taps; k0 += FD_TAPS_LDS) {
            const int kc = n_taps - k0 < FD_TAPS_LDS ? n_taps - k0 : FD_TAPS_LDS;
            __syncthreads();
            for (int k = threadIdx.x; k < kc; k += blockDim.x) tp[k] = taps[k0 + k];
            __syncthreads();
            if (i < n_out) {
                int j = k0 > p ? (k0 - p + sps - 1) / sps : 0;
                for (int k = p + j * sps; k < k0 + kc; k += sps, ++j) {
                    const double2 t = tp[k - k0];
                    const double2 x = stream_at_f64(in, hist, hist_len, static_cast<long long>(m) - j, n_in);
                    const double pr = t.x * x.x - t.y * x.y;   // num::Complex Mul (no contraction: -ffp-contract=off)
                    const double pi = t.x * x.y + t.y * x.x;
                    ar = ar + pr;                              // Sum: fold from zero, taps ascending
                    ai = ai + pi;
                }
            }
        }
        if (i < n_out) out[i] = make_double2(ar, ai);
    }
}

}  // namespace comms

using namespace comms;

struct comms_fir_f64 : Handle {
    int n_eff = 0;      // taps that take part
    int sps = 1;
    int hist_len = 0;   // samples (FIR: n_eff) or symbols (pulse: ceil(n_taps / sps)) of history
    double2* d_taps = nullptr;
    double2* d_hist[2] = {nullptr, nullptr};
    int cur = 0;
};
struct comms_pulse_f64 : comms_fir_f64 {};

static void free_f64(comms_fir_f64* h) {
    (void)use_device(h->device);
    if (h->d_taps) (void)hipFree(h->d_taps);
    if (h->d_hist[0]) (void)hipFree(h->d_hist[0]);
    if (h->d_hist[1]) (void)hipFree(h->d_hist[1]);
    h->fini();
}

static void ring_from_state(std::vector<double2>& ring, const comms_c64* state, size_t n_state) {
    // device history is time-ordered (oldest first); the reference's state is newest first
    const size_t hl = ring.size();
    for (size_t k = 0; k < hl && k < n_state; ++k) ring[hl - 1 - k] = make_double2(state[k].re, state[k].im);
}

template <class H>
static comms_status_t create_f64(const comms_c64* taps, size_t n_eff, int sps, size_t hist_len, const comms_c64* state,
                                 size_t n_state, int32_t device, H** out) {
    H* h = new (std::nothrow) H;
    COMMS_ARG(h != nullptr, "out of host memory");
    comms_status_t st = h->init(device);
    if (st != COMMS_OK) {
        delete h;
        return st;
    }
    h->n_eff = static_cast<int>(n_eff);
    h->sps = sps;
    h->hist_len = static_cast<int>(hist_len);
    std::vector<double2> ring(hist_len, make_double2(0.0, 0.0));
    ring_from_state(ring, state, n_state);
    hipError_t e = hipMalloc(&h->d_taps, n_eff * sizeof(double2));
    if (e == hipSuccess) e = hipMemcpy(h->d_taps, taps, n_eff * sizeof(double2), hipMemcpyHostToDevice);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        e = hipMalloc(&h->d_hist[i], (hist_len ? hist_len : 1) * sizeof(double2));
        if (e == hipSuccess && hist_len) e = hipMemcpy(h->d_hist[i], ring.data(), hist_len * sizeof(double2), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        free_f64(h);
        delete h;
        return fail(COMMS_ERR_DEVICE, "f64 FIR alloc: %s", hipGetErrorString(e));
    }
    *out = h;
    return COMMS_OK;
}

static comms_status_t run_f64_dev(comms_fir_f64* h, const comms_c64* d_in, size_t n, comms_c64* d_out, void* stream) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG((d_in && d_out) || !n, "NULL device pointer");
    COMMS_TRY(use_device(h->device));
    if (!n) return COMMS_OK;
    COMMS_ARG(n <= SIZE_MAX / 16 / static_cast<size_t>(h->sps), "n * sam_per_sym overflows");
    COMMS_ARG(!ranges_overlap(d_in, n * 16, d_out, n * h->sps * 16), "the f64 FIR cannot run in place");
    COMMS_ARG(((reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_out)) & 15) == 0, "pointers must be aligned to one sample");
    hipStream_t s = nullptr;
    COMMS_TRY(h->enter(stream, &s));
    const size_t n_out = n * static_cast<size_t>(h->sps);
    size_t blocks = (n_out + 255) / 256;
    if (blocks > 8u * kNumCU) blocks = 8u * kNumCU;
    h->tic(s);
    fir_f64_kernel<<<dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s>>>(
        reinterpret_cast<const double2*>(d_in), h->d_hist[h->cur], h->hist_len, h->d_taps, h->n_eff, h->sps,
        reinterpret_cast<double2*>(d_out), n, h->d_hist[h->cur ^ 1]);
    h->toc(s);
    COMMS_TRY(launch_ok("fir_f64_kernel"));
    h->cur ^= 1;
    return COMMS_OK;
}

static comms_status_t run_f64_host(comms_fir_f64* h, const comms_c64* in, size_t n, comms_c64* out) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG((in && out) || !n, "NULL host pointer");
    COMMS_TRY(use_device(h->device));
    if (!n) return COMMS_OK;
    return h->run_host_units(in, n * 16, 16, out, n * h->sps * 16, static_cast<size_t>(h->sps) * 16, [&](void* d_in, void* d_out, size_t ib, size_t) {
        return run_f64_dev(h, static_cast<const comms_c64*>(d_in), ib / 16, static_cast<comms_c64*>(d_out), COMMS_STREAM_HANDLE);
    });
}

extern "C" {

comms_status_t comms_fir_f64_create(const comms_c64* taps, size_t n_taps, const comms_c64* state, size_t n_state,
                                    int32_t device, comms_fir_f64_t** out) {
    COMMS_ARG(out != nullptr, "out is NULL");
    *out = nullptr;
    COMMS_ARG(taps != nullptr && n_taps > 0, "taps must hold at least one tap (the reference panics on an empty state)");
    COMMS_ARG(state == nullptr || n_state > 0, "a user state must hold at least one sample");
    size_t n_eff = n_taps;
    if (state && n_state < n_eff) n_eff = n_state;  // zip(taps, state), fir.rs:53
    COMMS_ARG(n_eff <= (1u << 20), "too many taps (%zu)", n_eff);
    return create_f64(taps, n_eff, 1, n_eff, state, state ? n_state : 0, device, out);
}
comms_status_t comms_fir_f64_run(comms_fir_f64_t* h, const comms_c64* in, size_t n, comms_c64* out) {
    return run_f64_host(h, in, n, out);
}
comms_status_t comms_fir_f64_run_dev(comms_fir_f64_t* h, const comms_c64* d_in, size_t n, comms_c64* d_out, void* stream) {
    return run_f64_dev(h, d_in, n, d_out, stream);
}
comms_status_t comms_fir_f64_get_state(comms_fir_f64_t* h, comms_c64* state, size_t n_state) {
    COMMS_ARG(h && state, "NULL argument");
    COMMS_ARG(n_state <= static_cast<size_t>(h->hist_len), "n_state %zu exceeds the %d effective taps", n_state, h->hist_len);
    COMMS_TRY(use_device(h->device));
    COMMS_TRY(h->quiesce());
    std::vector<double2> ring(h->hist_len);
    COMMS_HIP_TRY(hipMemcpy(ring.data(), h->d_hist[h->cur], ring.size() * sizeof(double2), hipMemcpyDeviceToHost));
    for (size_t k = 0; k < n_state; ++k) {
        state[k].re = ring[h->hist_len - 1 - k].x;
        state[k].im = ring[h->hist_len - 1 - k].y;
    }
    return COMMS_OK;
}
comms_status_t comms_fir_f64_set_state(comms_fir_f64_t* h, const comms_c64* state, size_t n_state) {
    COMMS_ARG(h && state, "NULL argument");
    COMMS_ARG(n_state == static_cast<size_t>(h->hist_len), "state must hold exactly the %d effective taps", h->hist_len);
    COMMS_TRY(use_device(h->device));
    COMMS_TRY(h->quiesce());  // no pending launch may still read the buffer that is overwritten
    std::vector<double2> ring(h->hist_len, make_double2(0.0, 0.0));
    ring_from_state(ring, state, n_state);
    COMMS_HIP_TRY(hipMemcpy(h->d_hist[h->cur], ring.data(), ring.size() * sizeof(double2), hipMemcpyHostToDevice));
    return COMMS_OK;
}
comms_status_t comms_fir_f64_destroy(comms_fir_f64_t* h) {
    if (!h) return COMMS_OK;
    free_f64(h);
    delete h;
    return COMMS_OK;
}

comms_status_t comms_pulse_f64_create(const comms_c64* taps, size_t n_taps, size_t sam_per_sym, int32_t device,
                                      comms_pulse_f64_t** out) {
    COMMS_ARG(out != nullptr, "out is NULL");
    *out = nullptr;
    COMMS_ARG(taps != nullptr && n_taps > 0, "taps must hold at least one tap");
    COMMS_ARG(sam_per_sym >= 1 && sam_per_sym <= (1u << 16), "sam_per_sym must be in [1, 65536] (0 underflows in the reference)");
    COMMS_ARG(n_taps <= (1u << 20), "too many taps (%zu)", n_taps);
    const size_t hist = (n_taps + sam_per_sym - 1) / sam_per_sym;  // symbols the filter reaches back over
    return create_f64(taps, n_taps, static_cast<int>(sam_per_sym), hist, nullptr, 0, device, out);
}
comms_status_t comms_pulse_f64_run(comms_pulse_f64_t* h, const comms_c64* sym, size_t n_sym, comms_c64* out) {
    return run_f64_host(h, sym, n_sym, out);
}
comms_status_t comms_pulse_f64_run_dev(comms_pulse_f64_t* h, const comms_c64* d_sym, size_t n_sym, comms_c64* d_out,
                                       void* stream) {
    return run_f64_dev(h, d_sym, n_sym, d_out, stream);
}
comms_status_t comms_pulse_f64_destroy(comms_pulse_f64_t* h) {
    if (!h) return COMMS_OK;
    free_f64(h);
    delete h;
    return COMMS_OK;
}

}  // extern "C"
