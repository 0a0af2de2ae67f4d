// fir_int.hip -- FirNode / BatchFirNode / PulseNode over Complex<i16>.
//
// The reference's fir(), batch_fir() and PulseNode are generic over T: Num + Copy
// (src/filter/fir.rs:43-54, :87-102; src/pulse.rs:38-93) and its own tests run them on
// Complex<i16> (src/filter/fir_node.rs:259-313, src/pulse.rs:129-183).  Every BASELINE config
// is f32 -- those are the tuned kernels of fir.hip -- but a graph that carries integer samples
// must find its nodes too, with the integer type's arithmetic: products and sums wrap modulo
// 2^16 (Rust release builds; a debug build panics on overflow instead).  Z / 2^16 is a ring
// quotient of Z / 2^32, so the kernel accumulates in 32-bit wrapping arithmetic and truncates
// once: bit-identical to wrapping at every step.
//
// One kernel serves both nodes: out[m * sps + p] = sum_j taps[p + j * sps] * x[m - j]  (sps = 1:
// the FIR; k = p + j * sps ascending as fir() walks it, though the order cannot matter here).
// A plain tiled form -- taps in LDS, inputs through the cache -- and no claim on the roofline.
#include <vector>

#include "common.hpp"

namespace comms {

__device__ __forceinline__ short2 stream_at_i16(const short2* __restrict__ in, const short2* __restrict__ hist,
                                                int hist_len, long long g, size_t n) {
    if (g >= 0) return static_cast<size_t>(g) < n ? in[g] : make_short2(0, 0);
    return g >= -static_cast<long long>(hist_len) ? hist[hist_len + g] : make_short2(0, 0);
}

constexpr int FI_TAPS_LDS = 4096;  // taps staged per pass

__global__ __launch_bounds__(256) void fir_i16_kernel(const short2* __restrict__ in, const short2* __restrict__ hist,
                                                      int hist_len, const short2* __restrict__ taps, int n_taps,
                                                      int sps, short2* __restrict__ out, size_t n_in,
                                                      short2* __restrict__ new_hist) {
    __shared__ short2 tp[FI_TAPS_LDS];
    // new_hist = last hist_len samples of concat(old_hist, in)
    if (blockIdx.x == 0)
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
        unsigned ar = 0, ai = 0;  // wrapping accumulators
        for (int k0 = 0; k0 < n_taps; k0 += FI_TAPS_LDS) {
            const int kc = n_taps - k0 < FI_TAPS_LDS ? n_taps - k0 : FI_TAPS_LDS;
            __syncthreads();
            for (int k = threadIdx.x; k < kc; k += blockDim.x) tp[k] = taps[k0 + k];
            __syncthreads();
            if (i < n_out) {
                // taps k = p + j * sps inside [k0, k0 + kc)
                int j = k0 > p ? (k0 - p + sps - 1) / sps : 0;
                for (int k = p + j * sps; k < k0 + kc; k += sps, ++j) {
                    const short2 t = tp[k - k0];
                    const short2 x = stream_at_i16(in, hist, hist_len, static_cast<long long>(m) - j, n_in);
                    ar += static_cast<unsigned>(static_cast<int>(t.x) * x.x - static_cast<int>(t.y) * x.y);
                    ai += static_cast<unsigned>(static_cast<int>(t.x) * x.y + static_cast<int>(t.y) * x.x);
                }
            }
        }
        if (i < n_out) out[i] = make_short2(static_cast<short>(ar & 0xffffu), static_cast<short>(ai & 0xffffu));
    }
}

}  // namespace comms

using namespace comms;

// one handle type for both nodes (sps = 1: FIR with the reference's `state` semantics)
struct comms_fir_i16 : Handle {
    int n_eff = 0;      // taps that take part
    int sps = 1;
    int hist_len = 0;   // samples (FIR: n_eff) or symbols (pulse: ceil(n_taps / sps)) of history
    short2* d_taps = nullptr;
    short2* d_hist[2] = {nullptr, nullptr};
    int cur = 0;
};
struct comms_pulse_i16 : comms_fir_i16 {};

static void free_int(comms_fir_i16* h) {
    (void)use_device(h->device);
    if (h->d_taps) (void)hipFree(h->d_taps);
    if (h->d_hist[0]) (void)hipFree(h->d_hist[0]);
    if (h->d_hist[1]) (void)hipFree(h->d_hist[1]);
    h->fini();
}

template <class H>
static comms_status_t create_int(const comms_c16* taps, size_t n_eff, int sps, size_t hist_len, const comms_c16* state,
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
    // device history is time-ordered (oldest first); the reference's state is newest first
    std::vector<short2> ring(hist_len, make_short2(0, 0));
    for (size_t k = 0; k < hist_len && k < n_state; ++k) ring[hist_len - 1 - k] = make_short2(state[k].re, state[k].im);
    hipError_t e = hipMalloc(&h->d_taps, n_eff * sizeof(short2));
    if (e == hipSuccess) e = hipMemcpy(h->d_taps, taps, n_eff * sizeof(short2), hipMemcpyHostToDevice);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        e = hipMalloc(&h->d_hist[i], (hist_len ? hist_len : 1) * sizeof(short2));
        if (e == hipSuccess && hist_len) e = hipMemcpy(h->d_hist[i], ring.data(), hist_len * sizeof(short2), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        free_int(h);
        delete h;
        return fail(COMMS_ERR_DEVICE, "integer FIR alloc: %s", hipGetErrorString(e));
    }
    *out = h;
    return COMMS_OK;
}

static comms_status_t run_int_dev(comms_fir_i16* h, const comms_c16* d_in, size_t n, comms_c16* d_out, void* stream) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG((d_in && d_out) || !n, "NULL device pointer");
    COMMS_TRY(use_device(h->device));
    if (!n) return COMMS_OK;
    COMMS_ARG(n <= SIZE_MAX / 4 / static_cast<size_t>(h->sps), "n * sam_per_sym overflows");
    COMMS_ARG(!ranges_overlap(d_in, n * 4, d_out, n * h->sps * 4), "the integer FIR cannot run in place");
    COMMS_ARG(((reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_out)) & 3) == 0, "pointers must be aligned to one sample");
    hipStream_t s = nullptr;
    COMMS_TRY(h->enter(stream, &s));
    const size_t n_out = n * static_cast<size_t>(h->sps);
    size_t blocks = (n_out + 255) / 256;
    if (blocks > 8u * kNumCU) blocks = 8u * kNumCU;
    h->tic(s);
    fir_i16_kernel<<<dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s>>>(
        reinterpret_cast<const short2*>(d_in), h->d_hist[h->cur], h->hist_len, h->d_taps, h->n_eff, h->sps,
        reinterpret_cast<short2*>(d_out), n, h->d_hist[h->cur ^ 1]);
    h->toc(s);
    COMMS_TRY(launch_ok("fir_i16_kernel"));
    h->cur ^= 1;
    return COMMS_OK;
}

static comms_status_t run_int_host(comms_fir_i16* h, const comms_c16* in, size_t n, comms_c16* out) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG((in && out) || !n, "NULL host pointer");
    COMMS_TRY(use_device(h->device));
    if (!n) return COMMS_OK;
    return h->run_host_units(in, n * 4, 4, out, n * h->sps * 4, static_cast<size_t>(h->sps) * 4, [&](void* d_in, void* d_out, size_t ib, size_t) {
        return run_int_dev(h, static_cast<const comms_c16*>(d_in), ib / 4, static_cast<comms_c16*>(d_out), COMMS_STREAM_HANDLE);
    });
}

extern "C" {

comms_status_t comms_fir_i16_create(const comms_c16* taps, size_t n_taps, const comms_c16* state, size_t n_state,
                                    int32_t device, comms_fir_i16_t** out) {
    COMMS_ARG(out != nullptr, "out is NULL");
    *out = nullptr;
    COMMS_ARG(taps != nullptr && n_taps > 0, "taps must hold at least one tap (the reference panics on an empty state)");
    COMMS_ARG(state == nullptr || n_state > 0, "a user state must hold at least one sample");
    size_t n_eff = n_taps;
    if (state && n_state < n_eff) n_eff = n_state;  // zip(taps, state), fir.rs:53
    COMMS_ARG(n_eff <= (1u << 20), "too many taps (%zu)", n_eff);
    return create_int(taps, n_eff, 1, n_eff, state, state ? n_state : 0, device, out);
}
comms_status_t comms_fir_i16_run(comms_fir_i16_t* h, const comms_c16* in, size_t n, comms_c16* out) {
    return run_int_host(h, in, n, out);
}
comms_status_t comms_fir_i16_run_dev(comms_fir_i16_t* h, const comms_c16* d_in, size_t n, comms_c16* d_out, void* stream) {
    return run_int_dev(h, d_in, n, d_out, stream);
}
comms_status_t comms_fir_i16_get_state(comms_fir_i16_t* h, comms_c16* state, size_t n_state) {
    COMMS_ARG(h && state, "NULL argument");
    COMMS_ARG(n_state <= static_cast<size_t>(h->hist_len), "n_state %zu exceeds the %d effective taps", n_state, h->hist_len);
    COMMS_TRY(use_device(h->device));
    COMMS_TRY(h->quiesce());
    std::vector<short2> ring(h->hist_len);
    COMMS_HIP_TRY(hipMemcpy(ring.data(), h->d_hist[h->cur], ring.size() * sizeof(short2), hipMemcpyDeviceToHost));
    for (size_t k = 0; k < n_state; ++k) {
        state[k].re = ring[h->hist_len - 1 - k].x;
        state[k].im = ring[h->hist_len - 1 - k].y;
    }
    return COMMS_OK;
}
comms_status_t comms_fir_i16_destroy(comms_fir_i16_t* h) {
    if (!h) return COMMS_OK;
    free_int(h);
    delete h;
    return COMMS_OK;
}

comms_status_t comms_pulse_i16_create(const comms_c16* taps, size_t n_taps, size_t sam_per_sym, int32_t device,
                                      comms_pulse_i16_t** out) {
    COMMS_ARG(out != nullptr, "out is NULL");
    *out = nullptr;
    COMMS_ARG(taps != nullptr && n_taps > 0, "taps must hold at least one tap");
    COMMS_ARG(sam_per_sym >= 1 && sam_per_sym <= (1u << 16), "sam_per_sym must be in [1, 65536] (0 underflows in the reference)");
    COMMS_ARG(n_taps <= (1u << 20), "too many taps (%zu)", n_taps);
    const size_t hist = (n_taps + sam_per_sym - 1) / sam_per_sym;  // symbols the filter reaches back over
    return create_int(taps, n_taps, static_cast<int>(sam_per_sym), hist, nullptr, 0, device, out);
}
comms_status_t comms_pulse_i16_run(comms_pulse_i16_t* h, const comms_c16* sym, size_t n_sym, comms_c16* out) {
    return run_int_host(h, sym, n_sym, out);
}
comms_status_t comms_pulse_i16_run_dev(comms_pulse_i16_t* h, const comms_c16* d_sym, size_t n_sym, comms_c16* d_out,
                                       void* stream) {
    return run_int_dev(h, d_sym, n_sym, d_out, stream);
}
comms_status_t comms_pulse_i16_destroy(comms_pulse_i16_t* h) {
    if (!h) return COMMS_OK;
    free_int(h);
    delete h;
    return COMMS_OK;
}

}  // extern "C"
