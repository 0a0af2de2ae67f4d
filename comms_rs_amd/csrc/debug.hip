// debug.hip -- memory-pattern probes used while tuning (not part of the public ABI).
#include "common.hpp"

namespace comms {

// mode 0: float4 grid-stride copy; mode 1: float2 grid-stride copy
template <typename V>
__global__ __launch_bounds__(256) void probe_copy_kernel(const V* __restrict__ in, V* __restrict__ out, size_t n) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = in[i];
}

// mode 2: the os1024 access pattern without the math: a wave owns a run of 768-sample
// tiles; per tile 12 x (64 lanes x 8 B) loads, then 12 x 512-B stores.
template <int TPL>
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
    } else {
        const size_t ntiles = n_c32 / 768, runs = static_cast<size_t>(waves_per_cu) * kNumCU;
        probe_tile4_kernel<<<dim3((runs + 3) / 4), dim3(256), 0, s>>>(static_cast<const float4*>(d_in), static_cast<float4*>(d_out), ntiles, runs);
    }
    return launch_ok("probe kernel");
}
