// Probe: forms of out[j] = in[j * R] for 8-byte elements (DecimateNode on Complex<f32>), 2^24 inputs.
//   a: one strided 8-byte load per lane and sweep (the product kernel up to round 3)
//   b: the same, four loads in flight per lane
//   c: whole lines read coalesced (16 B per lane, four loads in flight), the kept samples compacted through a per-wave LDS
//      block and stored as one run per wave
// build: hipcc --offload-arch=gfx950 -O3 scripts/probes/decimate_forms.hip -o comms_rs_amd/lib/probe_decimate_forms
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(256) void ka(const float2* __restrict__ in, float2* __restrict__ out, size_t n_out, size_t rate) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t j = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; j < n_out; j += stride) out[j] = in[j * rate];
}

__global__ __launch_bounds__(256) void kb(const float2* __restrict__ in, float2* __restrict__ out, size_t n_out, size_t rate) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    size_t j = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    for (; j + 3 * stride < n_out; j += 4 * stride) {
        float2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = in[(j + u * stride) * rate];
#pragma unroll
        for (int u = 0; u < 4; ++u) out[j + u * stride] = v[u];
    }
    for (; j < n_out; j += stride) out[j] = in[j * rate];
}

// a wave owns blocks of 512 inputs (4 KiB): lane l loads the 16-byte pairs l, 64 + l, 128 + l, 192 + l of the block
__global__ __launch_bounds__(256) void kc(const float2* __restrict__ in, float2* __restrict__ out, size_t n_in, unsigned rate) {
    __shared__ float2 keep[4][256];
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t n_blocks = n_in / 512;  // (probe: n_in a multiple of 512)
    const size_t waves = static_cast<size_t>(gridDim.x) * 4;
    const float inv = 1.0f / static_cast<float>(rate);
    for (size_t b = static_cast<size_t>(blockIdx.x) * 4 + wave; b < n_blocks; b += waves) {
        const size_t s0 = b * 512;
        const size_t j0 = (s0 + rate - 1) / rate;          // first output of the block
        const size_t j1 = (s0 + 512 + rate - 1) / rate;    // one past its last
        const unsigned r0 = static_cast<unsigned>(j0 * rate - s0);  // block-local index of the first kept sample
        float4 v[4];
        const float4* p = reinterpret_cast<const float4*>(in + s0);
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = p[u * 64 + lane];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned e = (u * 64 + lane) * 2;  // block-local index of v[u].xy
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned x = e + h;
                if (x >= r0) {
                    const unsigned d = x - r0;
                    const unsigned q = static_cast<unsigned>((static_cast<float>(d) + 0.5f) * inv);
                    if (q * rate == d) keep[wave][q] = h ? make_float2(v[u].z, v[u].w) : make_float2(v[u].x, v[u].y);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const unsigned cnt = static_cast<unsigned>(j1 - j0);
        for (unsigned i = lane; i < cnt; i += 64) out[j0 + i] = keep[wave][i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

template <class F>
static float timeit(F f) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    std::vector<float> ts;
    for (int it = 0; it < 60; ++it) {
        hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (it >= 30) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2] * 1e3f;
}

int main() {
    const size_t n = size_t(1) << 24;
    float2 *in, *out, *ref;
    hipMalloc(&in, n * 8); hipMalloc(&out, n * 8); hipMalloc(&ref, n * 8);
    std::vector<float2> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = make_float2(float(i & 0xffff), float(i >> 16));
    hipMemcpy(in, h.data(), n * 8, hipMemcpyHostToDevice);
    for (unsigned rate : {2u, 3u, 4u, 5u, 8u, 16u, 100u}) {
        const size_t n_out = (n + rate - 1) / rate;
        const unsigned blocks = static_cast<unsigned>(std::min<size_t>((n_out + 255) / 256, 8 * 256));
        float ta = timeit([&] { ka<<<blocks, 256>>>(in, ref, n_out, rate); });
        float tb = timeit([&] { kb<<<blocks, 256>>>(in, out, n_out, rate); });
        for (int g : {4, 8, 16}) {
            hipMemset(out, 0xff, n_out * 8);
            float tc = timeit([&] { kc<<<256 * g, 256>>>(in, out, n, rate); });
            std::vector<float2> x(n_out), y(n_out);
            hipMemcpy(x.data(), ref, n_out * 8, hipMemcpyDeviceToHost);
            hipMemcpy(y.data(), out, n_out * 8, hipMemcpyDeviceToHost);
            size_t bad = 0;
            for (size_t i = 0; i < n_out; ++i) bad += x[i].x != y[i].x || x[i].y != y[i].y;
            printf("rate %3u: strided %.1f us, four in flight %.1f us, coalesced + compaction (%d workgroups per CU) %.1f us%s\n", rate, ta, tb, g, tc,
                   bad ? "  MISMATCH" : "");
        }
    }
    return 0;
}
