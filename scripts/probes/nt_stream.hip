// Probe: non-temporal loads / stores on the two pointwise shapes of the headline chain's tail --
//   copy16: 16 B per lane in and out (the mixer's traffic shape), 2^24 and 2^26 Complex<f32>
//   dec8:   out[j] = in[8 j] (DecimateNode at rate 8)
// each as plain / nt load / nt store / both.
// build: hipcc --offload-arch=gfx950 -O3 scripts/probes/nt_stream.hip -o comms_rs_amd/lib/probe_nt_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int NTL, int NTS>
__global__ __launch_bounds__(256) void copy16(const f4* __restrict__ in, f4* __restrict__ out, size_t n4) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f4 v = NTL ? __builtin_nontemporal_load(in + i) : in[i];
        v.x += 1.0f;
        if (NTS) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}
// U loads in flight per lane (the product mixer has one)
template <int U>
__global__ __launch_bounds__(256) void copy16u(const f4* __restrict__ in, f4* __restrict__ out, size_t n4) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = in[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { v[u].x += 1.0f; out[i + u * stride] = v[u]; }
    }
    for (; i < n4; i += stride) { f4 v = in[i]; v.x += 1.0f; out[i] = v; }
}
template <int NTL, int NTS>
__global__ __launch_bounds__(256) void dec8(const f2* __restrict__ in, f2* __restrict__ out, size_t n_out) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t j = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; j < n_out; j += stride) {
        const f2 v = NTL ? __builtin_nontemporal_load(in + j * 8) : in[j * 8];
        if (NTS) __builtin_nontemporal_store(v, out + j); else out[j] = v;
    }
}
template <class F>
static float timeit(F f) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    std::vector<float> ts;
    for (int it = 0; it < 70; ++it) {
        hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (it >= 30) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2] * 1e3f;
}
int main() {
    for (int lg : {24, 26}) {
        const size_t n = size_t(1) << lg;
        float2 *in, *out;
        hipMalloc(&in, n * 8); hipMalloc(&out, n * 8);
        hipMemset(in, 0, n * 8);
        const unsigned blocks = 8 * 256;
        const size_t n4 = n / 2, no = n / 8;
        printf("2^%d samples  copy16: plain %.1f  nt-load %.1f  nt-store %.1f  both %.1f us\n", lg,
               timeit([&] { copy16<0, 0><<<blocks, 256>>>((const f4*)in, (f4*)out, n4); }),
               timeit([&] { copy16<1, 0><<<blocks, 256>>>((const f4*)in, (f4*)out, n4); }),
               timeit([&] { copy16<0, 1><<<blocks, 256>>>((const f4*)in, (f4*)out, n4); }),
               timeit([&] { copy16<1, 1><<<blocks, 256>>>((const f4*)in, (f4*)out, n4); }));
        printf("2^%d samples  copy16, loads in flight per lane: 1 %.1f  2 %.1f  4 %.1f us;  16 workgroups per CU, 2 in flight %.1f us\n", lg,
               timeit([&] { copy16u<1><<<blocks, 256>>>((const f4*)in, (f4*)out, n4); }),
               timeit([&] { copy16u<2><<<blocks, 256>>>((const f4*)in, (f4*)out, n4); }),
               timeit([&] { copy16u<4><<<blocks, 256>>>((const f4*)in, (f4*)out, n4); }),
               timeit([&] { copy16u<2><<<2 * blocks, 256>>>((const f4*)in, (f4*)out, n4); }));
        printf("2^%d samples  dec8:   plain %.1f  nt-load %.1f  nt-store %.1f  both %.1f us\n", lg,
               timeit([&] { dec8<0, 0><<<blocks, 256>>>((const f2*)in, (f2*)out, no); }),
               timeit([&] { dec8<1, 0><<<blocks, 256>>>((const f2*)in, (f2*)out, no); }),
               timeit([&] { dec8<0, 1><<<blocks, 256>>>((const f2*)in, (f2*)out, no); }),
               timeit([&] { dec8<1, 1><<<blocks, 256>>>((const f2*)in, (f2*)out, no); }));
        hipFree(in); hipFree(out);
    }
    return 0;
}
