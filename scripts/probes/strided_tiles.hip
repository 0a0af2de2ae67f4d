// Probe: what do small strided pieces cost on the READ side against the WRITE side of a 16384-point tile pass?
// A workgroup of 1024 lanes owns tiles of 16384 complex points (128 KiB), one workgroup per CU (128 KiB of LDS
// requested), next tile prefetched into registers while the current one is stored -- the shape of fft_rx1024_kernel
// without the arithmetic.  One side of the copy is contiguous, the other is XPT adjacent columns (pieces of XPT * 8 B)
// of a [16384 / XPT][1024] matrix, i.e. pieces at a stride of 8 KiB.
//   mode r: gather the pieces, store contiguous      mode w: load contiguous, scatter the pieces
//   grouped: the 16 / XPT tiles that share 128-byte lines run on one XCD in the same step (1) or in consecutive steps of one
//            workgroup (2)
// build: hipcc --offload-arch=gfx950 -O3 scripts/probes/strided_tiles.hip -o comms_rs_amd/lib/probe_strided_tiles
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

template <int XPT, bool GATHER>
__global__ __launch_bounds__(1024) void k(const float2* __restrict__ in, float2* __restrict__ out, size_t n_tiles, int grouped, unsigned cols) {
    extern __shared__ char smem[];
    constexpr unsigned NR = 16384 / XPT;        // rows of the matrix one tile spans
    constexpr unsigned R = 16 / XPT;             // tiles per 128-byte line
    const unsigned tid = threadIdx.x;
    const unsigned tpx = cols / XPT;             // tiles per matrix
    auto tile_of = [&](size_t g) -> size_t {
        if (!grouped) return blockIdx.x + g * gridDim.x;
        if (grouped == 2) return static_cast<size_t>(R) * (blockIdx.x + 256u * (g / R)) + g % R;  // the R tiles of a line in R consecutive steps of one workgroup
        const unsigned x = blockIdx.x & 7u, y = blockIdx.x >> 3;
        return static_cast<size_t>(R) * (x + 8u * (y / R) + (256u / R) * g) + (y % R);
    };
    auto strided = [&](size_t tix, unsigned e) -> size_t {
        const size_t m = tix / tpx;
        const unsigned c0 = static_cast<unsigned>(tix % tpx) * XPT;
        return m * (static_cast<size_t>(NR) * cols) + static_cast<size_t>(e / XPT) * cols + c0 + e % XPT;
    };
    const size_t n_steps = (n_tiles + gridDim.x - 1) / gridDim.x;
    float2 pre[16], cur[16];
    auto fetch = [&](size_t tix) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const unsigned e = 1024u * u + tid;
            pre[u] = GATHER ? in[strided(tix, e)] : in[tix * 16384u + e];
        }
    };
    if (tile_of(0) < n_tiles) fetch(tile_of(0));
    for (size_t g = 0; g < n_steps; ++g) {
        const size_t tix = tile_of(g);
        if (tix >= n_tiles) continue;
#pragma unroll
        for (int u = 0; u < 16; ++u) cur[u] = pre[u];
        if (g + 1 < n_steps && tile_of(g + 1) < n_tiles) fetch(tile_of(g + 1));
        if (smem[0] == 77) cur[0].x += 1.f;  // keeps the LDS request alive
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const unsigned e = 1024u * u + tid;
            if (GATHER) out[tix * 16384u + e] = cur[u];
            else out[strided(tix, e)] = cur[u];
        }
    }
}

template <int XPT, bool G>
static float run(const float2* in, float2* out, size_t n_tiles, int grouped, unsigned cols) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<XPT, G>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    std::vector<float> ts;
    for (int it = 0; it < 12; ++it) {
        hipEventRecord(a);
        k<XPT, G><<<256, 1024, 128 * 1024>>>(in, out, n_tiles, grouped, cols);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (it >= 4) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main() {
    const size_t n = size_t(1) << 26;  // 512 MiB each way
    float2 *in, *out;
    hipMalloc(&in, n * 9); hipMalloc(&out, n * 9);
    hipMemset(in, 0, n * 9); hipMemset(out, 0, n * 9);
    const size_t n_tiles = n / 16384;
    printf("2^26 points per launch (512 MiB read + 512 MiB written), 256 workgroups of 1024 lanes, median of 8\n");
    for (unsigned cols : {1024u, 1040u}) {
        printf("matrix rows of %u points (stride %u B)\n", cols, cols * 8);
#define ROW(X) \
        { float r0 = run<X, true>(in, out, n_tiles, 0, cols), r1 = run<X, true>(in, out, n_tiles, 1, cols), r2 = run<X, true>(in, out, n_tiles, 2, cols); \
          float w0 = run<X, false>(in, out, n_tiles, 0, cols), w1 = run<X, false>(in, out, n_tiles, 1, cols), w2 = run<X, false>(in, out, n_tiles, 2, cols); \
          printf("  pieces of %3d B: gather %.3f ms (grouped on an XCD %.3f, in time %.3f)   scatter %.3f ms (grouped %.3f, in time %.3f)\n", X * 8, r0, r1, r2, w0, w1, w2); }
        ROW(16) ROW(8) ROW(4) ROW(2) ROW(1)
    }
    return 0;
}
