// fir_handle.hpp -- the FIR node handle and the stream helpers shared by fir.hip and
// fir_decim.hip.
#pragma once

#include <vector>

#include "common.hpp"

namespace comms {

// Folded into every FIR kernel (workgroup 0): new_hist = last HL samples of
// concat(old_hist[HL], in[n]) -- the reference's `state` after the batch.
__device__ __forceinline__ void hist_advance(const float2* __restrict__ old_hist,
                                             const float2* __restrict__ in, size_t n,
                                             float2* __restrict__ new_hist, int HL) {
    if (blockIdx.x != 0) return;
    for (int j = threadIdx.x; j < HL; j += blockDim.x) {
        const size_t p = n + static_cast<size_t>(j);
        new_hist[j] = p < static_cast<size_t>(HL) ? old_hist[p] : in[p - HL];
    }
}

// Sample g of the logical stream [history | input | zeros]
__device__ __forceinline__ float2 stream_at(const float2* __restrict__ in,
                                            const float2* __restrict__ hist, int hist_len,
                                            long long g, size_t n) {
    if (g >= 0) return static_cast<size_t>(g) < n ? in[g] : make_float2(0.f, 0.f);
    return g >= -static_cast<long long>(hist_len) ? hist[hist_len + g] : make_float2(0.f, 0.f);
}

// FM::demod step (reference src/modulation/analog.rs:27-28): arg(x * conj(p)), unfused
__device__ __forceinline__ float fm_step(float2 x, float2 p) {
    const float pcr = p.x, pci = -p.y;
    const float re = x.x * pcr - x.y * pci;
    const float im = x.x * pci + x.y * pcr;
    return atan2f(im, re);
}

}  // namespace comms

struct comms_fir : comms::Handle {
    int n_eff = 0;       // taps that take part: min(n_taps, n_state)
    bool real_taps = false;
    int algo = COMMS_FIR_AUTO;
    bool os1024_fixed = false;  // COMMS_FIR_OS1024_FIXED: never the ticketed kernel
    // direct form
    int NP = 0;          // taps padded to a multiple of 8
    float2* d_taps_pad = nullptr;
    // overlap-save, F = 1024 (wave per segment)
    bool w_ready = false;
    float2* d_wtw1 = nullptr;
    float2* d_wtw2 = nullptr;
    float2* d_whdev = nullptr;
    // overlap-save, F = 4096 (workgroup per segment)
    bool os_ready = false;
    int hblk = 0;        // halo = 256*hblk >= (taps per partition) - 1
    // overlap-save, F = 16384 (workgroup per segment), > 2049 taps; partitions of 4097 taps
    bool x_ready = false;
    int x_part = 1;
    float2* d_xt[4] = {nullptr, nullptr, nullptr, nullptr};  // tw1, tw2, ta, tb
    std::vector<float2*> d_xh;                                // spectrum per partition
    int n_part = 1;      // (4096-pt kernel, forced) > 3841 taps: partitions of OS_PART taps
    std::vector<float2*> d_hparts;  // filter spectrum per partition (d_hdev = partition 0)
    float2* d_tw1 = nullptr;
    float2* d_tw2 = nullptr;
    float2* d_hdev = nullptr;
    // history: last n_eff input samples, time order, ping-pong
    float2* d_hist[2] = {nullptr, nullptr};
    int cur = 0;
    std::vector<comms_c32> taps;  // effective taps (host copy)
};

