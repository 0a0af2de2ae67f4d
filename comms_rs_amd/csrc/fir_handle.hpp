// fir_handle.hpp -- the FIR node handle and the stream helpers shared by fir.hip and
// fir_decim.hip.
#pragma once

#include <vector>

#include "common.hpp"

namespace comms {

// ---- raw-IQ input formats read by the first kernel of a chain: the wire
// samples are converted in registers on their way in, with exactly iqformat.hip's arithmetic, so HBM sees
// 4 B (i16) or 2 B (u8) per sample instead of 8 + the 12 / 10 B of a separate conversion pass.
//   i16: cast_complex::<i16,f32> (src/util/math.rs:20-28) of IQInput's samples (src/io/raw_iq.rs:16,50-51), times scale
//   u8 : (x as f32 - 127.5) / 127.5 (examples/fm_radio.rs:82-90, RTL-SDR bytes)
struct InC32 {
    const float2* p;
    __device__ __forceinline__ float2 operator[](size_t i) const { return p[i]; }
};
// Complex<f32> read as two 4-byte loads: the 4096-point FIR kernel's load stage needs ~70 VGPRs less this way and
// fits three workgroups per CU without spilling (fir.hip)
struct InC32Split {
    const float* p;
    __device__ __forceinline__ float2 operator[](size_t i) const { return make_float2(p[2 * i], p[2 * i + 1]); }
};
struct InI16 {
    const short2* p;
    float scale;
    __device__ __forceinline__ float2 operator[](size_t i) const {
        const short2 v = p[i];
        return make_float2(static_cast<float>(v.x) * scale, static_cast<float>(v.y) * scale);
    }
};
// The division by 127.5 is correctly rounded in iqformat.hip (and in the reference).  q = a * fl(1/127.5)
// followed by one residual step  q + fl(a - q * 127.5) * fl(1/127.5)  reproduces it for all 256 byte values
// (checked exhaustively: tests/test_cpu_host.py and the GPU test): 4 flops instead of a ~10-instruction divide.
struct InU8 {
    const uchar2* p;
    static __device__ __forceinline__ float cvt(float x) {
        constexpr float kInv = 1.0f / 127.5f;
        const float a = x - 127.5f;
        const float q = a * kInv;
        const float r = __builtin_fmaf(-q, 127.5f, a);
        return __builtin_fmaf(r, kInv, q);
    }
    __device__ __forceinline__ float2 operator[](size_t i) const {
        const uchar2 v = p[i];
        return make_float2(cvt(static_cast<float>(v.x)), cvt(static_cast<float>(v.y)));
    }
};

// The same view / pointer moved forward by `off` samples (a workgroup-uniform offset: the loads that follow keep
// an SGPR base and a small per-lane offset)
__device__ __forceinline__ const float2* view_at(const float2* p, size_t off) { return p + off; }
__device__ __forceinline__ InC32 view_at(InC32 v, size_t off) { return InC32{v.p + off}; }
__device__ __forceinline__ InC32Split view_at(InC32Split v, size_t off) { return InC32Split{v.p + 2 * off}; }
__device__ __forceinline__ InI16 view_at(InI16 v, size_t off) { return InI16{v.p + off, v.scale}; }
__device__ __forceinline__ InU8 view_at(InU8 v, size_t off) { return InU8{v.p + off}; }

// ---- buffer-addressed rows: an SGPR resource (base + byte count), ONE per-lane byte offset and a scalar offset
// per row.  Global loads with 8-KiB row strides cannot use immediate offsets (13 bits), and the compiler then keeps a
// 64-bit address pair per row in VGPRs -- sixteen of them in the 16384-point FIR kernel, which spilled.  Reads past
// the byte count return zero and stores past it are dropped: the stream's end needs no guard.
typedef unsigned bv2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, size_t bytes) {
    const unsigned nrec = bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : static_cast<unsigned>(bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, static_cast<int>(nrec), 0x00020000);
}
template <class In>
struct BufRows;
template <>
struct BufRows<const float2*> {
    static constexpr unsigned E = 8;
    __amdgpu_buffer_rsrc_t r;
    __device__ __forceinline__ BufRows(const float2* in, size_t first, size_t count) : r(make_rsrc(in + first, count * E)) {}
    static __device__ __forceinline__ float2 get_from(__amdgpu_buffer_rsrc_t rs, unsigned lane_bytes, unsigned row_bytes) {
        const bv2u x = __builtin_amdgcn_raw_buffer_load_b64(rs, lane_bytes, row_bytes, 0);
        return make_float2(__uint_as_float(x.x), __uint_as_float(x.y));
    }
    __device__ __forceinline__ float2 get(unsigned lane_bytes, unsigned row_bytes) const { return get_from(r, lane_bytes, row_bytes); }
};
template <>
struct BufRows<InI16> {
    static constexpr unsigned E = 4;
    __amdgpu_buffer_rsrc_t r;
    float scale;
    __device__ __forceinline__ BufRows(InI16 in, size_t first, size_t count) : r(make_rsrc(in.p + first, count * E)), scale(in.scale) {}
    __device__ __forceinline__ float2 get(unsigned lane_bytes, unsigned row_bytes) const {
        const unsigned x = __builtin_amdgcn_raw_buffer_load_b32(r, lane_bytes, row_bytes, 0);
        return make_float2(static_cast<float>(static_cast<short>(x & 0xffffu)) * scale,
                           static_cast<float>(static_cast<short>(x >> 16)) * scale);
    }
};
template <>
struct BufRows<InU8> {
    static constexpr unsigned E = 2;
    __amdgpu_buffer_rsrc_t r;
    unsigned limit;  // bytes; past the end the SAMPLE is zero (the byte 0 would convert to -1)
    __device__ __forceinline__ BufRows(InU8 in, size_t first, size_t count)
        : r(make_rsrc(in.p + first, count * E)), limit(count * E > 0xFFFFFFFFull ? 0xFFFFFFFFu : static_cast<unsigned>(count * E)) {}
    __device__ __forceinline__ float2 get(unsigned lane_bytes, unsigned row_bytes) const {
        const unsigned x = __builtin_amdgcn_raw_buffer_load_b16(r, lane_bytes, row_bytes, 0);
        const float2 v = make_float2(InU8::cvt(static_cast<float>(x & 0xffu)), InU8::cvt(static_cast<float>((x >> 8) & 0xffu)));
        return lane_bytes + row_bytes < limit ? v : make_float2(0.f, 0.f);
    }
};

// `(scale * x) as i16` (examples/single_thread_bpsk.rs:40-44): Rust's float -> int `as` truncates toward zero,
// saturates, and maps NaN to 0 -- what IQOutput then writes (src/io/raw_iq.rs:173-178).  Shared by the
// stand-alone conversion kernel (iqformat.hip) and the pulse shaper's i16 store stage.
__device__ __forceinline__ short rust_as_i16(float v) {
    float c = __builtin_fminf(__builtin_fmaxf(v, -32768.0f), 32767.0f);  // saturate (selects, no branches)
    c = v != v ? 0.0f : c;                                                // NaN -> 0
    return static_cast<short>(static_cast<int>(c));                       // truncation toward zero
}
// 16 bytes to an address that is only 4-byte aligned (an i16 IQ stream at any sample offset): one store instruction
// (the compiler splits such a store into 12 + 4 bytes)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_b128_dword_aligned(void* p, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ short2 c32_as_i16(float2 v, float scale) {
    return make_short2(rust_as_i16(scale * v.x), rust_as_i16(scale * v.y));
}

// Folded into every FIR kernel (workgroup 0): new_hist = last HL samples of
// concat(old_hist[HL], in[n]) -- the reference's `state` after the batch.
template <class In = const float2*>
__device__ __forceinline__ void hist_advance(const float2* __restrict__ old_hist, In in, size_t n,
                                             float2* __restrict__ new_hist, int HL) {
    if (blockIdx.x != 0) return;
    for (int j = threadIdx.x; j < HL; j += blockDim.x) {
        const size_t p = n + static_cast<size_t>(j);
        new_hist[j] = p < static_cast<size_t>(HL) ? old_hist[p] : in[p - HL];
    }
}

// Sample g of the logical stream [history | input | zeros]
template <class In = const float2*>
__device__ __forceinline__ float2 stream_at(In in, const float2* __restrict__ hist, int hist_len,
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

// atan2 for the fused FM demod: |error| <= ~2e-7 rad (the reference's f32 atan2 is good to 1 ulp,
// 2.4e-7 near pi), about a third of the library routine's instructions.  Minimax fit of
// atan(t)/t in t^2 on [0, 1] (degree 8), octant folding on max/min, signed zeros as atan2.
__device__ __forceinline__ float fast_atan2f(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float t = mx > 0.f ? mn * __builtin_amdgcn_rcpf(mx) : 0.f;
    const float z = t * t;
    float p = 0.0024567078799009323f;
    p = __builtin_fmaf(p, z, -0.014401284977793694f);
    p = __builtin_fmaf(p, z, 0.03978108987212181f);
    p = __builtin_fmaf(p, z, -0.0723484456539154f);
    p = __builtin_fmaf(p, z, 0.10498938709497452f);
    p = __builtin_fmaf(p, z, -0.14161226153373718f);
    p = __builtin_fmaf(p, z, 0.19985906779766083f);
    p = __builtin_fmaf(p, z, -0.33332598209381104f);
    p = __builtin_fmaf(p, z, 0.9999998807907104f);
    float r = p * t;
    if (ay > ax) r = 1.57079637f - r;
    if (__builtin_signbit(x)) r = 3.14159274f - r;
    return __builtin_copysignf(r, y);
}
// FM::demod step (src/modulation/analog.rs:27-28) with the fast atan2
__device__ __forceinline__ float fm_step_fast(float2 x, float2 p) {
    const float pcr = p.x, pci = -p.y;
    return fast_atan2f(x.x * pci + x.y * pcr, x.x * pcr - x.y * pci);
}
// lane l takes lane l - 1's value, lane 0 keeps `first` (DPP wave_shr:1)
__device__ __forceinline__ float wave_shr1(float v, float first) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, first), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}

}  // namespace comms

struct comms_fir : comms::Handle {
    int n_eff = 0;       // taps that take part: min(n_taps, n_state)
    bool real_taps = false;
    int algo = COMMS_FIR_AUTO;
    bool os1024_fixed = false;  // COMMS_FIR_OS1024_FIXED: never the ticketed kernel
    int in_fmt = 0;             // COMMS_IQ_C32 / _I16 / _U8: what d_in of the run entries points to
    float in_scale = 1.0f;      // i16 only
    comms::Scratch conv;        // converted copy, for the kernels that do not read wire formats themselves
    float* d_qt = nullptr;      // decimating chain kernel, four outputs per lane: tap quadruples for rate qt_rate
    int qt_rate = 0;
    float2* d_any_taps = nullptr;  // any-rate chain kernel: taps zero-padded to 32 * any_nt
    int any_nt = 0;
    float2* d_p8 = nullptr;     // polyphase frequency-domain chain kernel (fir_poly8.hip): branch spectra + twiddle tables ...
    bool p8_pre = false;        //   ... built for this mixer order and (mixer first: folded into the taps) increment
    uint64_t p8_frac = 0;
    bool no_poly8 = false;      // COMMS_CHAIN_TIME_DOMAIN: the time-domain kernel on every call
    bool last_poly8 = false;    // the last decimating chain launch of this handle ran fir_poly8_kernel (comms_chain_is_fused)
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
    unsigned* err_host = nullptr;  // sticky error word of the 16384-point kernel (pinned host memory) ...
    unsigned* d_err = nullptr;     //   ... as the device addresses it
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

namespace comms {
// A launch of the 16384-point kernel whose LDS waits ran out has raised the handle's error word (fir_os16k_kernel):
// its outputs are wrong, and so is everything the handle would compute from the state it left.  Checked on entry to
// every run / state entry of the handle (all of them, round 5) and, by the host-pointer entries, once more AFTER the
// call's own launches have been waited for: the synchronous caller of the failing call gets COMMS_ERR_DEVICE, not the
// invalid samples with COMMS_OK.
inline comms_status_t fir_check_sticky(const comms_fir* h) {
    if (h->err_host && __atomic_load_n(h->err_host, __ATOMIC_RELAXED) != 0)
        return fail(COMMS_ERR_DEVICE, "fir_os16k_kernel: a workgroup's LDS wait ran out (code %u): the outputs of that call "
                                      "are invalid and the handle is unusable", *h->err_host);
    return COMMS_OK;
}
}  // namespace comms
