// iqformat.hip -- raw IQ wire formats <-> Complex<f32>, the step either side of the hot path
// (SURVEY.md section 8f rank 2).  HBM-bound conversions, 16 B per lane where alignment allows.
//
//   i16 pairs (re, im, host byte order) -- src/io/raw_iq.rs:16,50-51,173-178 (IQInput / IQOutput);
//       to f32 it is cast_complex::<i16,f32> (src/util/math.rs:20-28) times a user scale,
//       back it is `(scale * x) as i16` (examples/single_thread_bpsk.rs:40-44: 8192.0 * x as i16;
//       Rust `as`: truncate toward zero, saturate, NaN -> 0);
//   u8 pairs from an RTL-SDR -- examples/fm_radio.rs:82-90: (x as f32 - 127.5) / 127.5.
#include "common.hpp"
#include "fir_handle.hpp"

namespace comms {

__global__ __launch_bounds__(256) void i16_to_c32_kernel(const short2* __restrict__ in, float2* __restrict__ out,
                                                         size_t n, float scale) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const short2 v = in[i];
        out[i] = make_float2(static_cast<float>(v.x) * scale, static_cast<float>(v.y) * scale);
    }
}

__global__ __launch_bounds__(256) void c32_to_i16_kernel(const float2* __restrict__ in, short2* __restrict__ out,
                                                         size_t n, float scale) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float2 v = in[i];
        out[i] = make_short2(rust_as_i16(scale * v.x), rust_as_i16(scale * v.y));
    }
}

__global__ __launch_bounds__(256) void u8_to_c32_kernel(const uchar2* __restrict__ in, float2* __restrict__ out,
                                                        size_t n) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uchar2 v = in[i];
        out[i] = make_float2((static_cast<float>(v.x) - 127.5f) / 127.5f, (static_cast<float>(v.y) - 127.5f) / 127.5f);
    }
}

// The casts examples/fm_radio.rs wraps around its second filter (Convert2Node :93-117: x -> Complex(x, 0);
// Convert3Node :119-141: x -> x.re), so that its whole chain can stay device-resident: two samples per lane.
__global__ __launch_bounds__(256) void real_to_c32_kernel(const float* __restrict__ in, float2* __restrict__ out, size_t n) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    const size_t pairs = n / 2;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < pairs; i += stride) {
        const float2 v = reinterpret_cast<const float2*>(in)[i];
        reinterpret_cast<float4*>(out)[i] = make_float4(v.x, 0.f, v.y, 0.f);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) out[n - 1] = make_float2(in[n - 1], 0.f);
}
__global__ __launch_bounds__(256) void c32_re_kernel(const float2* __restrict__ in, float* __restrict__ out, size_t n) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    const size_t pairs = n / 2;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < pairs; i += stride) {
        const float4 v = reinterpret_cast<const float4*>(in)[i];
        reinterpret_cast<float2*>(out)[i] = make_float2(v.x, v.z);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) out[n - 1] = in[n - 1].x;
}

static unsigned conv_grid(size_t n) {
    size_t b = (n + 255) / 256;
    if (b > 8u * kNumCU) b = 8u * kNumCU;
    return static_cast<unsigned>(b ? b : 1);
}

// run(d_in, d_out, samples, stream): the device form on the borrowed handle's stream; in_e / out_e bytes per sample (long
// captures go through the chunked host pipeline, common.hpp)
template <typename F>
static comms_status_t via_device(const void* in, size_t n, size_t in_e, void* out, size_t out_e, int32_t device, F run) {
    COMMS_TRY(use_device(device));
    Handle* h = nullptr;
    COMMS_TRY(thread_handle(device, &h));
    return h->run_host_units(in, n * in_e, in_e, out, n * out_e, out_e,
                             [&](void* d_in, void* d_out, size_t ib, size_t) { return run(d_in, d_out, ib / in_e, h->stream); });
}

}  // namespace comms

using namespace comms;

extern "C" {

comms_status_t comms_iq_i16_to_c32_dev(const int16_t* d_in, size_t n, float scale, comms_c32* d_out, int32_t device,
                                       void* stream) {
    COMMS_ARG((d_in && d_out) || !n, "NULL device pointer");
    COMMS_ARG((reinterpret_cast<uintptr_t>(d_in) & 3) == 0 && (reinterpret_cast<uintptr_t>(d_out) & 7) == 0,
              "pointers must be aligned to one IQ sample");
    COMMS_TRY(use_device(device));
    if (!n) return COMMS_OK;
    i16_to_c32_kernel<<<dim3(conv_grid(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
        reinterpret_cast<const short2*>(d_in), reinterpret_cast<float2*>(d_out), n, scale);
    return launch_ok("i16_to_c32_kernel");
}
comms_status_t comms_iq_c32_to_i16_dev(const comms_c32* d_in, size_t n, float scale, int16_t* d_out, int32_t device,
                                       void* stream) {
    COMMS_ARG((d_in && d_out) || !n, "NULL device pointer");
    COMMS_ARG((reinterpret_cast<uintptr_t>(d_in) & 7) == 0 && (reinterpret_cast<uintptr_t>(d_out) & 3) == 0,
              "pointers must be aligned to one IQ sample");
    COMMS_TRY(use_device(device));
    if (!n) return COMMS_OK;
    c32_to_i16_kernel<<<dim3(conv_grid(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
        reinterpret_cast<const float2*>(d_in), reinterpret_cast<short2*>(d_out), n, scale);
    return launch_ok("c32_to_i16_kernel");
}
comms_status_t comms_iq_u8_to_c32_dev(const uint8_t* d_in, size_t n, comms_c32* d_out, int32_t device, void* stream) {
    COMMS_ARG((d_in && d_out) || !n, "NULL device pointer");
    COMMS_ARG((reinterpret_cast<uintptr_t>(d_in) & 1) == 0 && (reinterpret_cast<uintptr_t>(d_out) & 7) == 0,
              "pointers must be aligned to one IQ sample");
    COMMS_TRY(use_device(device));
    if (!n) return COMMS_OK;
    u8_to_c32_kernel<<<dim3(conv_grid(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
        reinterpret_cast<const uchar2*>(d_in), reinterpret_cast<float2*>(d_out), n);
    return launch_ok("u8_to_c32_kernel");
}

comms_status_t comms_iq_real_to_c32_dev(const float* d_in, size_t n, comms_c32* d_out, int32_t device, void* stream) {
    COMMS_ARG((d_in && d_out) || !n, "NULL device pointer");
    COMMS_ARG((reinterpret_cast<uintptr_t>(d_in) & 7) == 0 && (reinterpret_cast<uintptr_t>(d_out) & 15) == 0,
              "input must be 8-byte aligned, output 16-byte aligned");
    COMMS_ARG(!ranges_overlap(d_in, n * 4, d_out, n * 8), "the cast cannot run in place");
    COMMS_TRY(use_device(device));
    if (!n) return COMMS_OK;
    real_to_c32_kernel<<<dim3(conv_grid((n + 1) / 2)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
        d_in, reinterpret_cast<float2*>(d_out), n);
    return launch_ok("real_to_c32_kernel");
}
comms_status_t comms_iq_c32_re_dev(const comms_c32* d_in, size_t n, float* d_out, int32_t device, void* stream) {
    COMMS_ARG((d_in && d_out) || !n, "NULL device pointer");
    COMMS_ARG((reinterpret_cast<uintptr_t>(d_in) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_out) & 7) == 0,
              "input must be 16-byte aligned, output 8-byte aligned");
    COMMS_ARG(!ranges_overlap(d_in, n * 8, d_out, n * 4), "the cast cannot run in place");
    COMMS_TRY(use_device(device));
    if (!n) return COMMS_OK;
    c32_re_kernel<<<dim3(conv_grid((n + 1) / 2)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
        reinterpret_cast<const float2*>(d_in), d_out, n);
    return launch_ok("c32_re_kernel");
}

comms_status_t comms_iq_i16_to_c32(const int16_t* in, size_t n, float scale, comms_c32* out, int32_t device) {
    COMMS_ARG((in && out) || !n, "NULL host pointer");
    if (!n) return use_device(device);
    return via_device(in, n, 4, out, 8, device, [&](void* a, void* b, size_t m, void* st) {
        return comms_iq_i16_to_c32_dev(static_cast<const int16_t*>(a), m, scale, static_cast<comms_c32*>(b), device, st);
    });
}
comms_status_t comms_iq_c32_to_i16(const comms_c32* in, size_t n, float scale, int16_t* out, int32_t device) {
    COMMS_ARG((in && out) || !n, "NULL host pointer");
    if (!n) return use_device(device);
    return via_device(in, n, 8, out, 4, device, [&](void* a, void* b, size_t m, void* st) {
        return comms_iq_c32_to_i16_dev(static_cast<const comms_c32*>(a), m, scale, static_cast<int16_t*>(b), device, st);
    });
}
comms_status_t comms_iq_u8_to_c32(const uint8_t* in, size_t n, comms_c32* out, int32_t device) {
    COMMS_ARG((in && out) || !n, "NULL host pointer");
    if (!n) return use_device(device);
    return via_device(in, n, 2, out, 8, device, [&](void* a, void* b, size_t m, void* st) {
        return comms_iq_u8_to_c32_dev(static_cast<const uint8_t*>(a), m, static_cast<comms_c32*>(b), device, st);
    });
}

}  // extern "C"
