// fft_radix.hpp -- in-register radix-4 / radix-16 DFT butterflies on float2.
//
// One lane holds 16 complex points in 32 VGPRs; a 16-point DFT is two layers
// of radix-4 butterflies with the nine non-trivial W16 twiddles as compile-time
// constants.  DIR = -1: forward (e^{-2 pi i jk/N}), +1: inverse (unnormalised),
// matching rustfft's FFTplanner::new(inverse) convention used by the reference
// (src/fft/fft_node.rs:66).  FMAs are explicit (library builds with
// -ffp-contract=off).
#pragma once

#include <hip/hip_runtime.h>

namespace comms {

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// a * b with explicit FMAs (2 mul + 2 fma)
__device__ __forceinline__ float2 cmulf(float2 a, float2 b) {
    return make_float2(__builtin_fmaf(-a.y, b.y, a.x * b.x), __builtin_fmaf(a.y, b.x, a.x * b.y));
}
// a * conj(b)
__device__ __forceinline__ float2 cmulcf(float2 a, float2 b) {
    return make_float2(__builtin_fmaf(a.y, b.y, a.x * b.x), __builtin_fmaf(a.y, b.x, -(a.x * b.y)));
}
// multiply by (DIR * i):  forward (DIR=-1): -i*a = (a.y, -a.x); inverse: i*a = (-a.y, a.x)
template <int DIR>
__device__ __forceinline__ float2 mul_dir_i(float2 a) {
    return DIR < 0 ? make_float2(a.y, -a.x) : make_float2(-a.y, a.x);
}

// 4-point DFT in place: (a,b,c,d) = x[0..3] -> X[0..3]
template <int DIR>
__device__ __forceinline__ void radix4(float2& a, float2& b, float2& c, float2& d) {
    float2 t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), t3 = mul_dir_i<DIR>(csub(b, d));
    a = cadd(t0, t2);
    c = csub(t0, t2);
    b = cadd(t1, t3);
    d = csub(t1, t3);
}

// W16^m = cos(2 pi m/16) + DIR * i * sin(2 pi m/16)
template <int DIR, int M>
__device__ __forceinline__ float2 mul_w16(float2 a) {
    constexpr float C1 = 0.92387953251128675613f;  // cos(pi/8)
    constexpr float S1 = 0.38268343236508977173f;  // sin(pi/8)
    constexpr float R2 = 0.70710678118654752440f;  // sqrt(1/2)
    constexpr float sg = DIR < 0 ? -1.0f : 1.0f;
    if constexpr (M == 0) {
        return a;
    } else if constexpr (M == 4) {
        return mul_dir_i<DIR>(a);
    } else if constexpr (M == 2) {  // R2 * (1 + sg*i)
        return make_float2(R2 * (a.x - sg * a.y), R2 * (a.y + sg * a.x));
    } else if constexpr (M == 6) {  // R2 * (-1 + sg*i)
        return make_float2(-R2 * (a.x + sg * a.y), R2 * (sg * a.x - a.y));
    } else {
        constexpr float wr = (M == 1) ? C1 : (M == 3) ? S1 : (M == 9) ? -C1 : 0.0f;
        constexpr float wi = sg * ((M == 1) ? S1 : (M == 3) ? C1 : (M == 9) ? -S1 : 0.0f);
        static_assert(M == 1 || M == 3 || M == 9, "unsupported W16 power");
        return make_float2(__builtin_fmaf(-a.y, wi, a.x * wr), __builtin_fmaf(a.y, wr, a.x * wi));
    }
}

// 16-point DFT in place on v[0..15].  Input natural order; output X[k] lands in
// v[R16_POS(k)] with R16_POS(k) = 4*(k&3) + (k>>2).
#define R16_POS(k) (4 * ((k)&3) + ((k) >> 2))

template <int DIR>
__device__ __forceinline__ void radix16(float2 (&v)[16]) {
    // layer A: for each n0, DFT4 over n1 on {n0, n0+4, n0+8, n0+12} -> A[n0][k1] at v[n0+4*k1]
    radix4<DIR>(v[0], v[4], v[8], v[12]);
    radix4<DIR>(v[1], v[5], v[9], v[13]);
    radix4<DIR>(v[2], v[6], v[10], v[14]);
    radix4<DIR>(v[3], v[7], v[11], v[15]);
    // twiddle W16^{n0*k1}
    v[5] = mul_w16<DIR, 1>(v[5]);
    v[6] = mul_w16<DIR, 2>(v[6]);
    v[7] = mul_w16<DIR, 3>(v[7]);
    v[9] = mul_w16<DIR, 2>(v[9]);
    v[10] = mul_w16<DIR, 4>(v[10]);
    v[11] = mul_w16<DIR, 6>(v[11]);
    v[13] = mul_w16<DIR, 3>(v[13]);
    v[14] = mul_w16<DIR, 6>(v[14]);
    v[15] = mul_w16<DIR, 9>(v[15]);
    // layer B: for each k1, DFT4 over n0 on {4*k1 .. 4*k1+3} -> X[k1 + 4*k0] at v[4*k1 + k0]
    radix4<DIR>(v[0], v[1], v[2], v[3]);
    radix4<DIR>(v[4], v[5], v[6], v[7]);
    radix4<DIR>(v[8], v[9], v[10], v[11]);
    radix4<DIR>(v[12], v[13], v[14], v[15]);
}

}  // namespace comms
