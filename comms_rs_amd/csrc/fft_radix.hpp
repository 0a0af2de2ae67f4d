// fft_radix.hpp -- in-register radix-4 / radix-16 DFT butterflies on packed complex f32.
//
// A complex number is one 64-bit VGPR pair {re, im} (`cf`).  Every operation is a
// single VOP3P packed-f32 instruction (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32)
// whose op_sel / neg modifiers do the real/imag swizzles and sign flips in the
// operand read -- no marshaling moves:
//     a + b, a - b, a -+ i*b  : 1 instruction        a*b, a*conj(b) : 2 instructions
// Why by hand: on gfx950 a scalar wave64 f32 VALU op and a packed one both issue in
// ~4.4 cycles per SIMD (scripts/probe_valu.py), so packing doubles the FFT's VALU
// rate -- but hipcc's SLP vectoriser spends ~0.5 v_mov per packed op on the swizzles
// (measured slower than scalar code); the modifiers make them free.
//
// One lane holds 16 points in 32 VGPRs; a 16-point DFT is two layers of radix-4
// butterflies, 80 packed instructions.  DIR = -1: forward (e^{-2 pi i jk/N}), +1:
// inverse (unnormalised), matching rustfft's FFTplanner::new(inverse) as used by the
// reference (src/fft/fft_node.rs:66).  FMAs are explicit (library builds with
// -ffp-contract=off).
#pragma once

#include <hip/hip_runtime.h>

namespace comms {

typedef float cf __attribute__((ext_vector_type(2)));  // {re, im} in one aligned VGPR pair

__device__ __forceinline__ cf to_cf(float2 f) { return cf{f.x, f.y}; }
__device__ __forceinline__ float2 to_f2(cf c) { return make_float2(c.x, c.y); }

#define COMMS_PK2(name, text)                                          \
    __device__ __forceinline__ cf name(cf a, cf b) {                   \
        cf d;                                                          \
        asm(text : "=v"(d) : "v"(a), "v"(b));                          \
        return d;                                                      \
    }
// a + b
COMMS_PK2(cadd, "v_pk_add_f32 %0, %1, %2")
// a - b
COMMS_PK2(csub, "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]")
// a + (-i)*b = (a.re + b.im, a.im - b.re)
COMMS_PK2(cadd_mi, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")
// a + (+i)*b = (a.re - b.im, a.im + b.re)
COMMS_PK2(cadd_pi, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")
// i*(a - b) = (b.im - a.im, a.re - b.re)
COMMS_PK2(crot_sub, "v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[1,0] neg_hi:[0,1]")
// (a.re*b.re, a.re*b.im)  and  (a.re*b.re, -a.re*b.im): first halves of a*b and a*conj(b)
COMMS_PK2(cmul_lo, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]")
COMMS_PK2(cmulc_lo, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[1,0]")
#undef COMMS_PK2

// a * b = (ar*br - ai*bi, ar*bi + ai*br)
__device__ __forceinline__ cf cmulf(cf a, cf b) {
    cf p = cmul_lo(a, b), d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
        : "=v"(d)
        : "v"(a), "v"(b), "v"(p));
    return d;
}
// a * conj(b) = (ar*br + ai*bi, ai*br - ar*bi)
__device__ __forceinline__ cf cmulcf(cf a, cf b) {
    cf p = cmulc_lo(a, b), d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(p));
    return d;
}
// the same with the second factor in an SGPR pair (a wave-uniform twiddle: no VGPRs, no LDS read)
__device__ __forceinline__ cf cmulf_s(cf a, cf b) {
    cf p, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(p) : "v"(a), "s"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(d) : "v"(a), "s"(b), "v"(p));
    return d;
}
__device__ __forceinline__ cf cmulcf_s(cf a, cf b) {
    cf p, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[1,0]" : "=v"(p) : "v"(a), "s"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "s"(b), "v"(p));
    return d;
}
// a * w for DIR = -1, a * conj(w) for DIR = +1 (w = forward twiddle)
template <int DIR>
__device__ __forceinline__ cf tw_mul(cf a, cf w) {
    return DIR < 0 ? cmulf(a, w) : cmulcf(a, w);
}
// the same with a wave-uniform (compile-time) twiddle read from an SGPR pair: the W16 constants of radix16 below.
// With a "v" operand the compiler copies the constant into a VGPR pair in front of every use (v_mov_b64: five
// vector instructions per 16-point DFT, and the registers).
template <int DIR>
__device__ __forceinline__ cf tw_mul_s(cf a, cf w) {
    return DIR < 0 ? cmulf_s(a, w) : cmulcf_s(a, w);
}
// a + (DIR*i)*b  and  a - (DIR*i)*b
template <int DIR>
__device__ __forceinline__ cf cadd_di(cf a, cf b) {
    return DIR < 0 ? cadd_mi(a, b) : cadd_pi(a, b);
}
template <int DIR>
__device__ __forceinline__ cf csub_di(cf a, cf b) {
    return DIR < 0 ? cadd_pi(a, b) : cadd_mi(a, b);
}

// 4-point DFT in place: (a,b,c,d) = x[0..3] -> X[0..3]
template <int DIR>
__device__ __forceinline__ void radix4(cf& a, cf& b, cf& c, cf& d) {
    const cf t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), t3 = csub(b, d);
    a = cadd(t0, t2);
    c = csub(t0, t2);
    b = cadd_di<DIR>(t1, t3);
    d = csub_di<DIR>(t1, t3);
}
// Same, with the input c standing for (DIR*i)*c (the W16^4 twiddle folded into the adds)
template <int DIR>
__device__ __forceinline__ void radix4_c_rot(cf& a, cf& b, cf& c, cf& d) {
    const cf t0 = cadd_di<DIR>(a, c), t1 = csub_di<DIR>(a, c), t2 = cadd(b, d), t3 = csub(b, d);
    a = cadd(t0, t2);
    c = csub(t0, t2);
    b = cadd_di<DIR>(t1, t3);
    d = csub_di<DIR>(t1, t3);
}

// v_permlane32_swap / v_permlane16_swap on a complex register pair: (a, b) -> a keeps its lanes 0-31 and takes
// b's lanes 0-31 into 32-63, b takes a's lanes 32-63 into 0-31 and keeps its own 32-63 (32); the same with the
// four 16-lane rows, odd rows of a <-> even rows of b (16).  The builtins (not raw asm) so that the compiler
// knows the instruction; the packed butterflies next to them are plain vector adds for the same reason -- the
// hazard recogniser cannot see into an asm block -- and an asm operand produced right before a swap is fenced.
__device__ __forceinline__ void lane_swap_fence(cf& a, cf& b) { asm volatile("s_nop 1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void lane_swap32(cf& a, cf& b) {
    lane_swap_fence(a, b);
    const auto x = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
    const auto y = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
    a = cf{__uint_as_float(x[0]), __uint_as_float(y[0])};
    b = cf{__uint_as_float(x[1]), __uint_as_float(y[1])};
}
__device__ __forceinline__ void lane_swap16(cf& a, cf& b) {
    lane_swap_fence(a, b);
    const auto x = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
    const auto y = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
    a = cf{__uint_as_float(x[0]), __uint_as_float(y[0])};
    b = cf{__uint_as_float(x[1]), __uint_as_float(y[1])};
}

// 4-point DFTs ACROSS lanes: lane (q0, c) = (lane & 15, lane >> 4) holds r[k1], k1 = 0 ... 15, and the transform runs over
// c.  v_permlane32_swap pairs registers so that the two halves of the wave (c's upper bit) meet in one lane,
// v_permlane16_swap does the same for the 16-lane rows (c's lower bit); the W4 twiddle of the odd outputs is folded
// into the second butterfly.  Result: z[4 t + m] of lane q0 + 16 g + 32 h is output t of the transform whose other
// indices are (q0, k1 = 2 m + 8 g + h).  No LDS; 32 swaps + 32 packed adds.
template <int DIR>
__device__ __forceinline__ void radix4_lanes(const cf (&r)[16], cf (&z)[16]) {
    cf sd[16];  // [0..7] sums, [8..15] differences of the upper-bit halves
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        cf p = r[2 * m], q = r[2 * m + 1];
        lane_swap32(p, q);
        sd[m] = p + q;
        sd[8 + m] = p - q;
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        cf a = sd[m], b = sd[m + 4];
        lane_swap16(a, b);
        z[m] = a + b;      // t = 0
        z[8 + m] = a - b;  // t = 2
        cf c = sd[8 + m], d = sd[12 + m];
        lane_swap16(c, d);
        z[4 + m] = cadd_di<DIR>(c, d);   // t = 1
        z[12 + m] = csub_di<DIR>(c, d);  // t = 3
    }
}

// forward W16^m = (cos(2 pi m/16), -sin(2 pi m/16)); the inverse uses the conjugate
template <int M>
__device__ __forceinline__ cf w16() {
    constexpr float C1 = 0.92387953251128675613f;  // cos(pi/8)
    constexpr float S1 = 0.38268343236508977173f;  // sin(pi/8)
    constexpr float R2 = 0.70710678118654752440f;  // sqrt(1/2)
    static_assert(M == 1 || M == 2 || M == 3 || M == 6 || M == 9, "unsupported W16 power");
    if constexpr (M == 1) return cf{C1, -S1};
    if constexpr (M == 2) return cf{R2, -R2};
    if constexpr (M == 3) return cf{S1, -C1};
    if constexpr (M == 6) return cf{-R2, -R2};
    return cf{-C1, S1};  // M == 9
}

// 16-point DFT in place on v[0..15].  Input natural order; output X[k] lands in
// v[R16_POS(k)] with R16_POS(k) = 4*(k&3) + (k>>2).
#define R16_POS(k) (4 * ((k)&3) + ((k) >> 2))

template <int DIR>
__device__ __forceinline__ void radix16(cf (&v)[16]) {
    // layer A: for each n0, DFT4 over n1 on {n0, n0+4, n0+8, n0+12} -> A[n0][k1] at v[n0+4*k1]
    radix4<DIR>(v[0], v[4], v[8], v[12]);
    radix4<DIR>(v[1], v[5], v[9], v[13]);
    radix4<DIR>(v[2], v[6], v[10], v[14]);
    radix4<DIR>(v[3], v[7], v[11], v[15]);
    // twiddle W16^{n0*k1}; v[10] (W16^4 = -+i) is folded into its layer-B butterfly
    v[5] = tw_mul_s<DIR>(v[5], w16<1>());
    v[6] = tw_mul_s<DIR>(v[6], w16<2>());
    v[7] = tw_mul_s<DIR>(v[7], w16<3>());
    v[9] = tw_mul_s<DIR>(v[9], w16<2>());
    v[11] = tw_mul_s<DIR>(v[11], w16<6>());
    v[13] = tw_mul_s<DIR>(v[13], w16<3>());
    v[14] = tw_mul_s<DIR>(v[14], w16<6>());
    v[15] = tw_mul_s<DIR>(v[15], w16<9>());
    // layer B: for each k1, DFT4 over n0 on {4*k1 .. 4*k1+3} -> X[k1 + 4*k0] at v[4*k1 + k0]
    radix4<DIR>(v[0], v[1], v[2], v[3]);
    radix4<DIR>(v[4], v[5], v[6], v[7]);
    radix4_c_rot<DIR>(v[8], v[9], v[10], v[11]);
    radix4<DIR>(v[12], v[13], v[14], v[15]);
}

}  // namespace comms
