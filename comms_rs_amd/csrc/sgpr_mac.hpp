// sgpr_mac.hpp -- packed complex MACs whose tap comes from an SGPR pair (kernel-argument taps):
// shared by the decimating chain kernel (fir_decim.hip) and the polyphase pulse shaper (fir.hip).
#pragma once

#include "fft_radix.hpp"

namespace comms {

typedef float v2f __attribute__((ext_vector_type(2)));
// acc += t * u for a real tap held in the lo / hi half of an SGPR pair
__device__ __forceinline__ void mac_s_lo(cf& acc, cf u, v2f tp) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(u), "s"(tp));
}
__device__ __forceinline__ void mac_s_hi(cf& acc, cf u, v2f tp) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(u), "s"(tp));
}
// acc += (i * t) * u : acc.re -= t*u.im, acc.im += t*u.re
__device__ __forceinline__ void mac_si_lo(cf& acc, cf u, v2f tp) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0]" : "+v"(acc) : "v"(u), "s"(tp));
}
__device__ __forceinline__ void mac_si_hi(cf& acc, cf u, v2f tp) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "+v"(acc) : "v"(u), "s"(tp));
}

template <bool REAL>
__device__ __forceinline__ void mac_tap(cf& acc, cf u, const v2f& pre, const v2f& pim, bool hi) {
    if (hi) {
        mac_s_hi(acc, u, pre);
        if (!REAL) mac_si_hi(acc, u, pim);
    } else {
        mac_s_lo(acc, u, pre);
        if (!REAL) mac_si_lo(acc, u, pim);
    }
}

}  // namespace comms
