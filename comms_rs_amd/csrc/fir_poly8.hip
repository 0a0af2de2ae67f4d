// fir_poly8.hip -- FIR -> mixer -> keep every 8th sample (the BASELINE metric's chain as one launch; the mixer-first
// order through modulated taps), computed as EIGHT POLYPHASE BRANCHES IN THE FREQUENCY DOMAIN.
//
// Same results as BatchFirNode (src/filter/fir.rs:87-102), MixerNode (src/mixer.rs:73-85) and DecimateNode
// (src/util/resample_node.rs:53-65) in series, within the FIR tolerance.  Why another form: the time-domain chain
// kernel (fir_decim.hip) spends taps/8 packed FMAs per input sample -- at 255 taps it is bound by vector issue, not by
// HBM (NOTES.md round 5: ~30 us of issue in a 40-us launch) -- and the overlap-save fusion (fir.hip, MODE != 0)
// computes all eight outputs of which seven are dropped.  With
//     y[8j] = sum_c sum_m h[8m - c] x[8(j - m) + c]          (c = 0 ... 7: the sample's position within its group of 8)
// the kept outputs are the sum of eight 33-tap filters g_c[m] = h[8m - c], each running at the LOW rate on the phase
// stream v_c[i] = x[8i + c].  Per segment of 1024 input samples (128 per phase, 32 of them halo: 768 new samples,
// 96 outputs) a wave does eight 128-point forward transforms (as ONE 16-point DFT per lane, a twiddle, one exchange
// and two 8-point DFTs per lane), multiplies by the branch spectra G_c and sums over c, and ONE 128-point inverse
// transform: ~300 vector instructions per segment against ~500 for the same 768 samples in the time domain, and a
// third of the overlap-save kernel's.
//
// Lane maps (l = lane):
//   load      row a = 0..15: sample base + 64a + l -> phase c = l & 7, position q = 8a + (l >> 3) within the phase
//   stage 1   DFT16 over a in registers, x W128^{d k1} (d = l >> 3), exchange 1 [k1][l]
//   stage 2   lane (k1 = l & 15, cg = l >> 4) holds phases c = cg, cg + 4, all d: two DFT8 over d -> V_c[k1 + 16 k2]
//   multiply  P[k2] = sum_ci G_{cg + 4 ci}[k1 + 16 k2] V[ci][k2]            (spectra in LDS, lane-major)
//   reduce    exchange 2 [k2][cg][k1]; lane (k1, j = l >> 4) sums the four cg of k2 = j and k2 = j + 4
//   inverse   DFT8 over k2: in-lane radix 2, then v_permlane32_swap / v_permlane16_swap butterflies over j;
//             x w128^{q1 k1}; exchange 3 [q1][k1]; DFT4 over k1's upper digit; twiddle; exchange 4; DFT4 over the
//             lower digit -> lane nu = l & 31 holds z[nu + 32 t], t = 0..3, folded into two registers of 64 positions
//             (ya = z[l], yb = z[64 + l]); the first 8 HR positions are the segment's halo
//   store     out[OUT seg + l - 8 HR] (lanes from 8 HR on) and 64 further on, behind the mixer rotor (and the FM demodulator)
// A wave's LDS operations execute in order: the four exchanges share one private buffer, no barrier after set-up.
// Segments are drawn from an LDS ticket counter per workgroup, dealt in round-robin chunks as in fir_os1024_dyn_kernel; the next
// segment's rows are requested behind the forward half, with the previous segment's stores issued in front of them.
//
// Other rates on the same forward transforms (which are most of the work and are what every rate needs):
//   rate 4       the outputs y[8j + 4] are the same phase streams through a second set of branch filters h[8m - c + 4]: a second
//                spectra table, a second multiply and a second inverse half per segment; the two phases leave interleaved;
//   rates 8 m    (16 ... 64) every m-th output of the rate-8 form is stored.
// The FM demodulator (y[j-1] from the lane below; the halo position in front of a segment's first output serves that output, so it
// costs eight taps of the limit) comes at rates 8 and 4.  Halo rows by tap count: HR = 2 / 3 / 4 (up to 129 / 193 / 257 taps: 896 /
// 832 / 768 new samples per segment), and 5 / 6 / 8 for filters of up to 321 / 385 / 513 taps (704 / 640 / 512 new samples: more
// transforms per sample, still one launch where the alternative is the overlap-save FIR and a mixer-decimator behind it).  Raw
// i16 / u8 IQ converted in the load stage, four-wave workgroups for short batches.  Lane-accurate numpy model: scripts/proto_poly8.py.
#include <hip/hip_ext.h>

#include <cmath>
#include <vector>

#include "common.hpp"
#include "fft_radix.hpp"
#include "fir_handle.hpp"

namespace comms {

constexpr int P8_S1 = 66;               // exchange 1: row stride (k1), 2 mod 32: the read side's (k1, cg) spread over all banks
constexpr int P8_BUF = 16 * P8_S1;      // per-wave exchange buffer, in cf
// HR halo rows of 64 samples (8 HR positions per phase): branch filters of up to 8 HR + 1 taps, i.e. up to 64 HR + 1 taps of
// the FIR; the FM demodulator needs one more valid output in front of the segment's first (y[j-1]): 64 HR - 7 taps.
template <int HR>
struct P8Geom {
    static constexpr int HQ = 8 * HR, HALO = 64 * HR, NEW = 1024 - 64 * HR, OUT = 128 - 8 * HR;
};
// LDS: NPH spectra tables G [16][64], forward twiddle [16][8], sin / cos table [64], WPB exchange buffers, ticket
constexpr int P8_TAB = 128 + 64;  // forward twiddle + sin / cos table; in front of them NPH spectra tables of 1024
constexpr size_t p8_lds_bytes(int wpb, int nph) { return (1024 * static_cast<size_t>(nph) + P8_TAB + static_cast<size_t>(wpb) * P8_BUF) * sizeof(float2) + 16; }

struct P8Tables {
    const cf* g;     // [16][64]  G_{cg + 4 ci}[k1 + 16 k2] / 128 at [8 ci + k2][lane], k1 = lane & 15, cg = lane >> 4; behind it
                     //           the same for the outputs y[8j + 4] (branch filters h[8m - c + 4]): the second output phase
    const cf* tw;    // [16][8]   W128^{d k1} at [k1][d]
    const cf* sc;    // [64]      (cos, sin)(2 pi i / 64)
    const cf* lane;  // [7][64]   per-lane constants of the inverse (below)
};
struct P8Mix {
    uint64_t turns0, frac;  // mixer phase of input sample 0, increment per input sample (turns of fl(2 pi) x 2^64)
    float2 step512;         // e^{i 512 dphi}: 64 outputs on
    float2 step4;           // e^{i 4 dphi}: the second output phase (rate 4)
    unsigned keep;          // rates 8 m: every m-th output is kept (1: all) ...
    unsigned div_m, div_s;  //   ... and o / m for 32-bit o: t = mulhi(div_m, o); (t + ((o - t) >> 1)) >> div_s
};

__device__ __forceinline__ void p8_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// acc + a * b
__device__ __forceinline__ cf cmacf(cf acc, cf a, cf b) {
    cf p, d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(p) : "v"(a), "v"(b), "v"(acc));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(p));
    return d;
}

// 8-point DFT, natural order in and out: two radix-4 and the W8 butterflies (28 packed instructions)
template <int DIR>
__device__ __forceinline__ void radix8(const cf (&x)[8], cf (&X)[8]) {
    cf e0 = x[0], e1 = x[2], e2 = x[4], e3 = x[6];
    cf o0 = x[1], o1 = x[3], o2 = x[5], o3 = x[7];
    radix4<DIR>(e0, e1, e2, e3);
    radix4<DIR>(o0, o1, o2, o3);
    o1 = tw_mul_s<DIR>(o1, w16<2>());  // W8^1
    o3 = tw_mul_s<DIR>(o3, w16<6>());  // W8^3
    X[0] = cadd(e0, o0);
    X[4] = csub(e0, o0);
    X[1] = cadd(e1, o1);
    X[5] = csub(e1, o1);
    X[2] = cadd_di<DIR>(e2, o2);       // W8^2 = -+i
    X[6] = csub_di<DIR>(e2, o2);
    X[3] = cadd(e3, o3);
    X[7] = csub(e3, o3);
}

struct P8Lane {
    cf w8j;   // w8^{j}, j = lane >> 4                       (w = conjugate of the forward root)
    cf tau;   // i on odd 16-lane rows, 1 on even ones
    cf u0;    // w128^{q1 k1},       q1 = 2 (row & 1) + (lane >> 5), k1 = lane & 15
    cf u1;    // w128^{(q1 + 4) k1}
    cf m1, m2, m3;  // w16^{t ka}, t = 1..3, ka = (lane & 31) >> 3
};

// One segment, first half: v[a] = samples base + 64a + lane -> the spectrum products of the lane's two phases, written to
// exchange 2.  v is dead afterwards (the caller fetches the next segment's rows into it while the second half runs).
template <int NPH>
__device__ __forceinline__ void poly8_forward(cf (&v)[16], cf* lds, const cf* tw, const cf* gsp, int l, cf (&p2)[8]) {
    const int d = l >> 3;
    radix16<-1>(v);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        cf x = v[R16_POS(k)];
        if (k) x = cmulf(x, tw[k * 8 + d]);
        lds[k * P8_S1 + l] = x;
    }
    p8_lds_sync();
    const int k1 = l & 15, cg = l >> 4;
    cf xa[8], xb[8];
#pragma unroll
    for (int dd = 0; dd < 8; ++dd) {
        xa[dd] = lds[k1 * P8_S1 + cg + 8 * dd];
        xb[dd] = lds[k1 * P8_S1 + cg + 4 + 8 * dd];
    }
    p8_lds_sync();
    cf va[8], vb[8];
    radix8<-1>(xa, va);
    radix8<-1>(xb, vb);
    // ---- spectra, summed over the lane's two phases; exchange 2 [k2][cg][k1] (row stride 80: both j of a 32-lane read group on
    // different banks)
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) {
        cf p = cmulf(va[k2], gsp[k2 * 64 + l]);
        p = cmacf(p, vb[k2], gsp[(8 + k2) * 64 + l]);
        lds[k2 * 80 + cg * 16 + k1] = p;
        if (NPH == 2) {  // the same transforms through the second phase's branch spectra: kept in registers until the first
                         // phase's inverse half has read exchange 2
            p2[k2] = cmacf(cmulf(va[k2], gsp[1024 + k2 * 64 + l]), vb[k2], gsp[1024 + (8 + k2) * 64 + l]);
        }
    }
    p8_lds_sync();
}
// the second phase's products into exchange 2 (behind the first phase's inverse half)
__device__ __forceinline__ void poly8_put(cf* lds, const cf (&p2)[8], int l) {
    const int k1 = l & 15, cg = l >> 4;
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) lds[k2 * 80 + cg * 16 + k1] = p2[k2];
    p8_lds_sync();
}

// ... second half: the sum over the phases and the 128-point inverse transform -> ya = z[lane], yb = z[64 + lane], the
// segment's 128 positions (the first 8 HR of them are its halo), before the mixer.
__device__ __forceinline__ void poly8_inverse(cf* lds, const P8Lane& lc, int l, cf& ya, cf& yb) {
    const int k1 = l & 15, cg = l >> 4;
    const int j = cg;
    cf za = lds[j * 80 + k1], zb = lds[(j + 4) * 80 + k1];
#pragma unroll
    for (int g = 1; g < 4; ++g) {
        za = za + lds[j * 80 + g * 16 + k1];
        zb = zb + lds[(j + 4) * 80 + g * 16 + k1];
    }
    p8_lds_sync();
    // ---- inverse DFT8 over k2 = j + 4 kappa: radix 2 in the lane, radix 4 over j across the lanes.  After the swaps lane
    // (k1, row parity j0, half j1) holds T[k1][q1] for q1 = 2 j0 + j1 (u) and q1 + 4 (w).
    cf a = za + zb;
    cf b = cmulf(za - zb, lc.w8j);
    lane_swap32(a, b);
    cf s = a + b;
    cf dd = cmulf(a - b, lc.tau);
    lane_swap16(s, dd);
    const cf u = cmulf(s + dd, lc.u0);
    const cf w = cmulf(s - dd, lc.u1);
    // ---- exchange 3 [q1][k1], row stride 20
    const int q1a = 2 * ((l >> 4) & 1) + (l >> 5);
    lds[q1a * 20 + k1] = u;
    lds[(q1a + 4) * 20 + k1] = w;
    p8_lds_sync();
    const int nu = l & 31;
    const int r8 = nu & 7, r4 = nu >> 3;  // exchange 3 read: (q1, ka); exchange 4 read: (q1, t)
    cf y0 = lds[r8 * 20 + r4], y1 = lds[r8 * 20 + r4 + 4], y2 = lds[r8 * 20 + r4 + 8], y3 = lds[r8 * 20 + r4 + 12];
    p8_lds_sync();
    radix4<1>(y0, y1, y2, y3);
    y1 = cmulf(y1, lc.m1);
    y2 = cmulf(y2, lc.m2);
    y3 = cmulf(y3, lc.m3);
    // ---- exchange 4 [t][ka][q1], row stride 40
    lds[0 * 40 + r4 * 8 + r8] = y0;
    lds[1 * 40 + r4 * 8 + r8] = y1;
    lds[2 * 40 + r4 * 8 + r8] = y2;
    lds[3 * 40 + r4 * 8 + r8] = y3;
    p8_lds_sync();
    cf m0 = lds[r4 * 40 + r8], m1 = lds[r4 * 40 + 8 + r8], m2 = lds[r4 * 40 + 16 + r8], m3 = lds[r4 * 40 + 24 + r8];
    p8_lds_sync();
    radix4<1>(m0, m1, m2, m3);  // z[nu + 32 t], t = 0..3
    ya = l < 32 ? m0 : m1;
    yb = l < 32 ? m2 : m3;
}

// (cos, sin) of 2 pi u / 2^32: a 64-entry table for the upper six bits, a short series for the rest (|error| ~ 1e-7)
__device__ __forceinline__ cf p8_rotor(unsigned u, const cf* sc) {
    const cf t = sc[u >> 26];
    const float th = static_cast<float>(u & 0x3FFFFFFu) * 1.4629180792671596e-09f;  // 2 pi / 2^32
    const float z = th * th;
    float sp = __builtin_fmaf(z, 8.3333333e-3f, -1.6666667e-1f);
    sp = __builtin_fmaf(z, sp, 1.0f);
    const float sn = th * sp;
    float cp = __builtin_fmaf(z, -1.3888889e-3f, 4.1666667e-2f);
    cp = __builtin_fmaf(z, cp, -0.5f);
    const float cs = __builtin_fmaf(z, cp, 1.0f);
    return cmulf(t, cf{cs, sn});
}

struct P8Fm {
    const float2* prev;  // FM.prev before this call (the last decimated sample of the batch before)
    float2* prev_new;    //   ... and after it
};

// (Nontemporal rows and outputs, which help the wave-private time-domain kernel, were measured here and dropped: level at 2^26
// samples, 6 % slower at 2^25 -- a segment's halo rows are its neighbour's rows, and plain loads leave them in L2.)
// In: Complex<f32>, or raw i16 / u8 IQ converted in the load stage with iqformat.hip's arithmetic (fir_handle.hpp).
// WPB waves per workgroup: 16 (one workgroup per CU), or 4 for short batches -- four times as many workgroups, so that a
// batch of a few hundred segments still reaches every CU.
// NPH = 2: rate 4 -- the outputs y[8j + 4] too, from the same forward transforms through a second set of branch spectra
// (h[8m - c + 4]) and a second inverse half; the two phases leave interleaved.  mx.keep = m > 1: rates 8 m -- every m-th output of
// the rate-8 form is kept (the forward transforms, which are most of the work, are what any rate needs).
template <int HR, bool FM, class In, int WPB, int NPH>
__global__ __launch_bounds__(64 * WPB, 4) void fir_poly8_kernel(In in, const float2* __restrict__ hist,
                                                            int hist_len, void* __restrict__ out_any, size_t n, P8Tables tb,
                                                            float2* __restrict__ new_hist, unsigned chunk_log2, P8Mix mx,
                                                            P8Fm fmx, KStamp ks) {
    using Gm = P8Geom<HR>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    kstamp_begin(ks);
    hist_advance(hist, in, n, new_hist, hist_len);
    cf* gsp = reinterpret_cast<cf*>(smem);  // [NPH][16][64]
    cf* tw = gsp + 1024 * NPH;              // [16][8]
    cf* sc = tw + 128;                      // [64]
    const int l = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    cf* lds = sc + 64 + wave * P8_BUF;
    unsigned* ticket = reinterpret_cast<unsigned*>(sc + 64 + WPB * P8_BUF);
    for (int i = threadIdx.x; i < 1024 * NPH; i += 64 * WPB) gsp[i] = tb.g[i];
    if (threadIdx.x < 128) tw[threadIdx.x] = tb.tw[threadIdx.x];
    if (threadIdx.x < 64) sc[threadIdx.x] = tb.sc[threadIdx.x];
    if (threadIdx.x == 0) *ticket = 0;
    P8Lane lc;
    lc.w8j = tb.lane[0 * 64 + l];
    lc.tau = tb.lane[1 * 64 + l];
    lc.u0 = tb.lane[2 * 64 + l];
    lc.u1 = tb.lane[3 * 64 + l];
    lc.m1 = tb.lane[4 * 64 + l];
    lc.m2 = tb.lane[5 * 64 + l];
    lc.m3 = tb.lane[6 * 64 + l];
    __syncthreads();

    float2* out = static_cast<float2*>(out_any);
    float* outf = static_cast<float*>(out_any);
    const size_t n_out = n >> 3;
    const size_t nfull = n / Gm::NEW;  // segments whose new samples are all inside `in`
    const size_t inner = nfull > 1 ? nfull - 1 : 0;
    const bool contiguous = chunk_log2 >= 32u;
    const unsigned G = gridDim.x;
    const unsigned wg = (G % 8u == 0u) ? (blockIdx.x % 8u) * (G / 8u) + blockIdx.x / 8u : blockIdx.x;
    const size_t lo = 1 + blockIdx.x * inner / gridDim.x;
    const size_t hi = contiguous ? 1 + (blockIdx.x + 1) * inner / gridDim.x : nfull;
    auto draw = [&]() -> size_t {
        unsigned t = 0;
        if (l == 0) t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const unsigned tk = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(t)));
        if (contiguous) return lo + tk;
        const size_t c = static_cast<size_t>(tk >> chunk_log2) * G + wg;
        return 1 + (c << chunk_log2) + (tk & ((1u << chunk_log2) - 1u));
    };
    // Position l of a segment is output OUT seg + l - HQ (ya) and 64 more (yb); its mixer phase is that of input sample
    // 8 x the output index: a scalar part per segment + the lane's own
    const uint64_t lane_turns = mx.turns0 + static_cast<uint64_t>(static_cast<long long>(8 * (l - Gm::HQ))) * mx.frac;
    const uint64_t seg_turns = static_cast<uint64_t>(Gm::NEW) * mx.frac;
    const cf step512 = to_cf(mx.step512);
    const cf step4 = to_cf(mx.step4);
    auto emit = [&](size_t sg, cf ya, cf yb, cf ya2, cf yb2, bool guard) {
        const uint64_t tl = lane_turns + static_cast<uint64_t>(sg) * seg_turns;
        const cf rot = p8_rotor(static_cast<unsigned>(tl >> 32), sc);
        const cf rotb = cmulf_s(rot, step512);
        ya = cmulf(ya, rot);
        yb = cmulf(yb, rotb);
        const long long o = static_cast<long long>(sg * Gm::OUT) + l - Gm::HQ;  // ya's output; yb's: o + 64
        const bool a_on = l >= Gm::HQ && (!guard || static_cast<size_t>(o) < n_out);
        const bool b_on = !guard || static_cast<size_t>(o + 64) < n_out;
        if (NPH == 2) {  // rate 4: outputs 2 o (y[8 o]) and 2 o + 1 (y[8 o + 4]); n / 4 of them (n a multiple of 4, not of 8)
            ya2 = cmulf(ya2, cmulf_s(rot, step4));
            yb2 = cmulf(yb2, cmulf_s(rotb, step4));
            if (FM) {
                // the kept stream is A_l, B_l, A_{l+1}, ...: y[j - 1] of A_l is B of the lane below (position 64: lane 63 of the first
                // register pair), of B_l it is A_l.  Position HQ - 1 is a valid output of the segment before (the launcher's halo rule).
                const size_t n4f = n >> 2;
                float cx, cy;
                asm volatile("s_nop 1\n\tv_readlane_b32 %0, %2, 63\n\tv_readlane_b32 %1, %3, 63" : "=s"(cx), "=s"(cy) : "v"(ya2.x), "v"(ya2.y));
                float2 pa = make_float2(wave_shr1(ya2.x, 0.f), wave_shr1(ya2.y, 0.f));
                float2 pb = make_float2(wave_shr1(yb2.x, cx), wave_shr1(yb2.y, cy));
                if (guard && o == 0) pa = fmx.prev[0];
                if (Gm::HQ == 64 && guard && o == -64) pb = fmx.prev[0];  // (eight halo rows: the call's first output is position 64)
                const float f0 = fm_step_fast(to_f2(ya), pa), f1 = fm_step_fast(to_f2(ya2), to_f2(ya));
                const float g0 = fm_step_fast(to_f2(yb), pb), g1 = fm_step_fast(to_f2(yb2), to_f2(yb));
                const size_t fa = static_cast<size_t>(2 * o), fb = fa + 128;
                if (l >= Gm::HQ) {
                    if (!guard || fa < n4f) outf[fa] = f0;
                    if (!guard || fa + 1 < n4f) outf[fa + 1] = f1;
                    if (fa + 1 == n4f) fmx.prev_new[0] = to_f2(ya);
                    if (fa + 2 == n4f) fmx.prev_new[0] = to_f2(ya2);
                }
                if (!guard || fb < n4f) outf[fb] = g0;
                if (!guard || fb + 1 < n4f) outf[fb + 1] = g1;
                if (fb + 1 == n4f) fmx.prev_new[0] = to_f2(yb);
                if (fb + 2 == n4f) fmx.prev_new[0] = to_f2(yb2);
                return;
            }
            const size_t n4 = n >> 2;
            const size_t oa = static_cast<size_t>(2 * o), ob = oa + 128;
            if (mx.keep > 1) {  // rates 4 m (m odd): every m-th output of the rate-4 stream (32-bit indices: the launcher's condition)
                const unsigned f[4] = {static_cast<unsigned>(oa), static_cast<unsigned>(oa) + 1u, static_cast<unsigned>(ob), static_cast<unsigned>(ob) + 1u};
                const cf y4[4] = {ya, ya2, yb, yb2};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned t = __umulhi(mx.div_m, f[i]);
                    const unsigned q = (t + ((f[i] - t) >> 1)) >> mx.div_s;
                    if ((i >= 2 || l >= Gm::HQ) && q * mx.keep == f[i] && (!guard || f[i] < n4)) out[q] = to_f2(y4[i]);
                }
            } else {
                if (l >= Gm::HQ) {
                    if (!guard || oa < n4) out[oa] = to_f2(ya);
                    if (!guard || oa + 1 < n4) out[oa + 1] = to_f2(ya2);
                }
                if (!guard || ob < n4) out[ob] = to_f2(yb);
                if (!guard || ob + 1 < n4) out[ob + 1] = to_f2(yb2);
            }
        } else if (!FM) {
            if (mx.keep > 1) {  // rates 8 m: output o is kept when m divides it (o < 2^32: the launcher's condition)
                const unsigned oa = static_cast<unsigned>(o), ob = oa + 64u;
                const unsigned ta = __umulhi(mx.div_m, oa), tb2 = __umulhi(mx.div_m, ob);
                const unsigned qa = (ta + ((oa - ta) >> 1)) >> mx.div_s, qb = (tb2 + ((ob - tb2) >> 1)) >> mx.div_s;
                if (a_on && qa * mx.keep == oa) out[qa] = to_f2(ya);
                if (b_on && qb * mx.keep == ob) out[qb] = to_f2(yb);
            } else {
                if (a_on) out[o] = to_f2(ya);
                if (b_on) out[o + 64] = to_f2(yb);
            }
        } else {
            // y[j - 1]: the lane below; position 64's comes from lane 63 of ya.  Position HQ - 1 is a valid output (of the
            // segment before) whenever the filter leaves one spare halo position, which the launcher guarantees.
            float cx, cy;
            asm volatile("s_nop 1\n\tv_readlane_b32 %0, %2, 63\n\tv_readlane_b32 %1, %3, 63" : "=s"(cx), "=s"(cy) : "v"(ya.x), "v"(ya.y));
            float2 pa = make_float2(wave_shr1(ya.x, 0.f), wave_shr1(ya.y, 0.f));
            float2 pb = make_float2(wave_shr1(yb.x, cx), wave_shr1(yb.y, cy));
            if (guard && o == 0) pa = fmx.prev[0];  // the call's first output: FM.prev of the batch before
            if (Gm::HQ == 64 && guard && o == -64) pb = fmx.prev[0];  // (eight halo rows: it is position 64)
            const float fa = fm_step_fast(to_f2(ya), pa), fb = fm_step_fast(to_f2(yb), pb);
            if (a_on) outf[o] = fa;
            if (b_on) outf[o + 64] = fb;
            // the call's last output becomes FM.prev
            if (a_on && static_cast<size_t>(o) + 1 == n_out) fmx.prev_new[0] = to_f2(ya);
            if (b_on && static_cast<size_t>(o + 64) + 1 == n_out) fmx.prev_new[0] = to_f2(yb);
        }
    };

    cf v[16], p2[8], ya, yb, ya2 = cf{0.f, 0.f}, yb2 = cf{0.f, 0.f};
    // The stream's first segment (halo from the history) and its partial last one: wave 0 of the first / last workgroup,
    // before it joins the ticket loop
    if (wave == 0) {
        const size_t nseg = (n + Gm::NEW - 1) / Gm::NEW;
        for (int e = 0; e < 2; ++e) {
            const size_t sg = e ? nseg - 1 : 0;
            if (e ? (blockIdx.x != gridDim.x - 1 || nseg < 2 || nseg == nfull) : blockIdx.x != 0) continue;
            const long long b0 = static_cast<long long>(sg * Gm::NEW) - Gm::HALO + l;
#pragma unroll
            for (int a = 0; a < 16; ++a) v[a] = to_cf(stream_at(in, hist, hist_len, b0 + 64 * a, n));
            poly8_forward<NPH>(v, lds, tw, gsp, l, p2);
            poly8_inverse(lds, lc, l, ya, yb);
            if (NPH == 2) {
                poly8_put(lds, p2, l);
                poly8_inverse(lds, lc, l, ya2, yb2);
            }
            emit(sg, ya, yb, ya2, yb2, true);
        }
    }
    auto fetch = [&](size_t sg) {
        const size_t p = sg * Gm::NEW - Gm::HALO + l;
#pragma unroll
        for (int a = 0; a < 16; ++a) v[a] = to_cf(in[p + 64 * a]);
    };
    // Order within an iteration: forward half of this segment; the STORES of the segment before it; the loads of the
    // next one; inverse half.  vmcnt counts loads and stores in one in-order queue: with the stores issued in front of the
    // loads, the wait for the rows at the top of the loop does not wait for stores issued a moment earlier.
    size_t seg = draw(), seg_prev = 0;
    bool have = false;
    if (seg < hi) fetch(seg);
    while (seg < hi) {
        const size_t seg_next = draw();
        poly8_forward<NPH>(v, lds, tw, gsp, l, p2);
        if (have) emit(seg_prev, ya, yb, ya2, yb2, false);
        if (seg_next < hi) fetch(seg_next);  // v is dead: the rows travel while the inverse half runs (few registers)
        poly8_inverse(lds, lc, l, ya, yb);
        if (NPH == 2) {
            poly8_put(lds, p2, l);
            poly8_inverse(lds, lc, l, ya2, yb2);
        }
        seg_prev = seg;
        have = true;
        seg = seg_next;
    }
    if (have) emit(seg_prev, ya, yb, ya2, yb2, false);
    kstamp_end(ks);
}

}  // namespace comms

using namespace comms;

namespace {

const double kPi8 = 3.14159265358979323846264338327950288;
float2 root(long long e, int denom, int sign) {  // e^{sign 2 pi i e / denom}
    e %= denom;
    if (e < 0) e += denom;
    const double a = 2.0 * kPi8 * static_cast<double>(e) / denom;
    return make_float2(static_cast<float>(std::cos(a)), static_cast<float>(sign * std::sin(a)));
}

}  // namespace

// Tables of the handle for mixer increment `frac` (the mixer-first chain folds the mixer into the taps:
// sum_k h[k] x[n-k] e^{i phi(n-k)} = e^{i phi(n)} sum_k (h[k] e^{-i k dphi}) x[n-k]).  Built once per (handle, frac, order).
static comms_status_t poly8_prepare(comms_fir* h, bool pre, uint64_t frac, hipStream_t s) {
    if (h->d_p8 && h->p8_pre == pre && (!pre || h->p8_frac == frac)) return COMMS_OK;
    const int N = h->n_eff;
    std::vector<double> hr(528, 0.0), hi(528, 0.0);
    const double dphi = static_cast<double>(frac >> 11) * (kMixT * 0x1.0p-53);
    for (int k = 0; k < N; ++k) {
        double tr = h->taps[k].re, ti = h->taps[k].im;
        if (pre) {
            const double ang = -dphi * static_cast<double>(k);
            const double cr = std::cos(ang), ci = std::sin(ang);
            const double r = tr * cr - ti * ci;
            ti = tr * ci + ti * cr;
            tr = r;
        }
        hr[k] = tr;
        hi[k] = ti;
    }
    std::vector<float2> t(2048 + static_cast<size_t>(P8_TAB) + 7 * 64);
    float2* g = t.data();
    float2* tw = g + 2048;
    float2* sc = tw + 128;
    float2* ln = sc + 64;
    // G_c[k] = (1/128) sum_m h[8m - c + shift] W128^{mk}: shift 0 for the outputs y[8j], 4 for y[8j + 4] (rate 4)
    std::vector<double> cs(128), sn(128);
    for (int i = 0; i < 128; ++i) {
        cs[i] = std::cos(2.0 * kPi8 * i / 128.0);
        sn[i] = -std::sin(2.0 * kPi8 * i / 128.0);
    }
    for (int ph = 0; ph < 2; ++ph)
        for (int lane = 0; lane < 64; ++lane) {
            const int k1 = lane & 15, cg = lane >> 4;
            for (int ci = 0; ci < 2; ++ci)
                for (int k2 = 0; k2 < 8; ++k2) {
                    const int c = cg + 4 * ci, k = k1 + 16 * k2;
                    double re = 0.0, im = 0.0;
                    for (int m = 0; m <= 64; ++m) {
                        const int tap = 8 * m - c + 4 * ph;
                        if (tap < 0 || tap >= N) continue;
                        const int e = (m * k) & 127;
                        re += hr[tap] * cs[e] - hi[tap] * sn[e];
                        im += hr[tap] * sn[e] + hi[tap] * cs[e];
                    }
                    g[1024 * ph + (8 * ci + k2) * 64 + lane] = make_float2(static_cast<float>(re / 128.0), static_cast<float>(im / 128.0));
                }
        }
    for (int k1 = 0; k1 < 16; ++k1)
        for (int d = 0; d < 8; ++d) tw[k1 * 8 + d] = root(static_cast<long long>(d) * k1, 128, -1);
    for (int i = 0; i < 64; ++i) sc[i] = root(i, 64, +1);
    for (int lane = 0; lane < 64; ++lane) {
        const int k1 = lane & 15, j = lane >> 4, j0 = j & 1, j1 = j >> 1, q1 = 2 * j0 + j1, ka = (lane & 31) >> 3;
        ln[0 * 64 + lane] = root(j, 8, +1);
        ln[1 * 64 + lane] = j0 ? make_float2(0.f, 1.f) : make_float2(1.f, 0.f);
        ln[2 * 64 + lane] = root(static_cast<long long>(q1) * k1, 128, +1);
        ln[3 * 64 + lane] = root(static_cast<long long>(q1 + 4) * k1, 128, +1);
        for (int tq = 1; tq < 4; ++tq) ln[(3 + tq) * 64 + lane] = root(static_cast<long long>(tq) * ka, 16, +1);
    }
    if (!h->d_p8) COMMS_HIP_TRY(hipMalloc(&h->d_p8, t.size() * sizeof(float2)));
    // (in stream order behind the launches that still read the old tables; the source is pageable memory, so the call returns
    // once the copy has been staged)
    COMMS_HIP_TRY(hipMemcpyAsync(h->d_p8, t.data(), t.size() * sizeof(float2), hipMemcpyHostToDevice, s));
    COMMS_HIP_TRY(hipStreamSynchronize(s));
    h->p8_pre = pre;
    h->p8_frac = frac;
    return COMMS_OK;
}

namespace {

// Halo rows for (taps, FM demod), or 0 when the kernel cannot run them: 8 HR positions per phase hold the 8 HR + 1 taps of a
// branch filter; FM demod takes one position more (the output in front of the segment's first).
int poly8_halo_rows(int n_eff, bool fm) {
    const int hq = (n_eff - 1 + 7) / 8 + (fm ? 1 : 0);
    const int hr = hq <= 16 ? 2 : (hq + 7) / 8;
    return hr <= 6 ? hr : hr <= 8 ? 8 : 0;  // (built for 2 ... 6 and 8)
}

template <int HR, bool FM, class In, int WPB, int NPH>
comms_status_t poly8_launch_w(comms_fir* h, hipStream_t s, In in, void* out, size_t n, const P8Tables& tb, const P8Mix& mx,
                              const P8Fm& fmx) {
    using Gm = P8Geom<HR>;
    constexpr size_t lds = p8_lds_bytes(WPB, NPH);
    const size_t nseg = (n + Gm::NEW - 1) / Gm::NEW;
    const size_t want = (nseg + WPB - 1) / WPB;
    const size_t slots = static_cast<size_t>(kNumCU) * (16 / WPB);
    const dim3 grid(static_cast<unsigned>(want < slots ? want : slots));
    static const int chunk_knob = diag_knob("COMMS_POLY8_CHUNK_LOG2", -1);
    const unsigned chunk_log2 = chunk_knob >= 0 ? static_cast<unsigned>(chunk_knob) : nseg < 160u * static_cast<size_t>(kNumCU) ? 1u : 3u;
    static DeviceOnce attr_once;
    if (attr_once.need())
        COMMS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fir_poly8_kernel<HR, FM, In, WPB, NPH>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipEvent_t ea = nullptr, eb = nullptr;
    (void)h->take_events(ea, eb);
    const KStamp ks = h->next_stamp();
    if (ea)
        hipExtLaunchKernelGGL((fir_poly8_kernel<HR, FM, In, WPB, NPH>), grid, dim3(64 * WPB), static_cast<uint32_t>(lds), s, ea, eb, 0u, in,
                              h->d_hist[h->cur], h->n_eff, out, n, tb, h->d_hist[h->cur ^ 1], chunk_log2, mx, fmx, ks);
    else
        fir_poly8_kernel<HR, FM, In, WPB, NPH><<<grid, dim3(64 * WPB), lds, s>>>(in, h->d_hist[h->cur], h->n_eff, out, n, tb,
                                                                                h->d_hist[h->cur ^ 1], chunk_log2, mx, fmx, ks);
    return launch_ok("fir_poly8_kernel");
}

template <int HR, bool FM, class In, int NPH>
comms_status_t poly8_launch(comms_fir* h, hipStream_t s, In in, void* out, size_t n, const P8Tables& tb, const P8Mix& mx,
                            const P8Fm& fmx) {
    // short batches (fewer segments than the chip has wave slots): four-wave workgroups
    static const int wpb_knob = diag_knob("COMMS_POLY8_WPB", 0);
    const size_t nseg = (n + P8Geom<HR>::NEW - 1) / P8Geom<HR>::NEW;
    const bool small = wpb_knob ? wpb_knob == 4 : nseg < 16u * static_cast<size_t>(kNumCU);
    return small ? poly8_launch_w<HR, FM, In, 4, NPH>(h, s, in, out, n, tb, mx, fmx) : poly8_launch_w<HR, FM, In, 16, NPH>(h, s, in, out, n, tb, mx, fmx);
}

template <class In>
comms_status_t poly8_launch_in(int hr, bool fm, int nph, comms_fir* h, hipStream_t s, In in, void* out, size_t n, const P8Tables& tb,
                               const P8Mix& mx, const P8Fm& fmx) {
#define P8_CASE(HRV, NPHV) \
    case HRV: return fm ? poly8_launch<HRV, true, In, NPHV>(h, s, in, out, n, tb, mx, fmx) : poly8_launch<HRV, false, In, NPHV>(h, s, in, out, n, tb, mx, fmx)
    if (nph == 2) {
        switch (hr) {
            P8_CASE(2, 2);
            P8_CASE(3, 2);
            P8_CASE(4, 2);
            P8_CASE(5, 2);
            P8_CASE(6, 2);
            default: return fm ? poly8_launch<8, true, In, 2>(h, s, in, out, n, tb, mx, fmx) : poly8_launch<8, false, In, 2>(h, s, in, out, n, tb, mx, fmx);
        }
    }
    switch (hr) {
        P8_CASE(2, 1);
        P8_CASE(3, 1);
        P8_CASE(4, 1);
        P8_CASE(5, 1);
        P8_CASE(6, 1);
        default: return fm ? poly8_launch<8, true, In, 1>(h, s, in, out, n, tb, mx, fmx) : poly8_launch<8, false, In, 1>(h, s, in, out, n, tb, mx, fmx);
    }
#undef P8_CASE
}

// What the kernel does for a decimation rate: 1: 8 (one output phase), 2: 4 (two), 3: 8 m up to 64 (every m-th output of the rate-8
// form: the forward transforms are what any rate needs); 0: not this kernel
// 4: rates 4 m', m' odd, up to 60 (every m'-th output of the rate-4 form)
int poly8_rate_kind(uint32_t rate) {
    return rate == 8 ? 1 : rate == 4 ? 2 : (rate % 8 == 0 && rate >= 16 && rate <= 64) ? 3 : (rate % 4 == 0 && rate >= 12 && rate <= 60) ? 4 : 0;
}

}  // namespace

extern "C" {

// 0: no; 1: this chain (taps, rate, stages, batch) can run on the polyphase frequency-domain kernel; 2: and it is the faster form
int32_t comms_fir_poly8_supported(const comms_fir_t* h, uint32_t rate, int32_t mode, size_t n) {
    const int kind = poly8_rate_kind(rate);
    if (!h || !kind || h->n_eff < 1 || n < rate) return 0;
    const bool fm = (mode & COMMS_CHAIN_FM) != 0;
    if (!(mode & COMMS_CHAIN_DEC) || !poly8_halo_rows(h->n_eff, fm) || (fm && kind > 2)) return 0;  // (FM demod: rates 8 and 4)
    if (kind >= 3 && (n >> 2) > 0xFFFFFFFFull) return 0;  // (its output index arithmetic is 32 bits wide)
    static const int knob = diag_knob("COMMS_POLY8", 1);          // 0: never, 1: where it wins, 2: wherever it can run
    if (!knob || h->no_poly8) return 0;
    if (knob == 2) return 2;
    // Against the time-domain kernels (scripts/sweep_poly8.py, profiles/r05_sweep_poly8.txt: taps x batch length, with and
    // without FM demod): its time does not depend on the taps -- ahead from 64 taps at every batch length (255 taps: 47 -> 29 us
    // at 2^24 samples, 12 -> 7 us at 2^14), level with them on shorter filters up to 2^24 samples, 7 % ahead at 2^26.
    static const int min_taps = diag_knob("COMMS_POLY8_MIN_TAPS", 64);
    static const int min_log2 = diag_knob("COMMS_POLY8_MIN_LOG2", 25);
    const int N = h->n_eff;
    if (kind == 1) return N >= min_taps || n >= (static_cast<size_t>(1) << min_log2) ? 2 : 1;
    // Rates 4 and 8 m against the time-domain kernels and the overlap-save fusion (scripts/sweep_poly8_rates.py,
    // profiles/r05_sweep_poly8_rates.txt).  Rate 4 (two output phases; 255 taps at 2^24 samples: 61 -> 36 us): ahead from 64 taps
    // on long batches, from 128 taps at every length.  Rates 16 ... 56: ahead or level everywhere from 32 taps (255 taps at rate
    // 16: 38.8 -> 26.5 us).  Rate 64: the any-rate kernel reads only its windows when rate >= taps and keeps the short batches
    // (4.5 against 6.6 us at 2^16 samples); from 2^23 samples, or 128 taps, this one.
    if (kind == 2) return N >= 2 * min_taps || (N >= min_taps && n >= (static_cast<size_t>(1) << 22)) ? 2 : 1;
    // rates 12, 20, ... 60 (two output phases, every (rate / 4)-th output kept; 255 taps at 2^24 samples: rate 12 39.2 -> 34.7 us, 20
    // 41.8 -> 34.9, 28 level, 44 and 60 behind the any-rate kernel): ahead at 12 and 20 with long filters only
    // (beyond 257 taps the any-rate kernel is the alternative: 300 taps at 2^24 samples 70 us at rate 20, 50 at 32, 39 at 48 against
    // 36 ... 45 us here -- profiles/r05_sweep_long_taps.txt)
    if (kind == 4) return (rate <= 20 && N >= 3 * min_taps) || (N > 257 && rate <= 36) ? 2 : 1;
    if (rate < 64) return N >= min_taps / 2 ? 2 : 1;
    return N >= 2 * min_taps || n >= (static_cast<size_t>(1) << 23) ? 2 : 1;
}

comms_status_t comms_fir_run_poly8_dev(comms_fir_t* h, const void* d_in, size_t n, void* d_out, int32_t mode,
                                       uint64_t turns0, uint64_t frac, uint32_t rate, const void* fm_prev, void* fm_prev_new,
                                       void* stream) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG((d_in && d_out) || !n, "NULL device pointer");
    const int kind = poly8_rate_kind(rate);
    COMMS_ARG(kind != 0, "the polyphase kernel runs rates 4, 8 and multiples of 4 up to 64");
    COMMS_ARG(n % rate == 0, "n must be a multiple of the decimation rate");
    const bool fm = (mode & COMMS_CHAIN_FM) != 0;
    const int hr = poly8_halo_rows(h->n_eff, fm);
    COMMS_ARG(hr != 0, "the polyphase kernel takes <= 513 taps (505 with FM demod)");
    COMMS_ARG(!fm || kind <= 2, "the polyphase kernel demodulates at rates 8 and 4 only");
    COMMS_ARG(kind < 3 || (n >> 2) <= 0xFFFFFFFFull, "batch too long for the polyphase kernel at this rate");
    COMMS_ARG((mode & COMMS_CHAIN_DEC) && !((mode & COMMS_CHAIN_PRE) && (mode & COMMS_CHAIN_POST)), "bad chain mode");
    COMMS_ARG(!fm || (fm_prev && fm_prev_new), "FM demod needs its state");
    COMMS_TRY(fir_check_sticky(h));
    COMMS_TRY(use_device(h->device));
    if (!n) return COMMS_OK;
    const size_t in_elem = h->in_fmt == COMMS_IQ_I16 ? 4 : h->in_fmt == COMMS_IQ_U8 ? 2 : 8;
    COMMS_ARG(!ranges_overlap(d_in, n * in_elem, d_out, (n / rate) * (fm ? 4 : 8)), "the decimating chain cannot run in place");
    COMMS_ARG((reinterpret_cast<uintptr_t>(d_in) & (in_elem - 1)) == 0, "input must be aligned to one IQ sample");
    hipStream_t s = nullptr;
    COMMS_TRY(h->enter(stream, &s));
    COMMS_TRY(poly8_prepare(h, (mode & COMMS_CHAIN_PRE) != 0, frac, s));
    P8Tables tb;
    tb.g = reinterpret_cast<const cf*>(h->d_p8);
    tb.tw = tb.g + 2048;
    tb.sc = tb.tw + 128;
    tb.lane = tb.sc + 64;
    P8Mix mx{};
    mx.turns0 = turns0;
    mx.frac = frac;
    double c, sn;
    mix_host_rotor(512u * frac, c, sn);
    mx.step512 = make_float2(static_cast<float>(c), static_cast<float>(sn));
    mix_host_rotor(4u * frac, c, sn);
    mx.step4 = make_float2(static_cast<float>(c), static_cast<float>(sn));
    mx.keep = kind == 3 ? rate / 8 : kind == 4 ? rate / 4 : 1;
    if (mx.keep > 1) {  // o / m by multiplication (Granlund - Montgomery): l = ceil(log2 m), M = floor(2^32 (2^l - m) / m) + 1
        unsigned lg = 0;
        while ((1u << lg) < mx.keep) ++lg;
        mx.div_m = static_cast<unsigned>(((static_cast<uint64_t>((1u << lg) - mx.keep) << 32) / mx.keep) + 1);
        mx.div_s = lg - 1;
    }
    P8Fm fmx{static_cast<const float2*>(fm_prev), static_cast<float2*>(fm_prev_new)};
    const int nph = kind == 2 || kind == 4 ? 2 : 1;
    comms_status_t st;
    if (h->in_fmt == COMMS_IQ_I16)
        st = poly8_launch_in(hr, fm, nph, h, s, InI16{static_cast<const short2*>(d_in), h->in_scale}, d_out, n, tb, mx, fmx);
    else if (h->in_fmt == COMMS_IQ_U8)
        st = poly8_launch_in(hr, fm, nph, h, s, InU8{static_cast<const uchar2*>(d_in)}, d_out, n, tb, mx, fmx);
    else
        st = poly8_launch_in(hr, fm, nph, h, s, static_cast<const float2*>(d_in), d_out, n, tb, mx, fmx);
    COMMS_TRY(st);
    h->cur ^= 1;
    h->last_poly8 = true;
    return COMMS_OK;
}

}  // extern "C"
