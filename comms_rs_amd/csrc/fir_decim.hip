// fir_decim.hip -- the fused chain in time domain when a decimator follows the FIR.
//
//   mixer -> FIR -> keep every R-th [-> FM demod]      (BASELINE config 3; examples/fm_radio.rs:146-148)
//   FIR -> mixer -> keep every R-th [-> FM demod]      (the BASELINE metric's chain)
// Same results as MixerNode (src/mixer.rs:73-85), BatchFirNode (src/filter/fir.rs:87-102),
// DecimateNode (src/util/resample_node.rs:53-65) and FMDemodNode (src/modulation/analog.rs:22-35)
// in series.  Only every R-th filter output survives the decimator, so the kernel computes
// only those: N/R complex MACs per INPUT sample (16 for config 3) instead of a full-rate
// filter -- cheaper than the FFT path up to a few hundred taps, and 8 B read + 8/R (4/R) B
// written per input sample are all that touch HBM.
//
// Layout: a 256-lane workgroup owns 512 consecutive outputs (two per lane).  The inputs of
// the tile (mixed on the way in for the mixer-first chain) are staged in LDS split into
// 2R phase arrays -- sample s at [s mod 2R][s div 2R] -- so that for a fixed tap every lane
// reads the same phase array at consecutive indices (conflict-free ds_read_b64), and one
// LDS read feeds both outputs of the lane.  Taps are walked in the reference's order
// (k ascending) as packed FMAs on {re, im}; they come from the kernel-argument segment,
// i.e. scalar loads into SGPR pairs that the packed FMA reads directly (op_sel picks the
// half): no tap ever occupies a VGPR or an LDS slot.  Four workgroups per CU (37 KiB of LDS
// each at R = 8) cover each other's load / compute / store phases.
#include <hip/hip_ext.h>

#include <cmath>
#include <vector>

#include "common.hpp"
#include "fft_radix.hpp"
#include "fir_handle.hpp"
#include "sgpr_mac.hpp"

#ifndef COMMS_DECIM_PREFETCH
#define COMMS_DECIM_PREFETCH 0
#endif
#ifndef COMMS_DECIM_TILE_DEFAULT
#define COMMS_DECIM_TILE_DEFAULT 0
#endif
#ifndef COMMS_DECIM_NOMAC_DEFAULT
#define COMMS_DECIM_NOMAC_DEFAULT 0
#endif
#ifndef COMMS_DECIM_WAVE_DEFAULT
#define COMMS_DECIM_WAVE_DEFAULT 1
#endif
#ifndef COMMS_DECIM_WAVE_NT_DEFAULT
#define COMMS_DECIM_WAVE_NT_DEFAULT -1
#endif
#ifndef COMMS_DECIM_WAVE_SPLIT_DEFAULT
#define COMMS_DECIM_WAVE_SPLIT_DEFAULT 1
#endif
#ifndef COMMS_DECIM_WAVE_WPB_DEFAULT
#define COMMS_DECIM_WAVE_WPB_DEFAULT 1
#endif

namespace comms {

constexpr int DC_TILE = 512;        // outputs per tile (DcGeom<.., TILE>: 1024 in the diagnostic build's wide variant)
constexpr int DC_NMAX = 257;        // taps (kernel-argument budget)
constexpr int DC_AMAX = 384;        // padded tap array: OPL*R*nd + (OPL-1)*R + pair slack (worst: R = 12, OPL = 4: 376)
constexpr int DC_RMAX = 16;
constexpr int DC_OPLMAX = 4;

// A lane owns OPL consecutive outputs (2 or 4), a workgroup of 512 / OPL lanes one tile of 512.  Every LDS
// read of the filter loop feeds OPL MACs: at OPL = 2 the loop issues one ds_read_b64 per two packed FMAs and
// the LDS array is as busy as the vector ALU (the two do not overlap perfectly: ~10 cycles per MAC and SIMD
// measured against 4.9 for the FMA alone); OPL = 4 halves the LDS traffic for the same FMAs, at half the
// waves per CU (the staged inputs of a tile bound how many outputs can be resident).
template <int R, int OPL, int TILE = DC_TILE>
struct DcGeom {
    static constexpr int PR = OPL * R;                             // phases
    static constexpr int WG = TILE / OPL;                          // lanes per workgroup
    static constexpr int HLQ_MAX = (DC_NMAX - 1 + PR - 1) / PR;    // halo in phase-array elements
    static constexpr int HROWS = (HLQ_MAX * PR + WG - 1) / WG;     // halo rows of WG samples
    // phase-array stride: odd, so that a staging row (one ds_write_b64 per lane, sixteen lanes at a time over sixteen 8-byte bank
    // pairs) spreads over the banks -- exact for sixteen phases and the best there is for the other rates (brute force: 0.3-1 extra
    // cycle per group whatever the stride) except four and eight phases (rates 2 and 4), which need 16 / PR modulo 16 (see DwGeom)
    static constexpr int SB = WG + HLQ_MAX + 1;
    static constexpr int S = (PR == 4 || PR == 8) ? (SB + 15 - ((SB + 15 - 16 / PR) % 16)) : (SB | 1);
    static constexpr size_t LDS = static_cast<size_t>(PR) * S * sizeof(float2);
    static constexpr int WGPC = (R <= 8 ? 4 : R <= 12 ? 3 : 2) * DC_TILE / TILE;  // workgroups per CU that fit in LDS
    static constexpr int WAVES_PER_SIMD = (WGPC * WG / 64 + 3) / 4;  // __launch_bounds__' second argument: sets the VGPR budget
    // ... with the mixer in front (PRE: row rotors in registers) rate 12 does not fit three workgroups' budget of 168 VGPRs (148
    // bytes of scratch per lane): two -- 2^24 samples, 31 / 63 / 127 / 255 taps: 33.6 / 35.3 / 38.9 / 46.5 -> 27.9 / 30.1 / 36.3 / 46.0 us
    // (scripts/ab_pre_rates.py).  Rate 11 (92 bytes) gains as much on short filters and loses it on long ones (255 taps: 46.5 ->
    // 50.0 us): it stays at three.
    template <bool PRE>
    static constexpr int wgpc() { return PRE && R == 12 ? 2 * DC_TILE / TILE : WGPC; }
    template <bool PRE>
    static constexpr int waves_per_simd() { return (wgpc<PRE>() * WG / 64 + 3) / 4; }
    static_assert(PR * (HLQ_MAX + 1) + (OPL - 1) * R + 4 <= DC_AMAX, "tap table too small");
};

struct DecimArgs {
    const void* in;                // n samples in format `fmt`
    const float2* hist;
    float2* new_hist;
    void* out;
    const float2* fm_prev;
    float2* fm_prev_new;
    size_t n, n_out, n_tiles;
    int hist_len, hlq, nd, mode;   // hlq = ceil((N-1)/PR); nd = hlq + 1 tap blocks of PR
    int fmt;                       // COMMS_IQ_*
    float in_scale;
    uint64_t turns0, frac;         // mixer phase of input sample 0 and per-sample increment (turns)
    double tile_c, tile_s;         // e^{i * R * tile_step * dphi}
    float2 step_r[DC_OPLMAX];      // e^{i * c * R * dphi}  (output c of a lane, mixer-after-FIR)
    float2 step[DC_OPLMAX * DC_RMAX];  // e^{i * WG m * dphi}, staging row m
    float are[DC_AMAX];            // A[m] = Re h[m - (PR-1)], zero outside [0, N)
    float aim[DC_AMAX];
    const float* qt;               // OPL = 4: the taps of one LDS read stored together, Q[m][c] = A[m + R c] (device memory)
    unsigned long long* stamps;    // diagnostic (scripts/stamp_decim.py): per-wave cycles per phase, or NULL
    int interleave;                // tile t of workgroup b: b + i gridDim.x instead of a contiguous run (kept behind the tap
                                   // arrays: their offsets decide how many scalar-cache lines a 64-byte tap load touches)
    KStamp ks;                     // in-kernel begin / end stamps of a stamps timer, or null
    // wave-private form (fir_decim_wave_kernel): a wave's run of tiles, the row rotors of its halo rows
    int nt_chunk;                  // tiles of 128 outputs per run
    long long n_chunks;            // runs
    float2 step_h[4];              // e^{i * 64 (i - HR) * dphi}, halo row i
};

constexpr double kTwoPiD = 2.0 * 3.14159265358979323846264338327950288;

__device__ __forceinline__ void rotor_at(uint64_t turns, double& c, double& s) {
    sincos(static_cast<double>(turns >> 11) * (kTwoPiD * 0x1.0p-53), &s, &c);
}
__device__ __forceinline__ void rotor_step(double& c, double& s, double sc, double ss) {
    const double nc = c * sc - s * ss;
    s = c * ss + s * sc;
    c = nc;
}

// A sample of the input stream, read once by this launch (the tile's halo a second time).  COMMS_DECIM_NT=1 (trial
// build): Complex<f32> rows as nontemporal loads.
#ifndef COMMS_DECIM_NT
#define COMMS_DECIM_NT 0
#endif
template <class In>
__device__ __forceinline__ float2 ld_once(In in, size_t i) { return in[i]; }
#if COMMS_DECIM_NT
__device__ __forceinline__ float2 ld_once(const float2* in, size_t i) {
    typedef float nt_f2 __attribute__((ext_vector_type(2)));
    const nt_f2 q = __builtin_nontemporal_load(reinterpret_cast<const nt_f2*>(in) + i);
    return make_float2(q.x, q.y);
}
#endif

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, which would
// stall on the next tile's global loads that are deliberately left in flight across it.
__device__ __forceinline__ void lds_barrier() {  // (kept light: nothing global is shared between waves here)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// PRE: the mixer sits in front of the FIR (samples are mixed on their way into LDS); otherwise it
// follows the FIR (or is absent).
template <int R, int OPL, bool REAL, bool PRE, int TILE = DC_TILE, int CHX = 0>
__global__ __launch_bounds__((DcGeom<R, OPL, TILE>::WG), (DcGeom<R, OPL, TILE>::template waves_per_simd<PRE>())) void fir_decim_kernel(const DecimArgs a) {
    using G = DcGeom<R, OPL, TILE>;
    constexpr int PR = G::PR, S = G::S, WG = G::WG, HROWS = G::HROWS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* sh = reinterpret_cast<cf*>(smem);  // [PR][S]
    __shared__ float2 sh_y[WG / 64];
    switch (a.fmt) {
        case COMMS_IQ_I16: hist_advance(a.hist, InI16{static_cast<const short2*>(a.in), a.in_scale}, a.n, a.new_hist, a.hist_len); break;
        case COMMS_IQ_U8: hist_advance(a.hist, InU8{static_cast<const uchar2*>(a.in)}, a.n, a.new_hist, a.hist_len); break;
        default: hist_advance(a.hist, static_cast<const float2*>(a.in), a.n, a.new_hist, a.hist_len); break;
    }

    kstamp_begin(a.ks);
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    constexpr bool pre = PRE;
    const bool post = !PRE && (a.mode & COMMS_CHAIN_POST) != 0;
    const bool fm = (a.mode & COMMS_CHAIN_FM) != 0;
    const int ovl = fm ? 1 : 0;           // FM tiles recompute the previous tile's last output
    const long long ts = TILE - ovl;      // stored outputs per tile
    const int hl = a.hlq * PR;            // halo samples staged to the left of the tile

    // tiles of a workgroup: a contiguous run, or (a.interleave) every gridDim.x-th tile, so that the chip
    // sweeps the stream as one window
    const size_t t0 = a.interleave ? blockIdx.x : static_cast<size_t>(blockIdx.x) * a.n_tiles / gridDim.x;
    const size_t t1 = a.interleave ? a.n_tiles : static_cast<size_t>(blockIdx.x + 1) * a.n_tiles / gridDim.x;
    const size_t tstep = a.interleave ? gridDim.x : 1;
    if (t0 >= t1) return;

    // Mixer rotors.  The phase of input sample i = ib + tid + WG m of a tile splits into a part
    // that is the same for the whole tile, T = rot(ib) (f64, stepped per tile, applied to the
    // outputs after the filter -- the filter is linear), and a part that never changes,
    // lrow[m] = e^{i (tid + WG m) dphi} (f32, set up once): one multiply per staged sample.
    //   mixer after the FIR: ro = rot(R (jb + OPL tid)) is this lane's first output's rotor.
    // (prefetching build: the row rotors are re-made from the lane's l0 / h0 and the wave-uniform steps for every tile --
    // two packed operations per staged row -- instead of living in 2 (PR + HROWS) registers through the filter loop,
    // where the next tile's rows now wait; the products are the same f32 operations either way)
    constexpr bool kRowRotorsPerTile = COMMS_DECIM_PREFETCH != 0;
    constexpr int NROW = PRE && !kRowRotorsPerTile ? PR : 1;
    cf lrow[NROW], lhalo[kRowRotorsPerTile ? 1 : HROWS];
    double tt_c = 1.0, tt_s = 0.0, ro_c = 1.0, ro_s = 0.0;
    {
        const long long jb0 = static_cast<long long>(t0) * ts - ovl;
        if (pre) {
            double c, sn;
            rotor_at(static_cast<uint64_t>(tid) * a.frac, c, sn);
            const cf l0 = cf{static_cast<float>(c), static_cast<float>(sn)};
            rotor_at(static_cast<uint64_t>(static_cast<long long>(tid) - hl) * a.frac, c, sn);
            const cf h0 = cf{static_cast<float>(c), static_cast<float>(sn)};
#pragma unroll
            for (int m = 0; m < NROW; ++m) lrow[m] = m ? cmulf(l0, to_cf(a.step[m])) : l0;
#pragma unroll
            for (int m = 0; m < (kRowRotorsPerTile ? 1 : HROWS); ++m) lhalo[m] = m ? cmulf(h0, to_cf(a.step[m])) : h0;
            rotor_at(a.turns0 + static_cast<uint64_t>(R * jb0) * a.frac, tt_c, tt_s);
        }
        if (post) rotor_at(a.turns0 + static_cast<uint64_t>(R * (jb0 + OPL * tid)) * a.frac, ro_c, ro_s);
    }
    // LDS slots of this lane's staged samples (sample hl + tid + WG m of the tile; halo: tid + WG m)
    // (when PR divides WG, row m sits WG / PR elements behind row 0 in the same phase array: one register
    // and immediate offsets instead of PR registers)
    constexpr bool kLinearSlots = WG % PR == 0;
    constexpr int NSLOT = kLinearSlots ? 1 : PR;
    int slot[NSLOT], slot_h[HROWS];
#pragma unroll
    for (int m = 0; m < NSLOT; ++m) {
        const unsigned s = static_cast<unsigned>(tid + WG * m);
        slot[m] = static_cast<int>((s % PR) * S + a.hlq + s / PR);
    }
#pragma unroll
    for (int m = 0; m < HROWS; ++m) {
        const unsigned s = static_cast<unsigned>(tid + WG * m);
        slot_h[m] = s < static_cast<unsigned>(hl) ? static_cast<int>((s % PR) * S + s / PR) : -1;
    }

    // The tile's samples travel global -> VGPRs -> (format conversion, mixer) -> LDS.  (Requesting the next
    // tile's rows before this tile's filter loop was measured, twice, and does not pay: 47.5 -> 48.5 us on the
    // metric chain, 147 -> 152 us on config 3 -- the workgroups of a CU already cover each other's loads.)
    cf x[PR], xh[HROWS];
    auto load_tile_from = [&](auto in, size_t t) __attribute__((always_inline)) {
        const long long ib = R * (static_cast<long long>(t) * ts - ovl);
        if (ib - hl >= 0 && static_cast<size_t>(ib) + static_cast<size_t>(WG) * PR <= a.n) {  // interior tile: no edge handling
            const size_t base = static_cast<size_t>(ib) + static_cast<unsigned>(tid);
#pragma unroll
            for (int m = 0; m < PR; ++m) x[m] = to_cf(ld_once(in, base + WG * m));
#pragma unroll
            for (int m = 0; m < HROWS; ++m) xh[m] = slot_h[m] >= 0 ? to_cf(ld_once(in, base - hl + WG * m)) : cf{0.f, 0.f};
        } else {
#pragma unroll
            for (int m = 0; m < PR; ++m) x[m] = to_cf(stream_at(in, a.hist, a.hist_len, ib + tid + WG * m, a.n));
#pragma unroll
            for (int m = 0; m < HROWS; ++m)
                xh[m] = slot_h[m] >= 0 ? to_cf(stream_at(in, a.hist, a.hist_len, ib - hl + tid + WG * m, a.n))
                                       : cf{0.f, 0.f};
        }
    };
    auto load_tile = [&](size_t t) __attribute__((always_inline)) {
        switch (a.fmt) {   // wave-uniform: one scalar branch per tile
            case COMMS_IQ_I16: load_tile_from(InI16{static_cast<const short2*>(a.in), a.in_scale}, t); break;
            case COMMS_IQ_U8: load_tile_from(InU8{static_cast<const uchar2*>(a.in)}, t); break;
            default: load_tile_from(static_cast<const float2*>(a.in), t); break;
        }
    };

    // diagnostic only: cycles per phase, summed per wave (every stamp drains the memory counters)
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long st_prev = 0;
#define DC_STAMP(i)                                                     \
    if (a.stamps) {                                                     \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     \
        const unsigned long long st_now = __builtin_amdgcn_s_memtime(); \
        st_acc[i] += st_now - st_prev;                                  \
        st_prev = st_now;                                               \
    }
    if (a.stamps) st_prev = __builtin_amdgcn_s_memtime();

#if COMMS_DECIM_PREFETCH
    load_tile(t0);
#endif
    for (size_t t = t0; t < t1; t += tstep) {
        const long long jb = static_cast<long long>(t) * ts - ovl;  // first output computed by this tile
        // ---- stage the tile: PR rows of new samples, then the halo
#if !COMMS_DECIM_PREFETCH
        load_tile(t);
#endif
        DC_STAMP(0)  // global loads landed
        if (pre) {
            if constexpr (kRowRotorsPerTile) {
                cf l0 = lrow[0], h0 = lhalo[0];
                asm volatile("" : "+v"(l0), "+v"(h0));  // (opaque per tile: keeps the row rotors from being hoisted back out of the loop)
#pragma unroll
                for (int m = 0; m < PR; ++m) x[m] = cmulf(x[m], m ? cmulf(l0, to_cf(a.step[m])) : l0);
#pragma unroll
                for (int m = 0; m < HROWS; ++m) xh[m] = cmulf(xh[m], m ? cmulf(h0, to_cf(a.step[m])) : h0);
            } else {
#pragma unroll
                for (int m = 0; m < PR; ++m) x[m] = cmulf(x[m], lrow[m < NROW ? m : 0]);
#pragma unroll
                for (int m = 0; m < HROWS; ++m) xh[m] = cmulf(xh[m], lhalo[m < (kRowRotorsPerTile ? 1 : HROWS) ? m : 0]);
            }
        }
#pragma unroll
        for (int m = 0; m < PR; ++m) sh[kLinearSlots ? slot[0] + (WG / PR) * m : slot[m < NSLOT ? m : 0]] = x[m];
#pragma unroll
        for (int m = 0; m < HROWS; ++m)
            if (slot_h[m] >= 0) sh[slot_h[m]] = xh[m];
        DC_STAMP(1)  // mixer + LDS writes
        lds_barrier();
        DC_STAMP(2)
#if COMMS_DECIM_PREFETCH
        if (t + tstep < t1) load_tile(t + tstep);  // the next tile's rows travel while this tile's taps run
#endif

        // ---- outputs j = jb + OPL tid + c:  y_c = sum_k h[k] u[R j - k]
        // tile sample index of u[R j - k] is PR (tid + hlq) + (R c - k) = PR (tid + hlq - d) + p with
        // k = PR d + R c - p: block d, phase p descending = taps ascending; A[m] = h[m - (PR-1)], so that
        // with q = PR - 1 - p the tap of output c is A[PR d + q + R c].
        // Software pipeline over chunks of CH taps: the next chunk's LDS reads and tap loads are
        // issued before this chunk's FMAs.  The empty asm "uses" this chunk's registers first, so
        // the one s_waitcnt the compiler needs for them lands BEFORE the prefetch is issued (SMEM
        // returns out of order: any later wait would be lgkmcnt(0) and drain the prefetch too).
        cf acc[OPL];
#pragma unroll
        for (int c = 0; c < OPL; ++c) acc[c] = cf{0.f, 0.f};
        const cf* up = sh + tid + a.hlq;
        if constexpr (OPL == 4) {
            // Four outputs per lane: the taps of the four outputs that one LDS read feeds, A[m + R c] (c = 0..3), sit
            // together in a device-resident table, so a chunk of four reads needs ONE s_load_dwordx16 and 8 SGPR
            // pairs (the sliding window of the two-output form would need 14 and spill).  Real taps only.
            static_assert(REAL, "the four-output form is built for real taps");
            typedef const __attribute__((address_space(4))) float* cptr;  // constant address space: scalar loads
            const cptr qt = (cptr)(a.qt);
            constexpr int CH4 = PR % 16 == 0 ? 8 : 4, NCH4 = PR / CH4;  // reads per chunk (8: two s_load_dwordx16, 32 MACs of cover)
            static_assert(PR % CH4 == 0 && NCH4 % 2 == 0, "chunks must pair up inside a block (even R)");
            cf ua[CH4], ub[CH4];
            v2f ra[2 * CH4], rb[2 * CH4];
            auto fetch4 = [&](int d, int g, cf (&u)[CH4], v2f (&tr)[2 * CH4]) {
                const int m0 = PR * d + g * CH4;
#pragma unroll
                for (int i = 0; i < 2 * CH4; ++i) tr[i] = v2f{qt[4 * m0 + 2 * i], qt[4 * m0 + 2 * i + 1]};
#pragma unroll
                for (int i = 0; i < CH4; ++i) u[i] = up[(PR - 1 - (g * CH4 + i)) * S - d];
            };
            auto landed4 = [&](cf (&u)[CH4], v2f (&tr)[2 * CH4]) { asm volatile("" ::"v"(u[CH4 - 1]), "s"(tr[2 * CH4 - 1])); };
            auto macs4 = [&](const cf (&u)[CH4], const v2f (&tr)[2 * CH4]) {
#pragma unroll
                for (int i = 0; i < CH4; ++i) {
                    mac_s_lo(acc[0], u[i], tr[2 * i]);
                    mac_s_hi(acc[1], u[i], tr[2 * i]);
                    mac_s_lo(acc[2], u[i], tr[2 * i + 1]);
                    mac_s_hi(acc[3], u[i], tr[2 * i + 1]);
                }
            };
            fetch4(0, 0, ua, ra);
            for (int d = 0; d < a.nd; ++d) {
#pragma unroll
                for (int g = 0; g < NCH4; g += 2) {
                    landed4(ua, ra);
                    fetch4(d, g + 1, ub, rb);
                    macs4(ua, ra);
                    landed4(ub, rb);
                    if (g + 2 < NCH4)
                        fetch4(d, g + 2, ua, ra);
                    else if (d + 1 < a.nd)
                        fetch4(d + 1, 0, ua, ra);
                    macs4(ub, rb);
                }
            }
        } else {
        constexpr int CH = CHX ? CHX : OPL == 2 ? ((R <= 10 || R % 2) ? R : R / 2) : (R <= 4 ? R : R % 4 == 0 ? R / 2 : R <= 6 ? R : R / 2);  // taps per chunk (odd rates: whole blocks of R)
        constexpr int NCH = PR / CH;
        static_assert(PR % CH == 0 && (NCH == 1 || NCH % 2 == 0), "chunks must pair up inside a block (or be whole blocks)");
        constexpr int NP = ((CH & 1) + CH + (OPL - 1) * R + 1) / 2;  // SGPR pairs (A[2i], A[2i+1]) covering the chunk's taps of all OPL outputs (an odd chunk may start in a pair's hi half)
        cf ua[CH], ub[CH];
        v2f ra[NP], rb[NP], ia[NP], ib_[NP];
        // chunk g of block d: phases q = g*CH .. g*CH + CH - 1 (tap index ascending)
        auto fetch = [&](int d, int g, cf (&u)[CH], v2f (&tr)[NP], v2f (&ti)[NP]) {
            const int m0 = PR * d + ((g * CH) & ~1);
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                tr[i] = v2f{a.are[m0 + 2 * i], a.are[m0 + 2 * i + 1]};
                if (!REAL) ti[i] = v2f{a.aim[m0 + 2 * i], a.aim[m0 + 2 * i + 1]};
            }
#pragma unroll
            for (int i = 0; i < CH; ++i) u[i] = up[(PR - 1 - (g * CH + i)) * S - d];
        };
        auto landed = [&](cf (&u)[CH], v2f (&tr)[NP], v2f (&ti)[NP]) {
            asm volatile("" ::"v"(u[CH - 1]), "s"(tr[NP - 1]));
            if (!REAL) asm volatile("" ::"s"(ti[NP - 1]));
        };
        auto macs = [&](int g, const cf (&u)[CH], const v2f (&tr)[NP], const v2f (&ti)[NP]) {
            const int off = (g * CH) & 1;  // the chunk's first tap sits in the hi half of pair 0 when g*CH is odd
#pragma unroll
            for (int i = 0; i < CH; ++i) {
#pragma unroll
                for (int c = 0; c < OPL; ++c) {
                    const int e = off + i + R * c;
                    mac_tap<REAL>(acc[c], u[i], tr[e >> 1], ti[e >> 1], (e & 1) != 0);
                }
            }
        };
        fetch(0, 0, ua, ra, ia);
        if constexpr (NCH == 1) {  // whole blocks as chunks (diagnostic variant): blocks pair up, an odd last one runs alone
            int d = 0;
            for (; d + 1 < a.nd; d += 2) {
                landed(ua, ra, ia);
                fetch(d + 1, 0, ub, rb, ib_);
                macs(0, ua, ra, ia);
                landed(ub, rb, ib_);
                if (d + 2 < a.nd) fetch(d + 2, 0, ua, ra, ia);
                macs(0, ub, rb, ib_);
            }
            if (d < a.nd) {
                landed(ua, ra, ia);
                macs(0, ua, ra, ia);
            }
        } else
        for (int d = 0; d < a.nd; ++d) {
#pragma unroll
            for (int g = 0; g < NCH; g += 2) {
                landed(ua, ra, ia);
                fetch(d, g + 1, ub, rb, ib_);
                macs(g, ua, ra, ia);
                landed(ub, rb, ib_);
                if (g + 2 < NCH)
                    fetch(d, g + 2, ua, ra, ia);
                else if (d + 1 < a.nd)
                    fetch(d + 1, 0, ua, ra, ia);
                macs(g + 1, ub, rb, ib_);
            }
        }

        }

        // ---- epilogue: mixer after the FIR, FM demod, stores
        DC_STAMP(4)  // filter loop
        float2 y[OPL];
#pragma unroll
        for (int c = 0; c < OPL; ++c) y[c] = to_f2(acc[c]);
        if (pre) {
            const cf tt = cf{static_cast<float>(tt_c), static_cast<float>(tt_s)};
#pragma unroll
            for (int c = 0; c < OPL; ++c) y[c] = to_f2(cmulf(acc[c], tt));
            rotor_step(tt_c, tt_s, a.tile_c, a.tile_s);
        }
        if (post) {
            const cf ro = cf{static_cast<float>(ro_c), static_cast<float>(ro_s)};
#pragma unroll
            for (int c = 0; c < OPL; ++c) y[c] = to_f2(cmulf(acc[c], c ? cmulf(ro, to_cf(a.step_r[c])) : ro));
            rotor_step(ro_c, ro_s, a.tile_c, a.tile_s);
        }
        const long long j0 = jb + OPL * tid;
        if (fm) {
            if (j0 < 0) {  // FM.prev of the previous call stands in for y[-1]
                // (a scalar load: as a vector load into y[0] it made every later use of y[0] wait for vmcnt(0), i.e. for the
                // next tile's rows as well; another launch wrote the word, this one never does)
                typedef const __attribute__((address_space(4))) float* cfp;
                const cfp pp = (cfp)(a.fm_prev);
                y[0] = make_float2(pp[0], pp[1]);
            }
            if (l == 63) sh_y[w] = y[OPL - 1];
        }
        lds_barrier();  // sh_y visible; every lane is done reading the staged tile
        if (fm) {
            float2 p0 = make_float2(__shfl_up(y[OPL - 1].x, 1), __shfl_up(y[OPL - 1].y, 1));
            if (l == 0 && w > 0) p0 = sh_y[w - 1];
            float* o = static_cast<float*>(a.out);
            const long long n_out = static_cast<long long>(a.n_out);
            if (tid > 0 && j0 < n_out) o[j0] = fm_step_fast(y[0], p0);
#pragma unroll
            for (int c = 1; c < OPL; ++c)
                if (j0 + c < n_out) o[j0 + c] = fm_step_fast(y[c], y[c - 1]);
#pragma unroll
            for (int c = 0; c < OPL; ++c)
                if (j0 + c == n_out - 1) a.fm_prev_new[0] = y[c];
        } else {
            float2* o = static_cast<float2*>(a.out);
#pragma unroll
            for (int c = 0; c < OPL; ++c)
                if (j0 + c < static_cast<long long>(a.n_out)) o[j0 + c] = y[c];
        }
        DC_STAMP(5)  // barrier 2 + epilogue + stores
    }
    if (a.stamps && l == 0)
        for (int i = 0; i < 6; ++i) a.stamps[(static_cast<size_t>(blockIdx.x) * (WG / 64) + w) * 8 + i] = st_acc[i];
    kstamp_end(a.ks);
#undef DC_STAMP
}


// ---------------------------------------------------------------------------------------------------------------------
// The same chain, WAVE-PRIVATE (round 5): no workgroup barrier, no halo staged twice.
//
// What bounds fir_decim_kernel (profiles/r05_ab_chain_prefetch.txt, r05_pmc_config3.txt): nothing is saturated -- the
// vector ALU is busy 55-65 % of the time (at the 1.65 GHz the chip holds under this kernel), LDS 45 %, HBM 70 % of what a
// read stream of this shape reaches -- but its four waves meet at two barriers per tile, every tile stages its halo
// again, and a wave that waits for its rows has nothing else to do.  Requesting the next tile's rows early did not
// help there (the time moved from the load wait into the filter loop and the second barrier).
//
// Here a wave owns a RUN of consecutive tiles of 128 outputs (two per lane) and a private LDS image of PR = 2R phase
// arrays.  The 128 R new samples of a tile arrive as 2R rows of 64 (512-B loads); the halo -- the last 64 HR samples
// of the tile before -- is already in LDS: the wave kept those rows (raw) in registers and writes them, mixed for the
// new tile, into the halo slots once the filter loop has read the old ones.  A wave's LDS operations execute in order, so
// nothing but the data dependencies orders them: no barrier, no fence.  The next tile's rows are requested before the
// filter loop (the registers are free: the row rotors are re-made per tile from the lane's one rotor and the
// wave-uniform steps).  FM demod takes y[j-1] of a lane's first output from the lane below (DPP wave shift), lane 0
// from the tile before; the output before a run's first is made once per run by the whole wave from global memory
// (two taps per lane, summed across the lanes), so that every run is whole tiles and all waves end together.
// Same arithmetic per output as fir_decim_kernel (taps ascending, the same rotor products), except that run-start
// value, whose sum runs in another order (|d| ~ 1e-7 relative, one FM output per run).
//
// Built for PR | 64 (rates 4, 8, 16), Complex<f32> input; HR = 2 halo rows (taps <= 129) or 4 (<= 257).
template <int R, int HR>
struct DwGeom {
    static constexpr int PR = 2 * R;
    static constexpr int RPE = 64 / PR;        // phase-array elements per row of 64 samples
    static constexpr int SCAP = HR * RPE;      // halo elements in front of the 64 new ones of a phase array
    // phase-array stride.  A staging row is one ds_write_b64 per lane, served in groups of sixteen lanes over sixteen 8-byte bank
    // pairs; the group's lanes hold PR phases x 16 / PR consecutive elements, so the stride must be 16 / PR modulo 16 (rate 8, sixteen
    // phases: any odd stride; rate 4: 2 mod 16; rate 2: 4 mod 16 -- with an odd stride the rate-2 form spent 43 % of its LDS cycles
    // in bank conflicts, profiles/r05_pmc_rate2_wave.txt)
    static constexpr int SQ = 16 / PR > 0 ? 16 / PR : 1;
    static constexpr int S = PR >= 16 ? ((64 + SCAP) | 1) : (64 + SCAP + 15 - ((64 + SCAP + 15 - SQ) % 16));
    static constexpr int WAVE_CF = PR * S;
    static constexpr size_t LDS_WAVE = static_cast<size_t>(WAVE_CF) * sizeof(float2);
    static_assert(64 % PR == 0, "a row of 64 samples must be whole phase-array columns");
};


template <int R, bool REAL, bool PRE, int HR, int WPB, int AUX>
__global__ __launch_bounds__(64 * WPB, 4) void fir_decim_wave_kernel(const DecimArgs a) {
    using G = DwGeom<R, HR>;
    constexpr int PR = G::PR, S = G::S, RPE = G::RPE, SCAP = G::SCAP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int l = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));  // (wave-uniform, and the compiler must know: run, window and row offsets then live in SGPRs)
    cf* sh = reinterpret_cast<cf*>(smem) + w * G::WAVE_CF;
    const float2* in = static_cast<const float2*>(a.in);
    hist_advance(a.hist, in, a.n, a.new_hist, a.hist_len);
    kstamp_begin(a.ks);
    const bool post = !PRE && (a.mode & COMMS_CHAIN_POST) != 0;
    const bool fm = (a.mode & COMMS_CHAIN_FM) != 0;
    const long long n_out = static_cast<long long>(a.n_out);
    const long long C = 128LL * a.nt_chunk;  // outputs per run
    // (run k belongs to workgroup k: the sixteen waves that share a CU then work 32 MiB apart.  Giving them sixteen
    // ADJACENT runs instead -- b -> (b % 256) * 16 + b / 256, 2 MiB of the stream per CU -- was measured and is slower,
    // 130.6 against 124.6 us on config 3, as are four-wave workgroups with four adjacent runs: NOTES.md, round 5)
    const long long wave0 = static_cast<long long>(blockIdx.x) * WPB + w, n_waves = static_cast<long long>(gridDim.x) * WPB;

    // lane constants: LDS slot of the lane's sample of row 0 (row m: + RPE m; halo row i: - SCAP + RPE i), its rotor
    const int wslot = (l % PR) * S + SCAP + l / PR;
    cf l0 = cf{1.f, 0.f};
    if (PRE) {
        double c, sn;
        rotor_at(static_cast<uint64_t>(l) * a.frac, c, sn);
        l0 = cf{static_cast<float>(c), static_cast<float>(sn)};
    }
    const cf* up = sh + l + SCAP;
    const unsigned voff = static_cast<unsigned>(l) * 8u;

    for (long long k = wave0; k < a.n_chunks; k += n_waves) {
        const long long Jc = k * C;           // the run's first output
        const long long Bc = R * Jc;          // ... and its first new sample
        double tt_c = 1.0, tt_s = 0.0, ro_c = 1.0, ro_s = 0.0;
        if (PRE) rotor_at(a.turns0 + static_cast<uint64_t>(Bc) * a.frac, tt_c, tt_s);
        if (post) rotor_at(a.turns0 + static_cast<uint64_t>(R * (Jc + 2 * l)) * a.frac, ro_c, ro_s);
        // the run's window of the input as a buffer resource: reads past the stream's end return zero
        const long long w0 = Bc - 64 * HR;    // first halo sample (negative for the first run: history)
        const long long base_s = w0 < 0 ? 0 : w0;
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(in + base_s, (a.n - static_cast<size_t>(base_s)) * sizeof(float2));
        unsigned off = static_cast<unsigned>(Bc - base_s) * 8u;  // byte offset of the tile's row 0 in the window (wave-uniform)
        cf x[PR], xh[HR];
        if (w0 >= 0) {
#pragma unroll
            for (int i = 0; i < HR; ++i) xh[i] = to_cf(BufRows<const float2*>::get_from(rs, voff, 512u * i));
#pragma unroll
            for (int m = 0; m < PR; ++m) {
                const bv2u q = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, off + 512u * m, AUX & 2);
                x[m] = cf{__uint_as_float(q.x), __uint_as_float(q.y)};
            }
        } else {
#pragma unroll
            for (int i = 0; i < HR; ++i) xh[i] = to_cf(stream_at(in, a.hist, a.hist_len, w0 + 64 * i + l, a.n));
#pragma unroll
            for (int m = 0; m < PR; ++m) x[m] = to_cf(stream_at(in, a.hist, a.hist_len, Bc + 64 * m + l, a.n));
        }
#pragma unroll
        for (int i = 0; i < HR; ++i) sh[wslot - SCAP + RPE * i] = PRE ? cmulf(xh[i], cmulf_s(l0, to_cf(a.step_h[i]))) : xh[i];

        // FM: y[Jc - 1].  The first run: FM.prev of the call before; the others: the whole wave makes that one output
        // from global memory (lane l: taps l, l + 64, ...), mixed and rotated as the tile path does it
        cf carry = cf{0.f, 0.f};
        if (fm) {
            if (Jc == 0) {
                typedef const __attribute__((address_space(4))) float* cfp;
                const cfp pp = (cfp)(a.fm_prev);
                carry = cf{pp[0], pp[1]};
            } else {
                const long long P = Bc - R;  // newest sample of output Jc - 1
                cf part = cf{0.f, 0.f};
                for (int kk = l; kk < a.hist_len; kk += 64) {
                    cf xv = to_cf(in[P - kk]);
                    if (PRE) {
                        double c, sn;
                        rotor_at(a.turns0 + static_cast<uint64_t>(P - kk) * a.frac, c, sn);
                        xv = cmulf(xv, cf{static_cast<float>(c), static_cast<float>(sn)});
                    }
                    const float tr = a.are[kk + PR - 1];
                    part.x = __builtin_fmaf(tr, xv.x, part.x);
                    part.y = __builtin_fmaf(tr, xv.y, part.y);
                    if (!REAL) {
                        const float ti = a.aim[kk + PR - 1];
                        part.x = __builtin_fmaf(-ti, xv.y, part.x);
                        part.y = __builtin_fmaf(ti, xv.x, part.y);
                    }
                }
#pragma unroll
                for (int sft = 32; sft >= 1; sft >>= 1) {
                    part.x += __shfl_xor(part.x, sft);
                    part.y += __shfl_xor(part.y, sft);
                }
                if (post) {
                    double c, sn;
                    rotor_at(a.turns0 + static_cast<uint64_t>(R * (Jc - 1)) * a.frac, c, sn);
                    part = cmulf(part, cf{static_cast<float>(c), static_cast<float>(sn)});
                }
                carry = part;
            }
        }

        for (int i = 0; i < a.nt_chunk; ++i) {
            const long long J0 = Jc + 128LL * i;
            if (J0 >= n_out) break;
            // ---- stage the tile's rows (mixed on the way in); the last HR rows stay behind, raw, for the next tile's halo
#pragma unroll
            for (int m = 0; m < PR; ++m)
                sh[wslot + RPE * m] = PRE ? cmulf(x[m], m ? cmulf_s(l0, to_cf(a.step[m])) : l0) : x[m];
            cf keep[HR];
#pragma unroll
            for (int h = 0; h < HR; ++h) keep[h] = x[PR - HR + h];
            // ---- the next tile's rows travel while this tile's taps run
            off += 512u * PR;
            if (i + 1 < a.nt_chunk && J0 + 128 < n_out) {
#pragma unroll
                for (int m = 0; m < PR; ++m) {
                    const bv2u q = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, off + 512u * m, AUX & 2);
                    x[m] = cf{__uint_as_float(q.x), __uint_as_float(q.y)};
                }
            }
            // ---- outputs j = J0 + 2 l + c (as fir_decim_kernel's two-output loop: taps ascending, SGPR pairs, chunks of CH)
            cf acc[2] = {cf{0.f, 0.f}, cf{0.f, 0.f}};
            {
                constexpr int OPL = 2;
                constexpr int CH = (R <= 10 || R % 2) ? R : R / 2;
                constexpr int NCH = PR / CH;
                static_assert(PR % CH == 0 && NCH % 2 == 0, "chunks must pair up inside a block");
                constexpr int NP = ((CH & 1) + CH + (OPL - 1) * R + 1) / 2;
                cf ua[CH], ub[CH];
                v2f ra[NP], rb[NP], ia[NP], ib_[NP];
                auto fetch = [&](int d, int g, cf (&u)[CH], v2f (&tr)[NP], v2f (&ti)[NP]) __attribute__((always_inline)) {
                    const int m0 = PR * d + ((g * CH) & ~1);
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
                        tr[q] = v2f{a.are[m0 + 2 * q], a.are[m0 + 2 * q + 1]};
                        if (!REAL) ti[q] = v2f{a.aim[m0 + 2 * q], a.aim[m0 + 2 * q + 1]};
                    }
#pragma unroll
                    for (int q = 0; q < CH; ++q) u[q] = up[(PR - 1 - (g * CH + q)) * S - d];
                };
                auto landed = [&](cf (&u)[CH], v2f (&tr)[NP], v2f (&ti)[NP]) __attribute__((always_inline)) {
                    asm volatile("" ::"v"(u[CH - 1]), "s"(tr[NP - 1]));
                    if (!REAL) asm volatile("" ::"s"(ti[NP - 1]));
                };
                auto macs = [&](int g, const cf (&u)[CH], const v2f (&tr)[NP], const v2f (&ti)[NP]) __attribute__((always_inline)) {
                    const int o = (g * CH) & 1;
#pragma unroll
                    for (int q = 0; q < CH; ++q) {
#pragma unroll
                        for (int c = 0; c < OPL; ++c) {
                            const int e = o + q + R * c;
                            mac_tap<REAL>(acc[c], u[q], tr[e >> 1], ti[e >> 1], (e & 1) != 0);
                        }
                    }
                };
                fetch(0, 0, ua, ra, ia);
                for (int d = 0; d < a.nd; ++d) {
#pragma unroll
                    for (int g = 0; g < NCH; g += 2) {
                        landed(ua, ra, ia);
                        fetch(d, g + 1, ub, rb, ib_);
                        macs(g, ua, ra, ia);
                        landed(ub, rb, ib_);
                        if (g + 2 < NCH)
                            fetch(d, g + 2, ua, ra, ia);
                        else if (d + 1 < a.nd)
                            fetch(d + 1, 0, ua, ra, ia);
                        macs(g + 1, ub, rb, ib_);
                    }
                }
            }
            // ---- the next tile's halo: this tile's last rows, mixed for the NEW tile (the filter loop's reads of the old
            // halo were issued before these writes: a wave's LDS operations execute in order)
#pragma unroll
            for (int h = 0; h < HR; ++h) sh[wslot - SCAP + RPE * h] = PRE ? cmulf(keep[h], cmulf_s(l0, to_cf(a.step_h[h]))) : keep[h];
            // ---- epilogue: mixer after the FIR, FM demod, stores
            cf y0 = acc[0], y1 = acc[1];
            if (PRE) {
                const cf tt = cf{static_cast<float>(tt_c), static_cast<float>(tt_s)};
                y0 = cmulf(acc[0], tt);
                y1 = cmulf(acc[1], tt);
                rotor_step(tt_c, tt_s, a.tile_c, a.tile_s);
            }
            if (post) {
                const cf ro = cf{static_cast<float>(ro_c), static_cast<float>(ro_s)};
                y0 = cmulf(acc[0], ro);
                y1 = cmulf(acc[1], cmulf(ro, to_cf(a.step_r[1])));
                rotor_step(ro_c, ro_s, a.tile_c, a.tile_s);
            }
            const long long j0 = J0 + 2 * l;
            const bool whole = J0 + 128 <= n_out;  // (wave-uniform)
            if (fm) {
                const cf p0 = cf{wave_shr1(y1.x, carry.x), wave_shr1(y1.y, carry.y)};
                // (two __builtin_amdgcn_readlane of y1.x / y1.y came out of this compiler as ONE v_readlane of y1.x used
                // for both halves -- seen in the ISA, and as wrong angles at every tile's first output; asm keeps them apart)
                int cx, cy;
                asm volatile("s_nop 1\n\tv_readlane_b32 %0, %2, 63\n\tv_readlane_b32 %1, %3, 63" : "=s"(cx), "=s"(cy) : "v"(y1.x), "v"(y1.y));
                carry = cf{__builtin_bit_cast(float, cx), __builtin_bit_cast(float, cy)};
                const float o0 = fm_step_fast(to_f2(y0), to_f2(p0)), o1 = fm_step_fast(to_f2(y1), to_f2(y0));
                float* o = static_cast<float*>(a.out);
                if (whole) {  // (j0 is even and the host checked the buffer's alignment; a plain store, so that the compiler's
                              // vmcnt arithmetic sees it: behind an asm store it waited for the store's completion as well)
                    if (AUX & 4) {
                        typedef float nt_f2 __attribute__((ext_vector_type(2)));
                        __builtin_nontemporal_store(nt_f2{o0, o1}, reinterpret_cast<nt_f2*>(o + j0));
                    } else
                        *reinterpret_cast<float2*>(o + j0) = make_float2(o0, o1);
                } else {
                    if (j0 < n_out) o[j0] = o0;
                    if (j0 + 1 < n_out) o[j0 + 1] = o1;
                }
                if (j0 == n_out - 1) a.fm_prev_new[0] = to_f2(y0);
                if (j0 + 1 == n_out - 1) a.fm_prev_new[0] = to_f2(y1);
            } else {
                float2* o = static_cast<float2*>(a.out);
                if (whole)
                    *reinterpret_cast<float4*>(o + j0) = make_float4(y0.x, y0.y, y1.x, y1.y);
                else {
                    if (j0 < n_out) o[j0] = to_f2(y0);
                    if (j0 + 1 < n_out) o[j0 + 1] = to_f2(y1);
                }
            }
        }
    }
    kstamp_end(a.ks);
}


// Event pair of an attached kernel timer for the launch about to be made (set by comms_fir_run_decim_dev, taken by
// launch_decim_v): the kernel's own begin / end timestamps, as the FIR kernels' timed launches -- events recorded
// around the launch would include the dispatch gap in front of it whenever the launch before it carried no events.
static thread_local hipEvent_t g_decim_ev_start = nullptr, g_decim_ev_stop = nullptr;

template <int R, bool REAL, bool PRE, int HR, int WPB, int AUX>
static comms_status_t launch_decim_wave_v(const DecimArgs& a, hipStream_t s) {
    using G = DwGeom<R, HR>;
    // (single-wave workgroups ask for a sixteenth of the CU's LDS: seventeen would fit at HR = 2, and the dispatcher
    // would then fill some CUs with 17 waves and leave others 15)
    constexpr size_t lds = WPB == 1 && G::LDS_WAVE < 10240 ? 10240 : G::LDS_WAVE * WPB;
    const size_t want = (static_cast<size_t>(a.n_chunks) + WPB - 1) / WPB;
    const size_t slots = (160u * 1024u / lds) * kNumCU;
    const unsigned blocks = static_cast<unsigned>(want < slots ? want : slots);
    static DeviceOnce attr_once;
    if (attr_once.need())
        COMMS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fir_decim_wave_kernel<R, REAL, PRE, HR, WPB, AUX>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipEvent_t ea = g_decim_ev_start, eb = g_decim_ev_stop;
    g_decim_ev_start = g_decim_ev_stop = nullptr;
    if (ea)
        hipExtLaunchKernelGGL((fir_decim_wave_kernel<R, REAL, PRE, HR, WPB, AUX>), dim3(blocks), dim3(64 * WPB), static_cast<uint32_t>(lds), s, ea, eb, 0u, a);
    else
        fir_decim_wave_kernel<R, REAL, PRE, HR, WPB, AUX><<<dim3(blocks), dim3(64 * WPB), lds, s>>>(a);
    return launch_ok("fir_decim_wave_kernel");
}

template <int R, int OPL, bool REAL, bool PRE, int TILE = DC_TILE, int CHX = 0>
static comms_status_t launch_decim_v(const DecimArgs& a, hipStream_t s) {
    using G = DcGeom<R, OPL, TILE>;
    constexpr size_t lds = G::LDS;
    // persistent grid: one workgroup per slot of the chip; tile b, b + slots, ... (or a contiguous run) each
    const size_t slots = static_cast<size_t>(G::template wgpc<PRE>()) * kNumCU;
    const unsigned blocks = static_cast<unsigned>(a.n_tiles < slots ? a.n_tiles : slots);
    static DeviceOnce attr_once;
    if (attr_once.need())
        COMMS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fir_decim_kernel<R, OPL, REAL, PRE, TILE, CHX>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    DecimArgs b = a;
    if (a.interleave) {
        // a workgroup's next tile is `blocks` tiles on: the per-step rotor of its tile-wide phase follows the grid
        const uint64_t ts = static_cast<uint64_t>(TILE) - ((a.mode & COMMS_CHAIN_FM) ? 1 : 0);
        mix_host_rotor(static_cast<uint64_t>(R) * ts * blocks * a.frac, b.tile_c, b.tile_s);
    }
    hipEvent_t ea = g_decim_ev_start, eb = g_decim_ev_stop;
    g_decim_ev_start = g_decim_ev_stop = nullptr;
    if (ea)
        hipExtLaunchKernelGGL((fir_decim_kernel<R, OPL, REAL, PRE, TILE, CHX>), dim3(blocks), dim3(G::WG), static_cast<uint32_t>(lds), s, ea, eb, 0u, b);
    else
        fir_decim_kernel<R, OPL, REAL, PRE, TILE, CHX><<<dim3(blocks), dim3(G::WG), lds, s>>>(b);
    return launch_ok("fir_decim_kernel");
}

// Outputs per lane.  Two in the product.  Four (half the LDS reads per MAC, half the waves per CU, 28 tap
// registers per chunk instead of 16) was built and measured: slower wherever the chain kernel is chosen
// (metric chain 50 -> 60 us, config 3 153 -> 172 us) and ahead only beyond ~100 MACs per input sample
// (255 taps / 2: 194 -> 148 us), where the overlap-save fusion is used anyway.  The diagnostic build keeps
// it selectable (COMMS_DECIM_OPL=4) so that the comparison can be repeated.
static int decim_opl(bool real, int macs_per_input) {
    (void)macs_per_input;
#ifdef COMMS_DIAG
    static const int forced = diag_knob("COMMS_DECIM_OPL", 0);
    if (real && forced == 4) return 4;
#else
    (void)real;
#endif
    return 2;
}

// Tile width (diagnostic build: COMMS_DECIM_TILE=1024 = 512-lane workgroups, two per CU, so that the waves a
// SIMD holds are in the same phase more often)
static int decim_tile(bool real, int R, int opl) {
#ifdef COMMS_DIAG
    static const int forced = diag_knob("COMMS_DECIM_TILE", COMMS_DECIM_TILE_DEFAULT);
    if (real && opl == 2 && R == 8 && (forced == 1024 || forced == 256)) return forced;
#else
    (void)real; (void)R; (void)opl;
#endif
    return DC_TILE;
}

template <int R>
static comms_status_t launch_decim(const DecimArgs& a, bool real, int opl, int tile, hipStream_t s) {
    const bool pre = (a.mode & COMMS_CHAIN_PRE) != 0;
#ifdef COMMS_DIAG
    if constexpr (R == 8) {
        static const int chx = diag_knob("COMMS_DECIM_CH", 0);
        if (chx == 16 && real && opl == 2 && !pre && tile == DC_TILE) return launch_decim_v<R, 2, true, false, DC_TILE, 16>(a, s);
        if (tile == 1024) return pre ? launch_decim_v<R, 2, true, true, 1024>(a, s) : launch_decim_v<R, 2, true, false, 1024>(a, s);
        if (tile == 256) return pre ? launch_decim_v<R, 2, true, true, 256>(a, s) : launch_decim_v<R, 2, true, false, 256>(a, s);
    }
#else
    (void)tile;
#endif
    if (!real) return pre ? launch_decim_v<R, 2, false, true>(a, s) : launch_decim_v<R, 2, false, false>(a, s);
#ifdef COMMS_DIAG
    if constexpr (R <= 8 && R % 2 == 0) {
        if (opl == 4) return pre ? launch_decim_v<R, 4, true, true>(a, s) : launch_decim_v<R, 4, true, false>(a, s);
    }
#else
    (void)opl;
#endif
    return pre ? launch_decim_v<R, 2, true, true>(a, s) : launch_decim_v<R, 2, true, false>(a, s);
}

// odd rates above 10 (chunks of R taps): real taps only
template <int R>
static comms_status_t launch_decim_real(const DecimArgs& a, hipStream_t s) {
    return (a.mode & COMMS_CHAIN_PRE) ? launch_decim_v<R, 2, true, true>(a, s) : launch_decim_v<R, 2, true, false>(a, s);
}

}  // namespace comms

using namespace comms;

static unsigned long long* g_decim_stamps = nullptr;

extern "C" {

#ifdef COMMS_DIAG
// diagnostic hook (diagnostic build only): device buffer of 8 u64 per wave, or NULL
void comms_debug_decim_stamps(void* d_buf) { g_decim_stamps = static_cast<unsigned long long*>(d_buf); }
#endif

// MACs per input sample of (taps, rate) on the decimating kernel, or -1 when it cannot run there (no
// instantiation for the rate, or taps beyond the kernel-argument budget).
static int32_t decim_macs(const comms_fir_t* h, uint32_t rate) {
    if (!h || h->n_eff < 1 || h->n_eff > DC_NMAX) return -1;
    switch (rate) {
        case 2: case 3: case 4: case 5: case 6: case 7: case 8: case 9: case 10: case 12: case 14: case 16: break;
        case 11: case 13: case 15:  // chunks of R taps: the SGPR pairs of complex taps do not fit
            if (!h->real_taps) return -1;
            break;
        default: return -1;
    }
    return (h->n_eff + static_cast<int>(rate) - 1) / static_cast<int>(rate) * (h->real_taps ? 1 : 2);
}
// Whether (taps, rate) runs on the decimating kernel: 0 = no, 1 = it can, 2 = and with few enough MACs per
// input sample that it beats the alternative (measured at 2^24 samples, scripts under gpurun / DESIGN.md section 3):
//   * the fused overlap-save launch, where that exists (up to 257 taps): 67 us without FM demod (crossover 44 MACs),
//     60-68 us + the demodulator's own small kernel with it (127 taps / 3: 42 MACs, 63 against 68 us; 255 / 5: 51
//     MACs, 66 against ~65 -> 48);
//   * the four kernels in series, where it does not: 135-170 us, against 101 us at 85 MACs (255 taps / 3) and
//     186 us at 128 (255 / 2) -> 96.
int32_t comms_fir_decim_supported_for(const comms_fir_t* h, uint32_t rate, int32_t fm_demod, int32_t can_fuse) {
    const int macs = decim_macs(h, rate);
    if (macs < 0) return 0;
    static const int max_macs = diag_knob("COMMS_DECIM_MAX_MACS", 0);
    // (rate 2 computes every other output: its filter loop costs 12 us per 16 taps at 2^24 samples, and the overlap-save fusion
    // (57 us, 74 with the demodulator behind it) is level with it at 63 / 79 real taps -- scripts/ab_rate2_tf.py, the two taking turns:
    // 63 taps 56.1 against 57.9, 79 taps 63.4 against 57.5; FM chains 79 taps 71.7 against 74.4, 95 taps 80.7 against 74.0)
    const int limit = max_macs > 0 ? max_macs : !can_fuse ? 96 : rate == 2 ? (fm_demod ? 40 : 32) : fm_demod ? 48 : 44;
    return macs <= limit ? 2 : 1;
}
int32_t comms_fir_decim_supported(const comms_fir_t* h, uint32_t rate) {
    return comms_fir_decim_supported_for(h, rate, 0, 1);
}

comms_status_t comms_fir_run_decim_dev(comms_fir_t* h, const void* d_in, size_t n, void* d_out, int32_t mode,
                                       uint64_t turns0, uint64_t frac, uint32_t rate, const void* fm_prev,
                                       void* fm_prev_new, void* stream) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG((d_in && d_out) || !n, "NULL device pointer");
    COMMS_ARG(comms_fir_decim_supported(h, rate) != 0,
              "the decimating chain kernel supports <= 257 taps and rates 2 ... 16 (11, 13, 15: real taps)");
    COMMS_ARG(n % rate == 0, "n must be a multiple of the decimation rate");
    COMMS_ARG((mode & COMMS_CHAIN_DEC) && !((mode & COMMS_CHAIN_PRE) && (mode & COMMS_CHAIN_POST)), "bad chain mode");
    COMMS_TRY(fir_check_sticky(h));
    COMMS_TRY(use_device(h->device));
    if (!n) return COMMS_OK;
    // long filters on long batches at rate 8: the polyphase frequency-domain kernel (fir_poly8.hip)
    if (comms_fir_poly8_supported(h, rate, mode, n) == 2)
        return comms_fir_run_poly8_dev(h, d_in, n, d_out, mode, turns0, frac, rate, fm_prev, fm_prev_new, stream);
    h->last_poly8 = false;
    const size_t in_elem = h->in_fmt == COMMS_IQ_I16 ? 4 : h->in_fmt == COMMS_IQ_U8 ? 2 : 8;
    COMMS_ARG(!ranges_overlap(d_in, n * in_elem, d_out, (n / rate) * ((mode & COMMS_CHAIN_FM) ? 4 : 8)),
              "the decimating chain cannot run in place");
    COMMS_ARG((reinterpret_cast<uintptr_t>(d_in) & (in_elem - 1)) == 0, "input must be aligned to one IQ sample");
    hipStream_t s = nullptr;
    COMMS_TRY(h->enter(stream, &s));
    const bool real = h->real_taps;
    const int R = static_cast<int>(rate), N = h->n_eff;
    const int opl = (R <= 8 && R % 2 == 0) ? decim_opl(real, (N + R - 1) / R) : 2;  // (four outputs per lane: even R up to 8)
    const int tile = decim_tile(real, R, opl);
    const int PR = opl * R, WG = tile / opl;
    DecimArgs a{};
    a.in = d_in;
    a.fmt = h->in_fmt;
    a.in_scale = h->in_scale;
    a.hist = h->d_hist[h->cur];
    a.new_hist = h->d_hist[h->cur ^ 1];
    a.out = d_out;
    a.fm_prev = static_cast<const float2*>(fm_prev);
    a.fm_prev_new = static_cast<float2*>(fm_prev_new);
    a.n = n;
    a.n_out = n / rate;
    const bool fm = (mode & COMMS_CHAIN_FM) != 0;
    const size_t ts = static_cast<size_t>(tile) - (fm ? 1 : 0);
    a.n_tiles = (a.n_out + ts - 1) / ts;
    a.hist_len = h->n_eff;
    a.hlq = (N - 1 + PR - 1) / PR;
    a.nd = a.hlq + 1;
    a.mode = mode;
    a.turns0 = turns0;
    a.frac = frac;
    a.stamps = g_decim_stamps;
#ifdef COMMS_DIAG
    static const int nomac = diag_knob("COMMS_DECIM_DEBUG_NOMAC", COMMS_DECIM_NOMAC_DEFAULT);
    if (nomac) a.nd = 0;  // diagnostic build only: staging + epilogue without the filter loop (wrong results; timing)
#endif
    // tile order: interleaved over the workgroups by default (config 3 at 2^26: 145 -> 142 us per step, neutral
    // at 2^24); COMMS_DECIM_INTERLEAVE=0 gives every workgroup a contiguous run again
    static const int interleave = diag_knob("COMMS_DECIM_INTERLEAVE", 1);
    a.interleave = interleave;
    mix_host_rotor(static_cast<uint64_t>(R) * ts * frac, a.tile_c, a.tile_s);
    double c, sn;
    for (int k = 0; k < opl; ++k) {
        mix_host_rotor(static_cast<uint64_t>(R * k) * frac, c, sn);
        a.step_r[k] = make_float2(static_cast<float>(c), static_cast<float>(sn));
    }
    for (int m = 0; m < PR; ++m) {
        mix_host_rotor(static_cast<uint64_t>(WG * m) * frac, c, sn);
        a.step[m] = make_float2(static_cast<float>(c), static_cast<float>(sn));
    }
    COMMS_ARG(PR * (a.hlq + 1) + (opl - 1) * R + 4 <= DC_AMAX, "tap table overflow");
    if (opl == 4) {  // tap quadruples Q[m][c] = A[m + R c], A[m] = h[m - (PR-1)]: built once per rate, device resident
        if (h->qt_rate != R) {
            const int rows = PR * (a.hlq + 1);
            std::vector<float> q(static_cast<size_t>(rows) * 4, 0.f);
            for (int m = 0; m < rows; ++m)
                for (int c = 0; c < 4; ++c) {
                    const int k = m + R * c - (PR - 1);
                    if (k >= 0 && k < N) q[4 * m + c] = h->taps[k].re;
                }
            COMMS_TRY(h->quiesce());
            h->last_stream = s;  // (quiesce forgot the stream `enter` just recorded: the launch below must stay tracked)
            h->launched = true;
            if (h->d_qt) (void)hipFree(h->d_qt);
            h->d_qt = nullptr;
            COMMS_HIP_TRY(hipMalloc(&h->d_qt, q.size() * sizeof(float)));
            COMMS_HIP_TRY(hipMemcpy(h->d_qt, q.data(), q.size() * sizeof(float), hipMemcpyHostToDevice));
            h->qt_rate = R;
        }
        a.qt = h->d_qt;
    }
    for (int m = 0; m < DC_AMAX; ++m) {
        const int k = m - (PR - 1);
        const bool in_range = k >= 0 && k < N;
        a.are[m] = in_range ? h->taps[k].re : 0.f;
        a.aim[m] = in_range ? h->taps[k].im : 0.f;
    }
    // (the pair travels to launch_decim_v through thread-local slots: whatever the exit, none stays behind for the next
    // launch on this thread, possibly another handle's)
    struct EvGuard {
        ~EvGuard() { g_decim_ev_start = g_decim_ev_stop = nullptr; }
    } ev_guard;
    (void)h->take_events(g_decim_ev_start, g_decim_ev_stop);
    a.ks = h->next_stamp();
    comms_status_t st;
    // The wave-private form (fir_decim_wave_kernel): rate 8, Complex<f32> input, and a batch whose tiles of 128 outputs
    // spread evenly over the waves of the chip (every wave runs ONE contiguous run: the launch lasts as long as its
    // longest run, so a batch of 4.3 tiles per wave would pay for 5)
    static const int wave_knob = diag_knob("COMMS_DECIM_WAVE", COMMS_DECIM_WAVE_DEFAULT);
    // Up to 129 taps (two halo rows).  Longer filters stay on the workgroup kernel: with four halo rows a wave's LDS image is
    // 10.4 KB, fifteen waves per CU instead of sixteen, and the 255-tap chain then runs 164 -> 187 us at 2^26 samples and
    // 42.9 -> 48.8 us at 2^24 (scripts/ab_chain.py, CASES=c2,c2c,c2b with COMMS_DECIM_WAVE=2; NOTES.md round 5).
    // Rates 2 and 4 too (round 5, second half; COMMS_DECIM_WAVE_R24=0 in the diagnostic build: the workgroup kernel as before) -- 2^24
    // samples, 15 / 31 / 47 / 63 real taps: rate 2 41.1 / 50.4 / 55.3 / 61.8 -> 35.1 / 47.0 / 51.6 / 59.2 us, rate 4 level up to 47 taps,
    // 35.1 -> 32.3 at 63 (scripts/time_rate2.py)
    static const int wave_r24 = diag_knob("COMMS_DECIM_WAVE_R24", 1);
    if (wave_knob && (R == 8 || (wave_r24 && (R == 2 || R == 4))) && h->in_fmt == COMMS_IQ_C32 && opl == 2 && tile == DC_TILE && a.hlq * PR <= 128 &&
        (reinterpret_cast<uintptr_t>(d_out) & 15) == 0) {  // (its stores are 8 / 16 bytes per lane)
        constexpr int HR = 2;
        const long long tiles = static_cast<long long>((a.n_out + 127) / 128);
        const long long waves = 16 * static_cast<long long>(kNumCU);  // single-wave workgroups, 10240 B of LDS each
        const long long nt = (tiles + waves - 1) / waves;
        const bool balanced = tiles >= waves && nt * waves * 100 <= tiles * 104;
        if (balanced || wave_knob == 2) {
            a.nt_chunk = static_cast<int>(nt < 1 ? 1 : nt);
            static const int ntc_div = diag_knob("COMMS_DECIM_WAVE_SPLIT", COMMS_DECIM_WAVE_SPLIT_DEFAULT);  // runs per wave (trial)
            if (ntc_div > 1 && a.nt_chunk % ntc_div == 0) a.nt_chunk /= ntc_div;
            a.n_chunks = (tiles + a.nt_chunk - 1) / a.nt_chunk;
            a.interleave = 0;
            mix_host_rotor(static_cast<uint64_t>(R) * 128u * frac, a.tile_c, a.tile_s);
            for (int m = 0; m < PR; ++m) {
                mix_host_rotor(static_cast<uint64_t>(64 * m) * frac, c, sn);
                a.step[m] = make_float2(static_cast<float>(c), static_cast<float>(sn));
            }
            for (int i = 0; i < HR; ++i) {
                mix_host_rotor(static_cast<uint64_t>(static_cast<long long>(64 * (i - HR))) * frac, c, sn);
                a.step_h[i] = make_float2(static_cast<float>(c), static_cast<float>(sn));
            }
            const bool pre = (mode & COMMS_CHAIN_PRE) != 0;
            // nontemporal loads and stores once the batch is past what the 256 MiB Infinity Cache can hold for the next kernel
            // (config 3 at 2^26 samples: 128.3 -> 120.3 us; at 2^24 the re-read input of a benchmark loop would lose its
            // cache hits); the diagnostic build's COMMS_DECIM_WAVE_NT = 0 / 2 (loads) / 4 (stores) / 6 forces a form
            static const int nt_knob = diag_knob("COMMS_DECIM_WAVE_NT", COMMS_DECIM_WAVE_NT_DEFAULT);
            const int nt_loads = nt_knob >= 0 ? nt_knob : (n * sizeof(float2) > (192u << 20) ? 6 : 0);
#ifdef COMMS_DIAG
            static const int wpb4 = diag_knob("COMMS_DECIM_WAVE_WPB", COMMS_DECIM_WAVE_WPB_DEFAULT) == 4;
#define COMMS_DW(REAL_, PRE_, HR_) (wpb4 ? launch_decim_wave_v<8, REAL_, PRE_, HR_, 4, 0>(a, s) : nt_loads == 2 ? launch_decim_wave_v<8, REAL_, PRE_, HR_, 1, 2>(a, s) : nt_loads == 4 ? launch_decim_wave_v<8, REAL_, PRE_, HR_, 1, 4>(a, s) : nt_loads == 6 ? launch_decim_wave_v<8, REAL_, PRE_, HR_, 1, 6>(a, s) : launch_decim_wave_v<8, REAL_, PRE_, HR_, 1, 0>(a, s))
#else
#define COMMS_DW(REAL_, PRE_, HR_) (nt_loads ? launch_decim_wave_v<8, REAL_, PRE_, HR_, 1, 6>(a, s) : launch_decim_wave_v<8, REAL_, PRE_, HR_, 1, 0>(a, s))
#endif
#define COMMS_DWR(RV, REAL_, PRE_) (nt_loads ? launch_decim_wave_v<RV, REAL_, PRE_, 2, 1, 6>(a, s) : launch_decim_wave_v<RV, REAL_, PRE_, 2, 1, 0>(a, s))
            if (R == 2)
                st = real ? (pre ? COMMS_DWR(2, true, true) : COMMS_DWR(2, true, false)) : (pre ? COMMS_DWR(2, false, true) : COMMS_DWR(2, false, false));
            else if (R == 4)
                st = real ? (pre ? COMMS_DWR(4, true, true) : COMMS_DWR(4, true, false)) : (pre ? COMMS_DWR(4, false, true) : COMMS_DWR(4, false, false));
            else
                st = real ? (pre ? COMMS_DW(true, true, 2) : COMMS_DW(true, false, 2)) : (pre ? COMMS_DW(false, true, 2) : COMMS_DW(false, false, 2));
#undef COMMS_DWR
#undef COMMS_DW
            COMMS_TRY(st);
            h->cur ^= 1;
            return COMMS_OK;
        }
    }
    switch (R) {
        case 2: st = launch_decim<2>(a, real, opl, tile, s); break;
        case 3: st = launch_decim<3>(a, real, opl, tile, s); break;
        case 4: st = launch_decim<4>(a, real, opl, tile, s); break;
        case 5: st = launch_decim<5>(a, real, opl, tile, s); break;
        case 6: st = launch_decim<6>(a, real, opl, tile, s); break;
        case 7: st = launch_decim<7>(a, real, 2, tile, s); break;
        case 8: st = launch_decim<8>(a, real, opl, tile, s); break;
        case 9: st = launch_decim<9>(a, real, 2, tile, s); break;
        case 10: st = launch_decim<10>(a, real, 2, tile, s); break;
        case 11: st = launch_decim_real<11>(a, s); break;
        case 12: st = launch_decim<12>(a, real, 2, tile, s); break;
        case 13: st = launch_decim_real<13>(a, s); break;
        case 14: st = launch_decim<14>(a, real, 2, tile, s); break;
        case 15: st = launch_decim_real<15>(a, s); break;
        case 16: st = launch_decim<16>(a, real, 2, tile, s); break;
        default: return fail(COMMS_ERR_ARG, "no decimating kernel for rate %d", R);
    }
    COMMS_TRY(st);
    h->cur ^= 1;
    return COMMS_OK;
}

}  // extern "C"
