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
#include <cmath>

#include "common.hpp"
#include "fft_radix.hpp"
#include "fir_handle.hpp"
#include "sgpr_mac.hpp"

namespace comms {

constexpr int DC_WG = 256;          // lanes per workgroup
constexpr int DC_TILE = 2 * DC_WG;  // outputs per tile (two per lane)
constexpr int DC_NMAX = 257;        // taps (kernel-argument budget)
constexpr int DC_AMAX = 356;        // padded tap array: 2R*nd + R + 4 <= 256 + 63 + 16 + 4 (+ pair slack)
constexpr int DC_RMAX = 16;

template <int R>
struct DcGeom {
    static constexpr int PR = 2 * R;                               // phases
    static constexpr int HLQ_MAX = (DC_NMAX - 1 + PR - 1) / PR;    // halo in phase-array elements
    static constexpr int S = (DC_WG + HLQ_MAX + 1) | 1;            // phase-array stride (odd: staging writes spread over the banks)
    static constexpr size_t LDS = static_cast<size_t>(PR) * S * sizeof(float2);
    static constexpr int WGPC = R <= 8 ? 4 : R <= 12 ? 3 : 2;     // workgroups per CU that fit in LDS (and set the VGPR budget)
};

struct DecimArgs {
    const float2* in;
    const float2* hist;
    float2* new_hist;
    void* out;
    const float2* fm_prev;
    float2* fm_prev_new;
    size_t n, n_out, n_tiles;
    int hist_len, hlq, nd, mode;   // hlq = ceil((N-1)/2R); nd = hlq + 1 tap blocks of 2R
    uint64_t turns0, frac;         // mixer phase of input sample 0 and per-sample increment (turns)
    double tile_c, tile_s;         // e^{i * R * tile_step * dphi}
    float2 step_r;                 // e^{i * R * dphi}  (second output of a lane, mixer-after-FIR)
    float2 step[2 * DC_RMAX];      // e^{i * 256 m * dphi}, staging row m
    float are[DC_AMAX];            // A[m] = Re h[m - (2R-1)], zero outside [0, N)
    float aim[DC_AMAX];
    unsigned long long* stamps;    // diagnostic (scripts/stamp_decim.py): per-wave cycles per phase, or NULL
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

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, which would
// stall on the next tile's global loads that are deliberately left in flight across it.
__device__ __forceinline__ void lds_barrier() {  // (kept light: nothing global is shared between waves here)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// PRE: the mixer sits in front of the FIR (samples are mixed on their way into LDS); otherwise it
// follows the FIR (or is absent).  PF: the next tile's global loads are issued right after this
// tile has been staged, so that they fly during the filter loop (costs the 2R + 2 staging
// registers across the loop; the PRE form then derives its row rotors on the fly instead of
// keeping 2R of them in VGPRs).
template <int R, bool REAL, bool PRE, bool PF>
__global__ __launch_bounds__(DC_WG, DcGeom<R>::WGPC) void fir_decim_kernel(const DecimArgs a) {
    using G = DcGeom<R>;
    constexpr int PR = G::PR, S = G::S;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* sh = reinterpret_cast<cf*>(smem);  // [PR][S]
    __shared__ float2 sh_y[DC_WG / 64];
    hist_advance(a.hist, a.in, a.n, a.new_hist, a.hist_len);

    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    constexpr bool pre = PRE;
    const bool post = !PRE && (a.mode & COMMS_CHAIN_POST) != 0;
    const bool fm = (a.mode & COMMS_CHAIN_FM) != 0;
    const int ovl = fm ? 1 : 0;           // FM tiles recompute the previous tile's last output
    const long long ts = DC_TILE - ovl;   // stored outputs per tile
    const int hl = a.hlq * PR;            // halo samples staged to the left of the tile

    const size_t t0 = static_cast<size_t>(blockIdx.x) * a.n_tiles / gridDim.x;
    const size_t t1 = static_cast<size_t>(blockIdx.x + 1) * a.n_tiles / gridDim.x;
    if (t0 >= t1) return;

    // Mixer rotors.  The phase of input sample i = ib + tid + 256 m of a tile splits into a part
    // that is the same for the whole tile, T = rot(ib) (f64, stepped per tile, applied to the
    // two outputs after the filter -- the filter is linear), and a part that never changes,
    // lrow[m] = e^{i (tid + 256 m) dphi} (f32, set up once): one multiply per staged sample.
    //   mixer after the FIR: ro = rot(R (jb + 2 tid)) is this lane's first output's rotor.
    constexpr int NROW = PRE && !PF ? PR : 1;
    cf lrow[NROW], lhalo[2];
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
            for (int m = 0; m < 2; ++m) lhalo[m] = m ? cmulf(h0, to_cf(a.step[m])) : h0;
            rotor_at(a.turns0 + static_cast<uint64_t>(R * jb0) * a.frac, tt_c, tt_s);
        }
        if (post) rotor_at(a.turns0 + static_cast<uint64_t>(R * (jb0 + 2 * tid)) * a.frac, ro_c, ro_s);
    }
    // LDS slots of this lane's staged samples (sample hl + tid + 256 m of the tile; halo: tid + 256 m)
    int slot[PR], slot_h[2];
#pragma unroll
    for (int m = 0; m < PR; ++m) {
        const unsigned s = static_cast<unsigned>(tid + 256 * m);
        slot[m] = static_cast<int>((s % PR) * S + a.hlq + s / PR);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const unsigned s = static_cast<unsigned>(tid + 256 * m);
        slot_h[m] = s < static_cast<unsigned>(hl) ? static_cast<int>((s % PR) * S + s / PR) : -1;
    }

    // The tile's samples travel global -> VGPRs -> (mixer) -> LDS.  (Requesting the next tile's
    // rows before this tile's filter loop was measured and does not pay: it costs 36 VGPRs, i.e.
    // either the persistent row rotors or the fourth workgroup per CU, and the four workgroups
    // already cover each other's loads: 158 us without, 162-171 us with, config 3 at 2^26.)
    cf x[PR], xh[2];
    auto load_tile = [&](size_t t) {
        const long long ib = R * (static_cast<long long>(t) * ts - ovl);
        if (ib - hl >= 0 && static_cast<size_t>(ib) + 256u * PR <= a.n) {  // interior tile: no edge handling
            const float2* src = a.in + ib;
#pragma unroll
            for (int m = 0; m < PR; ++m) x[m] = to_cf((src + 256 * m)[static_cast<unsigned>(tid)]);
#pragma unroll
            for (int m = 0; m < 2; ++m)
                xh[m] = slot_h[m] >= 0 ? to_cf((src - hl + 256 * m)[static_cast<unsigned>(tid)]) : cf{0.f, 0.f};
        } else {
#pragma unroll
            for (int m = 0; m < PR; ++m) x[m] = to_cf(stream_at(a.in, a.hist, a.hist_len, ib + tid + 256 * m, a.n));
#pragma unroll
            for (int m = 0; m < 2; ++m)
                xh[m] = slot_h[m] >= 0 ? to_cf(stream_at(a.in, a.hist, a.hist_len, ib - hl + tid + 256 * m, a.n))
                                       : cf{0.f, 0.f};
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

    if (PF) load_tile(t0);
    for (size_t t = t0; t < t1; ++t) {
        const long long jb = static_cast<long long>(t) * ts - ovl;  // first output computed by this tile
        // ---- stage the tile: 2R rows of new samples, then the halo (<= 2 rows)
        if (!PF) load_tile(t);
        DC_STAMP(0)  // global loads landed
        if (pre) {
#pragma unroll
            for (int m = 0; m < PR; ++m)
                x[m] = cmulf(x[m], PF ? (m ? cmulf(lrow[0], to_cf(a.step[m])) : lrow[0]) : lrow[m < NROW ? m : 0]);
#pragma unroll
            for (int m = 0; m < 2; ++m) xh[m] = cmulf(xh[m], lhalo[m]);
        }
#pragma unroll
        for (int m = 0; m < PR; ++m) sh[slot[m]] = x[m];
#pragma unroll
        for (int m = 0; m < 2; ++m)
            if (slot_h[m] >= 0) sh[slot_h[m]] = xh[m];
        DC_STAMP(1)  // mixer + LDS writes
        lds_barrier();
        DC_STAMP(2)
        if (PF && t + 1 < t1) load_tile(t + 1);  // in flight during the filter loop

        // ---- outputs j = jb + 2 tid + c:  y_c = sum_k h[k] u[R j - k]
        // tile sample index of u[R j - k] is 2R (tid + hlq) + (R c - k) = 2R (tid + hlq - d) + p with
        // k = 2R d + R c - p: block d, phase p descending = taps ascending; A[m] = h[m - (2R-1)]
        // Software pipeline over chunks of CH taps: the next chunk's LDS reads and tap loads are
        // issued before this chunk's FMAs.  The empty asm "uses" this chunk's registers first, so
        // the one s_waitcnt the compiler needs for them lands BEFORE the prefetch is issued (SMEM
        // returns out of order: any later wait would be lgkmcnt(0) and drain the prefetch too).
        cf acc0 = cf{0.f, 0.f}, acc1 = cf{0.f, 0.f};
        const cf* up = sh + tid + a.hlq;
        constexpr int CH = R <= 10 ? R : R / 2;   // taps per chunk; PR / CH chunks (2 or 4) per block of 2R
        constexpr int NCH = PR / CH;
        constexpr int NP = (CH + R) / 2 + (R & 1);  // SGPR pairs (A[2i], A[2i+1]) covering the chunk's taps of both outputs
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
                const int e0 = off + i, e1 = off + i + R;
                mac_tap<REAL>(acc0, u[i], tr[e0 >> 1], ti[e0 >> 1], (e0 & 1) != 0);
                mac_tap<REAL>(acc1, u[i], tr[e1 >> 1], ti[e1 >> 1], (e1 & 1) != 0);
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

        // ---- epilogue: mixer after the FIR, FM demod, stores
        DC_STAMP(4)  // filter loop
        float2 y0 = to_f2(acc0), y1 = to_f2(acc1);
        if (pre) {
            const cf tt = cf{static_cast<float>(tt_c), static_cast<float>(tt_s)};
            y0 = to_f2(cmulf(acc0, tt));
            y1 = to_f2(cmulf(acc1, tt));
            rotor_step(tt_c, tt_s, a.tile_c, a.tile_s);
        }
        if (post) {
            const cf ro = cf{static_cast<float>(ro_c), static_cast<float>(ro_s)};
            y0 = to_f2(cmulf(acc0, ro));
            y1 = to_f2(cmulf(acc1, cmulf(ro, to_cf(a.step_r))));
            rotor_step(ro_c, ro_s, a.tile_c, a.tile_s);
        }
        const long long j0 = jb + 2 * tid, j1 = j0 + 1;
        if (fm) {
            if (j0 < 0) y0 = a.fm_prev[0];  // FM.prev of the previous call stands in for y[-1]
            if (l == 63) sh_y[w] = y1;
        }
        lds_barrier();  // sh_y visible; every lane is done reading the staged tile
        if (fm) {
            float2 p0 = make_float2(__shfl_up(y1.x, 1), __shfl_up(y1.y, 1));
            if (l == 0 && w > 0) p0 = sh_y[w - 1];
            float* o = static_cast<float*>(a.out);
            if (tid > 0 && j0 < static_cast<long long>(a.n_out)) o[j0] = fm_step_fast(y0, p0);
            if (j1 < static_cast<long long>(a.n_out)) o[j1] = fm_step_fast(y1, y0);
            if (j0 == static_cast<long long>(a.n_out) - 1) a.fm_prev_new[0] = y0;
            if (j1 == static_cast<long long>(a.n_out) - 1) a.fm_prev_new[0] = y1;
        } else {
            float2* o = static_cast<float2*>(a.out);
            if (j0 < static_cast<long long>(a.n_out)) o[j0] = y0;
            if (j1 < static_cast<long long>(a.n_out)) o[j1] = y1;
        }
        DC_STAMP(5)  // barrier 2 + epilogue + stores
    }
    if (a.stamps && l == 0)
        for (int i = 0; i < 6; ++i) a.stamps[(static_cast<size_t>(blockIdx.x) * (DC_WG / 64) + w) * 8 + i] = st_acc[i];
#undef DC_STAMP
}

template <int R, bool REAL, bool PRE, bool PF>
static comms_status_t launch_decim_v(const DecimArgs& a, unsigned blocks, size_t lds, hipStream_t s) {
    static DeviceOnce attr_once;
    if (attr_once.need())
        COMMS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fir_decim_kernel<R, REAL, PRE, PF>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    fir_decim_kernel<R, REAL, PRE, PF><<<dim3(blocks), dim3(DC_WG), lds, s>>>(a);
    return launch_ok("fir_decim_kernel");
}

// prefetch variant per chain form: COMMS_DECIM_PREFETCH = bit 0 (mixer after the FIR), bit 1 (mixer first)
static int decim_prefetch_mask() {
    static const int m = [] {
        const char* v = getenv("COMMS_DECIM_PREFETCH");
        return v && *v ? atoi(v) : 3;
    }();
    return m;
}

template <int R>
static comms_status_t launch_decim(const DecimArgs& a, bool real, hipStream_t s) {
    constexpr size_t lds = DcGeom<R>::LDS;
    // persistent grid: every workgroup slot of the chip gets a contiguous run of tiles
    const size_t slots = static_cast<size_t>(DcGeom<R>::WGPC) * kNumCU;
    const unsigned blocks = static_cast<unsigned>(a.n_tiles < slots ? a.n_tiles : slots);
    const bool pre = (a.mode & COMMS_CHAIN_PRE) != 0;
    const bool pf = (decim_prefetch_mask() >> (pre ? 1 : 0)) & 1;
    switch ((real ? 4 : 0) | (pre ? 2 : 0) | (pf ? 1 : 0)) {
        case 0: return launch_decim_v<R, false, false, false>(a, blocks, lds, s);
        case 1: return launch_decim_v<R, false, false, true>(a, blocks, lds, s);
        case 2: return launch_decim_v<R, false, true, false>(a, blocks, lds, s);
        case 3: return launch_decim_v<R, false, true, true>(a, blocks, lds, s);
        case 4: return launch_decim_v<R, true, false, false>(a, blocks, lds, s);
        case 5: return launch_decim_v<R, true, false, true>(a, blocks, lds, s);
        case 6: return launch_decim_v<R, true, true, false>(a, blocks, lds, s);
        default: return launch_decim_v<R, true, true, true>(a, blocks, lds, s);
    }
}

}  // namespace comms

using namespace comms;

static unsigned long long* g_decim_stamps = nullptr;

extern "C" {

#ifdef COMMS_DIAG
// diagnostic hook (diagnostic build only): device buffer of 8 u64 per wave, or NULL
void comms_debug_decim_stamps(void* d_buf) { g_decim_stamps = static_cast<unsigned long long*>(d_buf); }
#endif

// Whether (taps, rate) runs on the decimating kernel: 0 = no (no instantiation for the rate, or
// taps beyond the kernel-argument budget), 1 = it can, 2 = and with few enough MACs per input
// sample that it beats the fused overlap-save launch (measured crossover).
int32_t comms_fir_decim_supported(const comms_fir_t* h, uint32_t rate) {
    if (!h || h->n_eff < 1 || h->n_eff > DC_NMAX) return 0;
    switch (rate) {
        case 2: case 3: case 4: case 5: case 6: case 8: case 10: case 12: case 16: break;
        default: return 0;
    }
    static const int max_macs = [] {
        const char* v = getenv("COMMS_DECIM_MAX_MACS");
        return v && *v ? atoi(v) : 44;
    }();
    const int macs = (h->n_eff + static_cast<int>(rate) - 1) / static_cast<int>(rate) * (h->real_taps ? 1 : 2);
    return macs <= max_macs ? 2 : 1;
}

comms_status_t comms_fir_run_decim_dev(comms_fir_t* h, const comms_c32* d_in, size_t n, void* d_out, int32_t mode,
                                       uint64_t turns0, uint64_t frac, uint32_t rate, const void* fm_prev,
                                       void* fm_prev_new, void* stream) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG((d_in && d_out) || !n, "NULL device pointer");
    COMMS_ARG(comms_fir_decim_supported(h, rate) != 0,
              "the decimating chain kernel supports <= 257 taps and rates 2,3,4,5,6,8,10,12,16");
    COMMS_ARG(n % rate == 0, "n must be a multiple of the decimation rate");
    COMMS_ARG((mode & COMMS_CHAIN_DEC) && !((mode & COMMS_CHAIN_PRE) && (mode & COMMS_CHAIN_POST)), "bad chain mode");
    COMMS_TRY(use_device(h->device));
    if (!n) return COMMS_OK;
    COMMS_ARG(!ranges_overlap(d_in, n * 8, d_out, (n / rate) * ((mode & COMMS_CHAIN_FM) ? 4 : 8)),
              "the decimating chain cannot run in place");
    hipStream_t s = nullptr;
    COMMS_TRY(h->enter(stream, &s));
    const int R = static_cast<int>(rate), PR = 2 * R, N = h->n_eff;
    DecimArgs a{};
    a.in = reinterpret_cast<const float2*>(d_in);
    a.hist = h->d_hist[h->cur];
    a.new_hist = h->d_hist[h->cur ^ 1];
    a.out = d_out;
    a.fm_prev = static_cast<const float2*>(fm_prev);
    a.fm_prev_new = static_cast<float2*>(fm_prev_new);
    a.n = n;
    a.n_out = n / rate;
    const bool fm = (mode & COMMS_CHAIN_FM) != 0;
    const size_t ts = DC_TILE - (fm ? 1 : 0);
    a.n_tiles = (a.n_out + ts - 1) / ts;
    a.hist_len = h->n_eff;
    a.hlq = (N - 1 + PR - 1) / PR;
    a.nd = a.hlq + 1;
    a.mode = mode;
    a.turns0 = turns0;
    a.frac = frac;
    a.stamps = g_decim_stamps;
    if (getenv("COMMS_DECIM_DEBUG_NOMAC")) a.nd = 0;  // diagnostic: staging + epilogue only
    mix_host_rotor(static_cast<uint64_t>(R) * ts * frac, a.tile_c, a.tile_s);
    double c, sn;
    mix_host_rotor(static_cast<uint64_t>(R) * frac, c, sn);
    a.step_r = make_float2(static_cast<float>(c), static_cast<float>(sn));
    for (int m = 0; m < PR; ++m) {
        mix_host_rotor(static_cast<uint64_t>(256 * m) * frac, c, sn);
        a.step[m] = make_float2(static_cast<float>(c), static_cast<float>(sn));
    }
    COMMS_ARG(PR * a.nd + R + 4 <= DC_AMAX, "tap table overflow");
    for (int m = 0; m < DC_AMAX; ++m) {
        const int k = m - (PR - 1);
        const bool in_range = k >= 0 && k < N;
        a.are[m] = in_range ? h->taps[k].re : 0.f;
        a.aim[m] = in_range ? h->taps[k].im : 0.f;
    }
    const bool real = h->real_taps;
    h->tic(s);
    comms_status_t st;
    switch (R) {
        case 2: st = launch_decim<2>(a, real, s); break;
        case 3: st = launch_decim<3>(a, real, s); break;
        case 4: st = launch_decim<4>(a, real, s); break;
        case 5: st = launch_decim<5>(a, real, s); break;
        case 6: st = launch_decim<6>(a, real, s); break;
        case 8: st = launch_decim<8>(a, real, s); break;
        case 10: st = launch_decim<10>(a, real, s); break;
        case 12: st = launch_decim<12>(a, real, s); break;
        case 16: st = launch_decim<16>(a, real, s); break;
        default: return fail(COMMS_ERR_ARG, "no decimating kernel for rate %d", R);
    }
    h->toc(s);
    COMMS_TRY(st);
    h->cur ^= 1;
    return COMMS_OK;
}

}  // extern "C"
