// demod.hip -- TimingEstimator and NCO.
//
// TimingEstimator::push  src/demodulation/timing_estimator.rs:85-112 (Mengali 8.4, f64):
//     r[i]   = exp(-i*pi*i/n)                       (:90)
//     qin[i] = conj(s[i])*r[i],  din[i] = s[i]*r[i] (:93-94)
//     qout   = batch_fir(qin, q(t) taps, zeros)     (:102, fresh state every push)
//     dout   = batch_fir(din, delay of n*d, zeros)  (:103)
//     est    = -n * arg( sum qout*dout ) / (2*pi)   (:108-111)
//   One kernel: a workgroup stages the mixed window of its 256 outputs in LDS, every lane
//   runs the q filter for its output in the reference's tap order (k = 0.. ascending, so
//   qout is the same f64 value), multiplies by the delayed sample and the products are
//   tree-reduced (the reference folds them sequentially: rounding-level difference only).
//
// Nco::push  src/demodulation/nco.rs:71-77: phase += dphase + perr; single wrap above 2*pi;
//   out = exp(i*phase).  The recurrence is a prefix sum, so a block of phase errors is a
//   scan: increments are converted to 64-bit fixed-point turns (wrap-around of the integer
//   is the mod-2*pi), scanned exactly, and one f64 sincos per sample produces the output.
//   Three launches: tile sums, scan of the tile sums (one workgroup), apply (a persistent grid;
//   every lane owns coalesced sample pairs).  A single-launch decoupled look-back was built and
//   measured: 3.6 ms against 0.19 -- with ~2000 tiles in flight a tile walks back through that
//   many descriptors before it meets an inclusive prefix, and the walk is a chain of dependent
//   L2 round trips; reading the 8-byte phase errors twice costs far less.
//   exp(i*phase) comes from a 1024-entry f64 (cos, sin) table of the top 10 phase bits in LDS
//   and a 6th / 7th-order Taylor rotor of the remainder (|lo| <= pi/1024: error < 1e-21), a
//   tenth of the library sincos' instructions.  Differences to the reference's sequentially
//   rounded f64 phase stay below ~n * 2^-63 turns.
#include <cmath>
#include <vector>

#include "common.hpp"

namespace comms {

constexpr double kPi = 3.14159265358979323846264338327950288;

// ---------------------------------------------------------------- timing estimator
constexpr int TE_WG = 256;                  // lanes per workgroup
constexpr int TE_OPL = 8;                   // consecutive outputs per lane
constexpr int TE_TILE = TE_WG * TE_OPL;     // outputs per tile
constexpr int TE_QMAX = 1020;               // q(t) taps one pass stages (a multiple of 12; 2*n*d + 1 taps in all: longer filters take several passes)

// sin / cos of a large f64 angle (|th| up to ~1e8: the reference feeds -pi*i/n with i the sample index):
// three-term Cody-Waite reduction by pi/2 with FMAs, then fdlibm's kernel polynomials on |r| <= pi/4.
// ~30 f64 operations against the library routine's several hundred (it also handles |th| -> inf); absolute
// error < 4e-16 for |th| < 2^27, far inside the estimator's 1e-9.
__device__ __forceinline__ void te_sincos(double th, double& sn, double& cs) {
    const double kf = rint(th * 6.36619772367581382433e-01);
    double r = __fma_rn(-kf, 1.57079632673412561417e+00, th);
    r = __fma_rn(-kf, 6.07710050630396597660e-11, r);
    r = __fma_rn(-kf, 2.02226624871116645580e-21, r);
    r = __fma_rn(-kf, 8.47842766036889956997e-32, r);
    const double z = r * r;
    double ps = __fma_rn(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = __fma_rn(z, ps, 2.75573137070700676789e-06);
    ps = __fma_rn(z, ps, -1.98412698298579493134e-04);
    ps = __fma_rn(z, ps, 8.33333333332248946124e-03);
    ps = __fma_rn(z, ps, -1.66666666666666324348e-01);
    const double s0 = __fma_rn(r * z, ps, r);
    double pc = __fma_rn(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = __fma_rn(z, pc, -2.75573143513906633035e-07);
    pc = __fma_rn(z, pc, 2.48015872894767294178e-05);
    pc = __fma_rn(z, pc, -1.38888888888741095749e-03);
    pc = __fma_rn(z, pc, 4.16666666666666019037e-02);
    const double c0 = __fma_rn(z * z, pc, __fma_rn(z, -0.5, 1.0));
    const int q = static_cast<int>(static_cast<long long>(kf)) & 3;
    const double a = (q & 1) ? c0 : s0, b = (q & 1) ? s0 : c0;
    sn = (q & 2) ? -a : a;
    cs = ((q + 1) & 2) ? -b : b;
}

__device__ __forceinline__ double2 te_rotor(long long i, double n_sps) {
    // Complex::new(0.0, -PI * i as f64 / n as f64).exp()  -> (cos, sin) of that angle
    const double th = (-kPi * static_cast<double>(i)) / n_sps;
    double s, c;
    te_sincos(th, s, c);
    return make_double2(c, s);
}

// LDS image of the mixed window: element e at e + (e >> 3) -- one pad slot per eight elements, so that lanes
// reading at a stride of eight elements (their OPL consecutive outputs; 144 bytes apart) spread over all the banks
__device__ __forceinline__ int te_slot(int e) { return e + (e >> 3); }

// A 256-lane workgroup owns 2048 consecutive outputs, eight per lane.  The mixed window (qin) of the tile is staged
// in LDS once and the q(t) filter slides a register window over the LDS image: one 16-byte LDS read feeds a tap of
// all eight outputs (8 complex x real MACs = 16 f64 FMAs; four outputs per lane, 8 FMAs per read, kept the LDS as
// busy as the FP64 pipe: 281 us at 2^24 samples), the taps themselves arrive by scalar loads.  Taps are walked in
// the reference's order (k ascending); the MACs are fused (the reference's are not: the estimate moves by < 1e-12,
// the parity tests allow 1e-9).  The delayed sample din[i] = x[i - nd] r[i - nd] of the final product is formed
// from memory again (the tile's samples are still in L2) -- 16 LDS bytes per output less, which is what lets
// four such workgroups share a CU.
__global__ __launch_bounds__(TE_WG, 4) void timing_kernel(const double2* __restrict__ x, size_t len,
                                                       const double* __restrict__ qtaps, uint32_t n_q, uint32_t k_lo,
                                                       uint32_t nd, double n_sps, double2 w_wg, double2 w_1, int with_delay,
                                                       double2* __restrict__ qacc, double2* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) char te_smem[];
    double2* sh_q = reinterpret_cast<double2*>(te_smem);                       // window, padded image
    const int nk = static_cast<int>(n_q);                                      // taps of this pass
    const int win = TE_TILE + nk - 1;                                          // (+ 2 elements staged past it: positions -8 .. -1 of the last lane)
    __shared__ double2 wsum[TE_WG / 64];
    const int tid = threadIdx.x;
    const size_t ntiles = (len + TE_TILE - 1) / TE_TILE;
    double2 total = make_double2(0.0, 0.0);
    for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long i0 = static_cast<long long>(tile) * TE_TILE;
        __syncthreads();
        // ---- stage qin over [i0 - k_lo - (nk - 1), i0 - k_lo + TILE]
        const long long w0 = i0 - static_cast<long long>(k_lo) - (nk - 1);
        // (one sincos per lane and tile: the lane's further elements sit TE_WG samples apart, their rotor is the
        // previous one times w_wg = e^{-i pi TE_WG / n}; a sincos + f64 division per element cost as much as the filter)
        double2 r = te_rotor(w0 + tid, n_sps);
        for (int j = tid; j <= win + 1; j += TE_WG) {
            const long long idx = w0 + j;
            double2 v = make_double2(0.0, 0.0);
            if (idx >= 0 && idx < static_cast<long long>(len)) {
                const double2 sm = x[idx];
                const double ci = -sm.y;  // conj
                v = make_double2(sm.x * r.x - ci * r.y, sm.x * r.y + ci * r.x);
            }
            sh_q[te_slot(j)] = v;
            r = make_double2(r.x * w_wg.x - r.y * w_wg.y, r.x * w_wg.y + r.y * w_wg.x);
        }
        __syncthreads();
        // ---- q_c = sum_k t[k] v[n_c - k],  n_c = i0 + 8 tid + c:  window element of (c, k) is e0 + c - k
        const int e0 = (nk - 1) + TE_OPL * tid;
        double2 q[TE_OPL];
        if (qacc && k_lo) {
#pragma unroll
            for (int c = 0; c < TE_OPL; ++c) {
                const long long i = i0 + TE_OPL * tid + c;
                q[c] = i < static_cast<long long>(len) ? qacc[i] : make_double2(0.0, 0.0);
            }
        } else {
#pragma unroll
            for (int c = 0; c < TE_OPL; ++c) q[c] = make_double2(0.0, 0.0);
        }
        // Window positions p = e0 - e grow with k: tap k of output c reads position k - c.  Positions live in
        // register blocks of four; the taps of block b (k = 4b .. 4b + 3) touch blocks b - 2, b - 1 and b.  Three
        // blocks per loop iteration rotate through three register sets (R0, R1, R2), (R1, R2, R0), (R2, R0, R1),
        // so no value is ever moved between registers.  (The host pads the taps with zeros to a multiple of 12.)
        // Block b holds the elements e0 - 4b - m, m < 4 (0 <= e <= win + 1: nk is a multiple of 12, the image holds
        // win + 2 elements).  e0 + 8 is 3 or 7 modulo 8 in every lane, so a block never straddles a pad slot and the
        // blocks' lowest slots lie alternately 4 and 5 slots apart: one subtraction of a wave-uniform step per block
        // instead of a padded-index computation per read.
        int blk_addr = (te_slot(e0 + 8) - 3) * static_cast<int>(sizeof(double2));  // block -2
        int blk_step = ((nk - 1) & 7) == 7 ? 4 * static_cast<int>(sizeof(double2)) : 5 * static_cast<int>(sizeof(double2));
        auto load_block = [&](int, double2 (&R)[4]) {  // called for blocks -2, -1, 0, 1, ... in this order
#pragma unroll
            for (int m = 0; m < 4; ++m)
                R[m] = *reinterpret_cast<const double2*>(te_smem + blk_addr + (3 - m) * static_cast<int>(sizeof(double2)));
            blk_addr -= blk_step;
            blk_step = 9 * static_cast<int>(sizeof(double2)) - blk_step;
        };
        auto mac_block = [&](int blk, const double2 (&P2)[4], const double2 (&P1)[4], const double2 (&Q)[4]) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const double t = qtaps[4 * blk + m];  // wave-uniform: a scalar load
#pragma unroll
                for (int c = 0; c < TE_OPL; ++c) {
                    const int d = m - c;
                    const double2 v = d >= 0 ? Q[d] : d >= -4 ? P1[4 + d] : P2[8 + d];
                    q[c].x = __fma_rn(t, v.x, q[c].x);
                    q[c].y = __fma_rn(t, v.y, q[c].y);
                }
            }
        };
        double2 R0[4], R1[4], R2[4];
        load_block(-2, R0);
        load_block(-1, R1);
        const int nblk = nk / 4;  // a multiple of 3
        for (int blk = 0; blk < nblk; blk += 3) {
            load_block(blk, R2);
            mac_block(blk, R0, R1, R2);
            load_block(blk + 1, R0);
            mac_block(blk + 1, R1, R2, R0);
            load_block(blk + 2, R1);
            mac_block(blk + 2, R2, R0, R1);
        }
        // ---- delayed product, or (a pass that is not the last) the running filter sums back to memory
        // (the rotor of the lane's first delayed sample by sincos, the next seven by steps of w_1 = e^{-i pi / n})
        double2 rd = make_double2(1.0, 0.0);
        if (with_delay) rd = te_rotor(i0 + TE_OPL * tid - static_cast<long long>(nd), n_sps);
#pragma unroll
        for (int c = 0; c < TE_OPL; ++c) {
            const long long i = i0 + TE_OPL * tid + c;
            if (i < static_cast<long long>(len)) {
                if (!with_delay) {
                    qacc[i] = q[c];
                } else if (i >= static_cast<long long>(nd)) {
                    const double2 sm = x[i - static_cast<long long>(nd)];
                    const double2 d = make_double2(sm.x * rd.x - sm.y * rd.y, sm.x * rd.y + sm.y * rd.x);
                    total.x += q[c].x * d.x - q[c].y * d.y;
                    total.y += q[c].x * d.y + q[c].y * d.x;
                }
            }
            rd = make_double2(rd.x * w_1.x - rd.y * w_1.y, rd.x * w_1.y + rd.y * w_1.x);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        total.x += __shfl_down(total.x, off);
        total.y += __shfl_down(total.y, off);
    }
    if ((tid & 63) == 0) wsum[tid >> 6] = total;
    __syncthreads();
    if (tid == 0) {
        double2 s = wsum[0];
        for (int w = 1; w < TE_WG / 64; ++w) {
            s.x += wsum[w].x;
            s.y += wsum[w].y;
        }
        partials[blockIdx.x] = s;
    }
}

// ---------------------------------------------------------------- NCO
constexpr int NCO_WG = 256;
constexpr int NCO_PER = 8;                    // samples per lane
constexpr int NCO_TILE = NCO_WG * NCO_PER;    // 2048 samples per workgroup
constexpr double kTwoPi = 2.0 * kPi;          // the reference's 2.0 * PI

// (dphase + perr) -> fixed-point turns (two's complement, modulo one turn).  1/(2*pi) is
// carried as hi + lo so that the conversion error stays near 2^-64 turns for |x| <= 2*pi.
__device__ __forceinline__ uint64_t nco_turns(double x) {
    constexpr double inv_hi = 0x1.45f306dc9c883p-3;   // fl(1/(2*pi))
    constexpr double inv_lo = -0x1.6b01ec5417056p-57;  // 1/(2*pi) - inv_hi
    const double hi = x * inv_hi;
    const double lo = __fma_rn(x, inv_hi, -hi) + x * inv_lo;
    // hi = whole + frac; whole turns vanish modulo 2^64
    const double fr = hi - rint(hi);  // exact, in [-0.5, 0.5]
    const uint64_t a = static_cast<uint64_t>(static_cast<long long>(fr * 0x1.0p63)) << 1;  // fr * 2^64 mod 2^64
    const uint64_t b = static_cast<uint64_t>(static_cast<long long>(rint(lo * 0x1.0p64)));
    return a + b;
}

__device__ __forceinline__ uint64_t wave_incl_scan(uint64_t v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint64_t o = __shfl_up(v, off);
        if (lane >= off) v += o;
    }
    return v;
}

constexpr int NCO_ROWS = 4;                   // a lane owns the sample pairs (2 tid, 2 tid + 1) + 512 j, j < 4

// exp(i * 2*pi * ph / 2^64): table of the nearest of 1024 directions times the Taylor rotor of the rest
__device__ __forceinline__ double2 nco_rotor(uint64_t ph, const double2* __restrict__ tab) {
    const uint64_t r = ph + (static_cast<uint64_t>(1) << 53);
    const unsigned idx = static_cast<unsigned>(r >> 54);                              // nearest direction (mod 1024)
    const long long rem = static_cast<long long>(ph - (static_cast<uint64_t>(idx) << 54));  // in [-2^53, 2^53)
    const double lo = static_cast<double>(rem) * (kTwoPi * 0x1.0p-64);                // |lo| <= pi / 1024
    const double z = lo * lo;
    const double sl = lo * __fma_rn(z, __fma_rn(z, 1.0 / 120.0, -1.0 / 6.0), 1.0);
    const double cl = __fma_rn(z, __fma_rn(z, __fma_rn(z, -1.0 / 720.0, 1.0 / 24.0), -0.5), 1.0);
    const double2 t = tab[idx & 1023];
    return make_double2(__fma_rn(-t.y, sl, t.x * cl), __fma_rn(t.x, sl, t.y * cl));
}

// the lane's increments of tile t: p0[j] / p1[j] = running sum after the first / second sample of pair j
__device__ __forceinline__ void nco_load_pairs(const double* __restrict__ perr, size_t n, double dphase, size_t base,
                                               uint64_t (&p0)[NCO_ROWS], uint64_t (&p1)[NCO_ROWS]) {
#pragma unroll
    for (int j = 0; j < NCO_ROWS; ++j) {
        const size_t i = base + 512 * static_cast<size_t>(j);
        double e0 = 0.0, e1 = 0.0;
        if (i + 1 < n) {
            const double2 e = *reinterpret_cast<const double2*>(perr + i);
            e0 = e.x;
            e1 = e.y;
        } else if (i < n) {
            e0 = perr[i];
        }
        p0[j] = i < n ? nco_turns(dphase + e0) : 0;
        p1[j] = p0[j] + (i + 1 < n ? nco_turns(dphase + e1) : 0);
    }
}

// tile sums
__global__ __launch_bounds__(NCO_WG) void nco_sum_kernel(const double* __restrict__ perr, size_t n, double dphase,
                                                         uint64_t* __restrict__ tile_sum, size_t ntiles) {
    __shared__ uint64_t wsum[NCO_WG / 64];
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        uint64_t p0[NCO_ROWS], p1[NCO_ROWS];
        nco_load_pairs(perr, n, dphase, t * NCO_TILE + 2 * static_cast<size_t>(threadIdx.x), p0, p1);
        uint64_t acc = 0;
#pragma unroll
        for (int j = 0; j < NCO_ROWS; ++j) acc += p1[j];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t s = 0;
            for (int w = 0; w < NCO_WG / 64; ++w) s += wsum[w];
            tile_sum[t] = s;
        }
    }
}

// exclusive scan of the tile sums in place (one workgroup), seeded with the node's phase;
// the total (phase after the block) goes to *phase_io.
__global__ __launch_bounds__(1024) void nco_scan_kernel(uint64_t* __restrict__ tile_sum, size_t ntiles,
                                                        uint64_t* __restrict__ phase_io) {
    // a lane owns 8 consecutive tile sums per sweep (8192 per sweep: one sweep up to 2^24 samples); its 8 loads are in
    // flight together, the lane totals are scanned per wave by shuffles and across the 16 waves through LDS
    constexpr int PER = 8;
    __shared__ uint64_t wtot[16];
    __shared__ uint64_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) carry_s = *phase_io;
    __syncthreads();
    for (size_t b0 = 0; b0 < ntiles; b0 += 1024 * PER) {
        const size_t i0 = b0 + static_cast<size_t>(tid) * PER;
        uint64_t v[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) v[k] = i0 + k < ntiles ? tile_sum[i0 + k] : 0;
        uint64_t tot = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) tot += v[k];
        const uint64_t inc = wave_incl_scan(tot, lane);
        if (lane == 63) wtot[w] = inc;
        __syncthreads();
        uint64_t off = carry_s + inc - tot;  // exclusive prefix of this lane's first element
        for (int k = 0; k < w; ++k) off += wtot[k];
        uint64_t run = off;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            if (i0 + k < ntiles) tile_sum[i0 + k] = run;
            run += v[k];
        }
        __syncthreads();
        if (tid == 1023) carry_s = run;
        __syncthreads();
    }
    if (tid == 0) *phase_io = carry_s;
}

// out[i] = exp(i * phase_i),  phase_i = phase_before + sum_{j<=i} (dphase + perr[j])
__global__ __launch_bounds__(NCO_WG) void nco_apply_kernel(const double* __restrict__ perr, size_t n, double dphase,
                                                           const uint64_t* __restrict__ tile_off, size_t ntiles,
                                                           const double2* __restrict__ tab_g, double2* __restrict__ out) {
    __shared__ double2 tab[1024];
    __shared__ uint64_t wtot[NCO_ROWS][NCO_WG / 64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < 1024; i += NCO_WG) tab[i] = tab_g[i];
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const size_t base = t * NCO_TILE + 2 * static_cast<size_t>(tid);
        uint64_t p0[NCO_ROWS], p1[NCO_ROWS], inc[NCO_ROWS];
        nco_load_pairs(perr, n, dphase, base, p0, p1);
        __syncthreads();  // table in place; the previous tile's wtot no longer read
#pragma unroll
        for (int j = 0; j < NCO_ROWS; ++j) {
            inc[j] = wave_incl_scan(p1[j], lane);
            if (lane == 63) wtot[j][w] = inc[j];
        }
        __syncthreads();
        uint64_t row_off = tile_off[t];
#pragma unroll
        for (int j = 0; j < NCO_ROWS; ++j) {
            uint64_t off = row_off + inc[j] - p1[j];
#pragma unroll
            for (int k = 0; k < NCO_WG / 64; ++k) {
                if (k < w) off += wtot[j][k];
                row_off += wtot[j][k];
            }
            const size_t i = base + 512 * static_cast<size_t>(j);
            const double2 a = nco_rotor(off + p0[j], tab);
            const double2 b = nco_rotor(off + p1[j], tab);
            if (i + 1 < n) {
                double4 v;
                v.x = a.x; v.y = a.y; v.z = b.x; v.w = b.y;
                *reinterpret_cast<double4*>(out + i) = v;
            } else if (i < n) {
                out[i] = a;
            }
        }
    }
}

}  // namespace comms

using namespace comms;

struct comms_timing : Handle {
    uint32_t n = 0, d = 0, n_q = 0;
    double* d_taps = nullptr;
    double2* d_part = nullptr;
    unsigned max_blocks = 4 * kNumCU;
    Scratch qacc;  // running filter sums between the passes of a filter longer than TE_QMAX taps
};

struct comms_nco : Handle {
    double dphase = 0.0;
    uint64_t* d_phase = nullptr;  // fixed-point turns, device resident
    double2* d_tab = nullptr;     // (cos, sin)(2 pi k / 1024), f64
    Scratch tiles;
};

static comms_status_t qfilt_host(uint32_t n_taps, double alpha, uint32_t sam_per_sym, std::vector<double>& out) {
    COMMS_ARG(alpha >= 0.0 && alpha <= 1.0, "InvalidRolloffError: alpha=%g outside [0,1]", alpha);
    COMMS_ARG(sam_per_sym >= 1, "sam_per_sym must be >= 1");
    COMMS_ARG(n_taps < (1u << 30), "n_taps too large");
    const uint32_t real_n = n_taps % 2 == 0 ? n_taps + 1 : n_taps;  // util/math.rs:317-320
    const int32_t half = static_cast<int32_t>(std::floor(static_cast<double>(real_n) / 2.0));
    out.resize(real_n);
    for (uint32_t i = 0; i < real_n; ++i) {
        const double tt = static_cast<double>(static_cast<int32_t>(i) - half) / static_cast<double>(sam_per_sym);
        const double two_alpha_tt = 2.0 * alpha * tt;
        if (std::fabs(two_alpha_tt) == 1.0) {  // l'Hospital branch, util/math.rs:331-333
            out[i] = std::sin(kPi * alpha * tt) / (8.0 * tt);
        } else {
            const double num = alpha * std::cos(kPi * alpha * tt);
            const double den = kPi * (1.0 - (two_alpha_tt * two_alpha_tt));
            out[i] = num / den;
        }
    }
    return COMMS_OK;
}

extern "C" {

size_t comms_qfilt_len(uint32_t n_taps) { return n_taps % 2 == 0 ? static_cast<size_t>(n_taps) + 1 : n_taps; }

comms_status_t comms_qfilt_taps(uint32_t n_taps, double alpha, uint32_t sam_per_sym, double* out) {
    COMMS_ARG(out != nullptr, "out is NULL");
    std::vector<double> t;
    COMMS_TRY(qfilt_host(n_taps, alpha, sam_per_sym, t));
    std::memcpy(out, t.data(), t.size() * sizeof(double));
    return COMMS_OK;
}

comms_status_t comms_timing_create(uint32_t n, uint32_t d, double alpha, int32_t device, comms_timing_t** out) {
    COMMS_ARG(out != nullptr, "out is NULL");
    *out = nullptr;
    COMMS_ARG(n >= 1, "samples per symbol must be >= 1");
    COMMS_ARG(static_cast<uint64_t>(n) * d < (1u << 24), "filter delay n*d too large");
    std::vector<double> taps;
    COMMS_TRY(qfilt_host(2 * n * d + 1, alpha, n, taps));
    comms_timing* h = new (std::nothrow) comms_timing;
    COMMS_ARG(h != nullptr, "out of host memory");
    comms_status_t st = h->init(device);
    if (st != COMMS_OK) {
        delete h;
        return st;
    }
    h->n = n;
    h->d = d;
    h->n_q = static_cast<uint32_t>(taps.size());
    taps.resize((taps.size() + 11) / 12 * 12, 0.0);  // the kernel walks the taps in blocks of 12; zero taps add nothing
    hipError_t e = hipMalloc(&h->d_taps, taps.size() * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(h->d_taps, taps.data(), taps.size() * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&h->d_part, h->max_blocks * sizeof(double2));
    if (e != hipSuccess) {
        if (h->d_taps) (void)hipFree(h->d_taps);
        h->fini();
        delete h;
        return fail(COMMS_ERR_DEVICE, "timing estimator alloc: %s", hipGetErrorString(e));
    }
    *out = h;
    return COMMS_OK;
}

comms_status_t comms_timing_push_dev(comms_timing_t* h, const double* d_samples, size_t len, double* estimate,
                                     void* stream) {
    COMMS_ARG(h != nullptr && estimate != nullptr, "NULL argument");
    COMMS_ARG(d_samples || !len, "NULL device pointer");
    COMMS_ARG((reinterpret_cast<uintptr_t>(d_samples) & 15) == 0, "samples must be 16-byte aligned");
    COMMS_TRY(use_device(h->device));
    hipStream_t s = nullptr;
    COMMS_TRY(h->enter(stream, &s));
    double re = 0.0, im = 0.0;
    if (len) {
        size_t blocks = (len + TE_TILE - 1) / TE_TILE;
        if (blocks > h->max_blocks) blocks = h->max_blocks;
        const uint32_t nq12 = (h->n_q + 11) / 12 * 12;
        const uint32_t n_pass = (nq12 + TE_QMAX - 1) / TE_QMAX;
        double2* qacc = nullptr;
        if (n_pass > 1) {
            COMMS_TRY(h->qacc.reserve(len * sizeof(double2)));
            qacc = static_cast<double2*>(h->qacc.p);
        }
        // rotor steps e^{-i pi m / n} for m = TE_WG and 1 (the kernel's lanes walk their samples with them)
        const double a_wg = -kPi * static_cast<double>(TE_WG % (2 * static_cast<uint64_t>(h->n))) / static_cast<double>(h->n);
        const double a_1 = -kPi / static_cast<double>(h->n);
        const double2 w_wg = make_double2(std::cos(a_wg), std::sin(a_wg)), w_1 = make_double2(std::cos(a_1), std::sin(a_1));
        h->tic(s);
        for (uint32_t p = 0; p < n_pass; ++p) {
            const uint32_t k_lo = p * TE_QMAX;
            const uint32_t nk = nq12 - k_lo < static_cast<uint32_t>(TE_QMAX) ? nq12 - k_lo : TE_QMAX;
            const int win = TE_TILE + static_cast<int>(nk) - 1;
            const size_t lds = (static_cast<size_t>(win + 1 + ((win + 1) >> 3)) + 1) * sizeof(double2);  // te_slot(win + 1) + 1 slots
            timing_kernel<<<dim3(static_cast<unsigned>(blocks)), dim3(TE_WG), lds, s>>>(
                reinterpret_cast<const double2*>(d_samples), len, h->d_taps + k_lo, nk, k_lo, h->n * h->d,
                static_cast<double>(h->n), w_wg, w_1, p + 1 == n_pass ? 1 : 0, qacc, h->d_part);
        }
        h->toc(s);
        COMMS_TRY(launch_ok("timing_kernel"));
        std::vector<double2> part(blocks);
        COMMS_HIP_TRY(hipMemcpyAsync(part.data(), h->d_part, blocks * sizeof(double2), hipMemcpyDeviceToHost, s));
        COMMS_HIP_TRY(hipStreamSynchronize(s));
        for (size_t b = 0; b < blocks; ++b) {
            re += part[b].x;
            im += part[b].y;
        }
    }
    // -(self.n as f64) * sum_value.arg() / (2.0 * PI)
    *estimate = (-static_cast<double>(h->n) * std::atan2(im, re)) / (2.0 * kPi);
    return COMMS_OK;
}

comms_status_t comms_timing_push(comms_timing_t* h, const double* samples, size_t len, double* estimate) {
    COMMS_ARG(h != nullptr && estimate != nullptr, "NULL argument");
    COMMS_ARG(samples || !len, "NULL host pointer");
    COMMS_TRY(use_device(h->device));
    if (len) {
        COMMS_TRY(h->in_scratch.reserve(len * 16));
        COMMS_HIP_TRY(hipMemcpyAsync(h->in_scratch.p, samples, len * 16, hipMemcpyHostToDevice, h->stream));
    }
    return comms_timing_push_dev(h, static_cast<const double*>(h->in_scratch.p), len, estimate, COMMS_STREAM_HANDLE);
}

comms_status_t comms_timing_destroy(comms_timing_t* h) {
    if (!h) return COMMS_OK;
    (void)use_device(h->device);
    if (h->d_taps) (void)hipFree(h->d_taps);
    if (h->d_part) (void)hipFree(h->d_part);
    h->qacc.release();
    h->fini();
    delete h;
    return COMMS_OK;
}

// ---- NCO -------------------------------------------------------------------------
comms_status_t comms_nco_create(double dphase, double phase, int32_t device, comms_nco_t** out) {
    COMMS_ARG(out != nullptr, "out is NULL");
    *out = nullptr;
    COMMS_ARG(std::isfinite(dphase) && std::isfinite(phase), "dphase and phase must be finite");
    comms_nco* h = new (std::nothrow) comms_nco;
    COMMS_ARG(h != nullptr, "out of host memory");
    comms_status_t st = h->init(device);
    if (st != COMMS_OK) {
        delete h;
        return st;
    }
    h->dphase = mix_wrap_dphase(dphase);  // Nco::new, nco.rs:41-49 (same wrap as Mixer::new)
    const uint64_t turns = mix_to_turns(phase);
    std::vector<double2> tab(1024);
    for (int k = 0; k < 1024; ++k) {
        const long double a = 2.0L * 3.14159265358979323846264338327950288L * static_cast<long double>(k) / 1024.0L;
        tab[k] = make_double2(static_cast<double>(cosl(a)), static_cast<double>(sinl(a)));
    }
    hipError_t e = hipMalloc(&h->d_phase, sizeof(uint64_t));
    if (e == hipSuccess) e = hipMemcpy(h->d_phase, &turns, sizeof(uint64_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&h->d_tab, tab.size() * sizeof(double2));
    if (e == hipSuccess) e = hipMemcpy(h->d_tab, tab.data(), tab.size() * sizeof(double2), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        comms_nco_destroy(h);
        return fail(COMMS_ERR_DEVICE, "nco state alloc: %s", hipGetErrorString(e));
    }
    *out = h;
    return COMMS_OK;
}

comms_status_t comms_nco_run_dev(comms_nco_t* h, const double* d_perr, size_t n, double* d_out, void* stream) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG((d_perr && d_out) || !n, "NULL device pointer");
    COMMS_ARG((reinterpret_cast<uintptr_t>(d_out) & 15) == 0, "output must be 16-byte aligned");
    COMMS_TRY(use_device(h->device));
    if (!n) return COMMS_OK;
    COMMS_ARG(!ranges_overlap(d_perr, n * 8, d_out, n * 16), "nco cannot run in place");
    hipStream_t s = nullptr;
    COMMS_TRY(h->enter(stream, &s));
    COMMS_ARG((reinterpret_cast<uintptr_t>(d_perr) & 15) == 0, "phase errors must be 16-byte aligned");
    const size_t ntiles = (n + NCO_TILE - 1) / NCO_TILE;
    COMMS_ARG(ntiles <= 0x7fffffffu, "block too long");
    COMMS_TRY(h->tiles.reserve(ntiles * sizeof(uint64_t)));
    uint64_t* tiles = static_cast<uint64_t*>(h->tiles.p);
    // persistent grids: the apply kernel stages a 16 KiB rotor table per workgroup (8 workgroups per CU)
    const size_t slots = static_cast<size_t>(8) * kNumCU;
    const unsigned blocks = static_cast<unsigned>(ntiles < slots ? ntiles : slots);
    h->tic(s);
    nco_sum_kernel<<<dim3(blocks), dim3(NCO_WG), 0, s>>>(d_perr, n, h->dphase, tiles, ntiles);
    nco_scan_kernel<<<dim3(1), dim3(1024), 0, s>>>(tiles, ntiles, h->d_phase);
    nco_apply_kernel<<<dim3(blocks), dim3(NCO_WG), 0, s>>>(d_perr, n, h->dphase, tiles, ntiles, h->d_tab,
                                                          reinterpret_cast<double2*>(d_out));
    h->toc(s);
    COMMS_TRY(launch_ok("nco kernels"));
    return COMMS_OK;
}

comms_status_t comms_nco_run(comms_nco_t* h, const double* perr, size_t n, double* out) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG((perr && out) || !n, "NULL host pointer");
    COMMS_TRY(use_device(h->device));
    if (!n) return COMMS_OK;
    COMMS_TRY(h->in_scratch.reserve(n * 8));
    COMMS_TRY(h->out_scratch.reserve(n * 16));
    COMMS_HIP_TRY(hipMemcpyAsync(h->in_scratch.p, perr, n * 8, hipMemcpyHostToDevice, h->stream));
    COMMS_TRY(comms_nco_run_dev(h, static_cast<const double*>(h->in_scratch.p), n,
                                static_cast<double*>(h->out_scratch.p), COMMS_STREAM_HANDLE));
    COMMS_HIP_TRY(hipMemcpyAsync(out, h->out_scratch.p, n * 16, hipMemcpyDeviceToHost, h->stream));
    COMMS_HIP_TRY(hipStreamSynchronize(h->stream));
    return COMMS_OK;
}

comms_status_t comms_nco_get_phase(comms_nco_t* h, double* phase) {
    COMMS_ARG(h != nullptr && phase != nullptr, "NULL argument");
    COMMS_TRY(use_device(h->device));
    uint64_t turns = 0;
    COMMS_TRY(h->quiesce());
    COMMS_HIP_TRY(hipMemcpy(&turns, h->d_phase, sizeof(uint64_t), hipMemcpyDeviceToHost));
    *phase = static_cast<double>(turns >> 11) * (kMixT * 0x1.0p-53);
    return COMMS_OK;
}

comms_status_t comms_nco_destroy(comms_nco_t* h) {
    if (!h) return COMMS_OK;
    (void)use_device(h->device);
    if (h->d_phase) (void)hipFree(h->d_phase);
    if (h->d_tab) (void)hipFree(h->d_tab);
    h->tiles.release();
    h->fini();
    delete h;
    return COMMS_OK;
}

}  // extern "C"
