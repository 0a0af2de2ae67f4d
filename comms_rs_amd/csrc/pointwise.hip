// pointwise.hip -- HBM-bound per-sample nodes: mixer, decimate, upsample, FM demod.
//
// Data layout: interleaved Complex<f32> (float2) streams in HBM, read and
// written once, 16 B per lane where alignment allows.  Built with
// -ffp-contract=off so the complex products keep the reference's unfused
// 4-mul/2-add form (num-complex `Mul`).
#include <cmath>

#include "common.hpp"

namespace comms {

// =============================================================== mixer
// Reference: Mixer::mix (src/mixer.rs:73-84).  The phase of sample n is
// phase0 + n*dphase reduced modulo T = fl(2*pi) exactly as the reference's
// `if phase > 2pi { phase -= 2pi }` does.  It is tracked as a 64-bit
// fixed-point fraction of T ("turns"), so any sample index can be evaluated
// in closed form: turns(n) = turns0 + n*frac (mod 2^64).  A thread evaluates
// its first rotor with one f64 sincos and then steps it by a constant rotor
// (f64 complex multiply) per grid sweep.
constexpr double kT = 2.0 * 3.14159265358979323846264338327950288;

__device__ inline void rotor_at(uint64_t turns, double& c, double& s) {
    double ang = static_cast<double>(turns >> 11) * (kT * 0x1.0p-53);
    sincos(ang, &s, &c);
}
__device__ inline float2 mix1(float2 x, double c, double s) {
    double xr = static_cast<double>(x.x), xi = static_cast<double>(x.y);
    // (xr + i xi) * (c + i s), num-complex form, f64, then `as f32`
    return make_float2(static_cast<float>(xr * c - xi * s), static_cast<float>(xr * s + xi * c));
}
__device__ inline void rot_step(double& c, double& s, double sc, double ss) {
    double nc = c * sc - s * ss;
    double ns = c * ss + s * sc;
    c = nc;
    s = ns;
}

// VEC = samples per thread per sweep (2 -> float4 accesses, needs 16-B alignment)
template <int VEC>
__global__ __launch_bounds__(256) void mixer_kernel(const float2* __restrict__ in,
                                                    float2* __restrict__ out, size_t n,
                                                    uint64_t turns0, uint64_t frac, double sweep_c,
                                                    double sweep_s, double d_c, double d_s) {
    const size_t nthreads = static_cast<size_t>(gridDim.x) * blockDim.x;
    const size_t gid = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const size_t ngroups = n / VEC;
    if (gid < ngroups) {
        double c0, s0;
        rotor_at(turns0 + static_cast<uint64_t>(gid * VEC) * frac, c0, s0);
        double c1 = c0, s1 = s0;
        if (VEC == 2) rot_step(c1, s1, d_c, d_s);
        for (size_t g = gid; g < ngroups; g += nthreads) {
            if (VEC == 2) {
                float4 x = reinterpret_cast<const float4*>(in)[g];
                float2 a = mix1(make_float2(x.x, x.y), c0, s0);
                float2 b = mix1(make_float2(x.z, x.w), c1, s1);
                reinterpret_cast<float4*>(out)[g] = make_float4(a.x, a.y, b.x, b.y);
                rot_step(c1, s1, sweep_c, sweep_s);
            } else {
                out[g] = mix1(in[g], c0, s0);
            }
            rot_step(c0, s0, sweep_c, sweep_s);
        }
    }
    // odd tail (VEC == 2 only): one thread, exact evaluation
    if (VEC == 2 && gid == 0 && (n & 1)) {
        double c, s;
        rotor_at(turns0 + static_cast<uint64_t>(n - 1) * frac, c, s);
        out[n - 1] = mix1(in[n - 1], c, s);
    }
}

// Mixer::mix::<f64> -- the instantiation the reference's own mixer tests use (src/mixer.rs:160-246, :250-336):
// the casts to and from f64 are identities, so the output is the f64 product itself.  One double2 (16 B) per lane.
__global__ __launch_bounds__(256) void mixer_f64_kernel(const double2* __restrict__ in, double2* __restrict__ out,
                                                        size_t n, uint64_t turns0, uint64_t frac, double sweep_c,
                                                        double sweep_s) {
    const size_t nthreads = static_cast<size_t>(gridDim.x) * blockDim.x;
    const size_t gid = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= n) return;
    double c, s;
    rotor_at(turns0 + static_cast<uint64_t>(gid) * frac, c, s);
    for (size_t g = gid; g < n; g += nthreads) {
        const double2 x = in[g];
        out[g] = make_double2(x.x * c - x.y * s, x.x * s + x.y * c);
        rot_step(c, s, sweep_c, sweep_s);
    }
}

// MixerNode -> DecimateNode in one pass: out[j] = mix(in[j * rate]) with the oscillator phase of sample j * rate --
// only the kept samples are mixed and only their sectors are read (the four-kernel chain with the mixer behind the
// FIR: 40 + 23 us for the two nodes at 2^24 samples and rate 8, 25 us for this).  Same arithmetic as mixer_kernel.
__global__ __launch_bounds__(256) void mix_decimate_kernel(const float2* __restrict__ in, float2* __restrict__ out,
                                                           size_t n_out, size_t rate, uint64_t turns0, uint64_t frac,
                                                           double sweep_c, double sweep_s) {
    const size_t nthreads = static_cast<size_t>(gridDim.x) * blockDim.x;
    const size_t gid = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= n_out) return;
    double c0, s0;
    rotor_at(turns0 + static_cast<uint64_t>(gid * rate) * frac, c0, s0);
    for (size_t j = gid; j < n_out; j += nthreads) {
        out[j] = mix1(in[j * rate], c0, s0);
        rot_step(c0, s0, sweep_c, sweep_s);
    }
}

// =============================================================== decimate / upsample
// Reference: resample_node.rs:53-65 / :120-131.  Byte-exact copies of a Copy type.
template <typename V>
__global__ __launch_bounds__(256) void decimate_kernel(const V* __restrict__ in,
                                                       V* __restrict__ out, size_t n_out,
                                                       size_t rate) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t j = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; j < n_out;
         j += stride)
        out[j] = in[j * rate];
}

template <typename V>
__global__ __launch_bounds__(256) void upsample_kernel(const V* __restrict__ in,
                                                       V* __restrict__ out, size_t n_out,
                                                       size_t rate) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    // one division per thread, then incremental quotient/remainder per sweep
    size_t q = i / rate, r = i - q * rate;
    const size_t sq = stride / rate, sr = stride - sq * rate;
    V zero;
    memset(&zero, 0, sizeof(V));
    for (; i < n_out; i += stride) {
        out[i] = (r == 0) ? in[q] : zero;
        q += sq;
        r += sr;
        if (r >= rate) {
            r -= rate;
            ++q;
        }
    }
}

// The same, 16 output bytes per lane and four chunks in flight per lane: the one-element kernel above is
// bound by the latency of its dependent load -> store pairs (2^24 Complex<f32> outputs at rate 4: 36.6 us
// against 26.5 us for a plain fill of that size + the 33 MB read).  `out` must be 16-byte aligned; the last,
// partial chunk is written element by element.
template <typename V>
__global__ __launch_bounds__(256) void upsample_vec_kernel(const V* __restrict__ in, V* __restrict__ out, size_t n_out,
                                                           size_t rate) {
    constexpr int E = 16 / sizeof(V);  // elements per chunk
    constexpr int U = 4;               // chunks per lane and sweep
    union Chunk {
        uint4 u;
        V v[E];
    };
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    const size_t n_chunks = (n_out + E - 1) / E;
    size_t c = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    // (q, r) = divmod(first element of the chunk, rate): one division per lane, then increments per chunk
    size_t q = (c * E) / rate, r = c * E - q * rate;
    const size_t sq = (stride * E) / rate, sr = stride * E - sq * rate;
    V zero;
    memset(&zero, 0, sizeof(V));
    for (; c < n_chunks; c += U * stride) {
        Chunk w[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            live[u] = c + u * stride < n_chunks;
            if (live[u]) {
                size_t qq = q, rr = r;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    // elements past n_out (last chunk only) read nothing: qq < n there is not guaranteed
                    w[u].v[e] = (rr == 0 && (c + u * stride) * E + e < n_out) ? in[qq] : zero;
                    if (++rr == rate) {
                        rr = 0;
                        ++qq;
                    }
                }
            }
            q += sq;
            r += sr;
            if (r >= rate) {
                r -= rate;
                ++q;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!live[u]) continue;
            const size_t i0 = (c + u * stride) * E;
            if (i0 + E <= n_out) {
                *reinterpret_cast<uint4*>(out + i0) = w[u].u;
            } else {
                for (int e = 0; e < E; ++e)
                    if (i0 + e < n_out) out[i0 + e] = w[u].v[e];
            }
        }
    }
}

// rate >= elements per chunk: a chunk holds at most one input element -- one predicated load and a select per
// chunk instead of a walk over its elements (the walk costs ~100 instructions per chunk and is issue-bound)
template <typename V>
__global__ __launch_bounds__(256) void upsample_sparse_kernel(const V* __restrict__ in, V* __restrict__ out,
                                                              size_t n_out, size_t rate) {
    constexpr int E = 16 / sizeof(V);
    constexpr int U = 4;
    union Chunk {
        uint4 u;
        V v[E];
    };
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    const size_t n_chunks = n_out / E;  // whole chunks; the < E elements after them are written one by one below
    size_t c = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    V zero;
    memset(&zero, 0, sizeof(V));
    if (c == 0)
        for (size_t i = n_chunks * E; i < n_out; ++i) out[i] = (i % rate == 0) ? in[i / rate] : zero;
    if (c >= n_chunks) return;
    size_t q = (c * E) / rate, r = c * E - q * rate;
    const size_t sq = (stride * E) / rate, sr = stride * E - sq * rate;
    for (; c < n_chunks; c += U * stride) {
        V val[U];
        int pos[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t d = r ? rate - r : 0;  // offset of the first multiple of `rate` at or after the chunk's start
            const bool has = c + u * stride < n_chunks && d < E;
            pos[u] = has ? static_cast<int>(d) : -1;
            val[u] = zero;
            if (has) val[u] = in[q + (r ? 1 : 0)];
            q += sq;
            r += sr;
            if (r >= rate) {
                r -= rate;
                ++q;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (c + u * stride >= n_chunks) break;
            Chunk w;
#pragma unroll
            for (int e = 0; e < E; ++e) w.v[e] = (e == pos[u]) ? val[u] : zero;
            *reinterpret_cast<uint4*>(out + (c + u * stride) * E) = w.u;
        }
    }
}

// =============================================================== FM demod
// Reference: FM::demod (src/modulation/analog.rs:22-35):
//   theta = samp * prev.conj();  out = atan2(theta.im, theta.re);  prev = samp.
__device__ inline float fm1(float2 x, float2 p) {
    float pcr = p.x, pci = -p.y;  // conj()
    float re = x.x * pcr - x.y * pci;
    float im = x.x * pci + x.y * pcr;
    return atan2f(im, re);
}

// prev / prev_new: FM.prev before and after this batch (two words of the handle, swapped per call, so that the
// lane that stores the new value can never overtake the lane that reads the old one)
template <bool ALIGNED>
__global__ __launch_bounds__(256) void fmdemod_kernel(const float2* __restrict__ in,
                                                      const float2* __restrict__ prev,
                                                      float2* __restrict__ prev_new,
                                                      float* __restrict__ out, size_t n) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x * 4;
    if (blockIdx.x == 0 && threadIdx.x == 0) prev_new[0] = in[n - 1];
    for (size_t i0 = (static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x) * 4; i0 < n;
         i0 += stride) {
        float2 p = (i0 == 0) ? prev[0] : in[i0 - 1];
        if (ALIGNED && i0 + 4 <= n) {
            float4 a = reinterpret_cast<const float4*>(in + i0)[0];
            float4 b = reinterpret_cast<const float4*>(in + i0)[1];
            float2 x0 = make_float2(a.x, a.y), x1 = make_float2(a.z, a.w);
            float2 x2 = make_float2(b.x, b.y), x3 = make_float2(b.z, b.w);
            float4 o = make_float4(fm1(x0, p), fm1(x1, x0), fm1(x2, x1), fm1(x3, x2));
            *reinterpret_cast<float4*>(out + i0) = o;
        } else {
            size_t end = i0 + 4 < n ? i0 + 4 : n;
            for (size_t i = i0; i < end; ++i) {
                float2 x = in[i];
                out[i] = fm1(x, p);
                p = x;
            }
        }
    }
}

inline unsigned grid_for(size_t work_items, unsigned per_block, unsigned max_blocks) {
    size_t b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return static_cast<unsigned>(b);
}

}  // namespace comms

using namespace comms;

// ------------------------------------------------------------------ mixer handle
struct comms_mixer : Handle {
    double dphase;       // wrapped into [0, T) as Mixer::new does
    uint64_t turns;      // current phase as a fraction of T, 2^-64 units
    uint64_t frac;       // dphase as a fraction of T
};

static uint64_t to_turns(double angle) { return mix_to_turns(angle); }
static void host_rotor(uint64_t turns, double& c, double& s) { mix_host_rotor(turns, c, s); }

extern "C" {

comms_status_t comms_mixer_create(double dphase, double phase, int32_t device,
                                  comms_mixer_t** out) {
    COMMS_ARG(out != nullptr, "out is NULL");
    *out = nullptr;
    COMMS_ARG(std::isfinite(dphase) && std::isfinite(phase), "dphase/phase must be finite");
    comms_mixer* h = new (std::nothrow) comms_mixer;
    COMMS_ARG(h != nullptr, "out of host memory");
    comms_status_t st = h->init(device);
    if (st != COMMS_OK) {
        delete h;
        return st;
    }
    dphase = mix_wrap_dphase(dphase);  // Mixer::new (src/mixer.rs:43-51)
    h->dphase = dphase;
    h->frac = to_turns(dphase);
    h->turns = to_turns(phase);
    *out = h;
    return COMMS_OK;
}

comms_status_t comms_mixer_run_dev(comms_mixer_t* h, const comms_c32* d_in, size_t n,
                                   comms_c32* d_out, void* stream) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG((d_in && d_out) || !n, "NULL device pointer");
    COMMS_TRY(use_device(h->device));
    if (!n) return COMMS_OK;
    hipStream_t s = h->pick(stream);
    const bool vec2 = ((reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_out)) & 15) == 0 && n >= 2;
    const int VEC = vec2 ? 2 : 1;
    unsigned blocks = grid_for(n / VEC, 256, 8 * kNumCU);
    uint64_t sweep = static_cast<uint64_t>(blocks) * 256u * VEC;
    double sc, ss, dc, ds;
    host_rotor(sweep * h->frac, sc, ss);
    host_rotor(h->frac, dc, ds);
    const float2* in = reinterpret_cast<const float2*>(d_in);
    float2* o = reinterpret_cast<float2*>(d_out);
    h->tic(s);
    if (vec2)
        mixer_kernel<2><<<dim3(blocks), dim3(256), 0, s>>>(in, o, n, h->turns, h->frac, sc, ss, dc, ds);
    else
        mixer_kernel<1><<<dim3(blocks), dim3(256), 0, s>>>(in, o, n, h->turns, h->frac, sc, ss, dc, ds);
    h->toc(s);
    COMMS_TRY(launch_ok("mixer_kernel"));
    h->turns += static_cast<uint64_t>(n) * h->frac;
    return COMMS_OK;
}

// internal (chain.hip): MixerNode over n samples followed by DecimateNode(rate), n a multiple of rate
comms_status_t comms_mixer_run_decim_dev(comms_mixer_t* h, const comms_c32* d_in, size_t n, size_t rate,
                                         comms_c32* d_out, void* stream) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG((d_in && d_out) || !n, "NULL device pointer");
    COMMS_ARG(rate >= 1 && n % rate == 0, "n must be a multiple of the rate");
    COMMS_TRY(use_device(h->device));
    if (!n) return COMMS_OK;
    hipStream_t s = h->pick(stream);
    const size_t n_out = n / rate;
    unsigned blocks = grid_for(n_out, 256, 8 * kNumCU);
    double sc, ss;
    host_rotor(static_cast<uint64_t>(blocks) * 256u * rate * h->frac, sc, ss);
    h->tic(s);
    mix_decimate_kernel<<<dim3(blocks), dim3(256), 0, s>>>(reinterpret_cast<const float2*>(d_in), reinterpret_cast<float2*>(d_out),
                                                         n_out, rate, h->turns, h->frac, sc, ss);
    h->toc(s);
    COMMS_TRY(launch_ok("mix_decimate_kernel"));
    h->turns += static_cast<uint64_t>(n) * h->frac;
    return COMMS_OK;
}

comms_status_t comms_mixer_run(comms_mixer_t* h, const comms_c32* in, size_t n, comms_c32* out) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG((in && out) || !n, "NULL host pointer");
    COMMS_TRY(use_device(h->device));
    if (!n) return COMMS_OK;
    return h->run_host_units(in, n * sizeof(comms_c32), sizeof(comms_c32), out, n * sizeof(comms_c32), sizeof(comms_c32), [&](void* d_in, void* d_out, size_t ib, size_t) {
        return comms_mixer_run_dev(h, static_cast<const comms_c32*>(d_in), ib / sizeof(comms_c32), static_cast<comms_c32*>(d_out), COMMS_STREAM_HANDLE);
    });
}

// MixerNode<f64> (src/mixer.rs:93-148 with T = f64): same handle, same oscillator, Complex<f64> samples
comms_status_t comms_mixer_run_f64_dev(comms_mixer_t* h, const comms_c64* d_in, size_t n, comms_c64* d_out,
                                       void* stream) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG((d_in && d_out) || !n, "NULL device pointer");
    COMMS_ARG(((reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_out)) & 15) == 0,
              "Complex<f64> streams must be 16-byte aligned");
    COMMS_TRY(use_device(h->device));
    if (!n) return COMMS_OK;
    hipStream_t s = h->pick(stream);
    unsigned blocks = grid_for(n, 256, 8 * kNumCU);
    double sc, ss;
    host_rotor(static_cast<uint64_t>(blocks) * 256u * h->frac, sc, ss);
    h->tic(s);
    mixer_f64_kernel<<<dim3(blocks), dim3(256), 0, s>>>(reinterpret_cast<const double2*>(d_in),
                                                       reinterpret_cast<double2*>(d_out), n, h->turns, h->frac, sc, ss);
    h->toc(s);
    COMMS_TRY(launch_ok("mixer_f64_kernel"));
    h->turns += static_cast<uint64_t>(n) * h->frac;
    return COMMS_OK;
}
comms_status_t comms_mixer_run_f64(comms_mixer_t* h, const comms_c64* in, size_t n, comms_c64* out) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG((in && out) || !n, "NULL host pointer");
    COMMS_TRY(use_device(h->device));
    if (!n) return COMMS_OK;
    return h->run_host_units(in, n * sizeof(comms_c64), sizeof(comms_c64), out, n * sizeof(comms_c64), sizeof(comms_c64), [&](void* d_in, void* d_out, size_t ib, size_t) {
        return comms_mixer_run_f64_dev(h, static_cast<const comms_c64*>(d_in), ib / sizeof(comms_c64), static_cast<comms_c64*>(d_out), COMMS_STREAM_HANDLE);
    });
}

comms_status_t comms_mixer_get_phase(const comms_mixer_t* h, double* out_phase) {
    COMMS_ARG(h && out_phase, "NULL argument");
    *out_phase = static_cast<double>(h->turns >> 11) * (kT * 0x1.0p-53);
    return COMMS_OK;
}

// Checkpoint / shard hand-over hook: the oscillator phase of the NEXT sample (any finite angle;
// reduced modulo fl(2*pi) as Mixer::new leaves `phase` and the per-sample wrap then treats it).
comms_status_t comms_mixer_set_phase(comms_mixer_t* h, double phase) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG(std::isfinite(phase), "phase must be finite");
    h->turns = to_turns(phase);
    return COMMS_OK;
}

comms_status_t comms_mixer_set_timer(comms_mixer_t* h, comms_timer_t* t) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    h->timer = t;
    return COMMS_OK;
}

comms_status_t comms_mixer_destroy(comms_mixer_t* h) {
    if (!h) return COMMS_OK;
    (void)use_device(h->device);
    h->fini();
    delete h;
    return COMMS_OK;
}

// ------------------------------------------------------------------ decimate / upsample
comms_status_t comms_decimate_out_len(size_t n, size_t rate, size_t* out_n) {
    COMMS_ARG(out_n != nullptr, "out_n is NULL");
    *out_n = (rate <= 1) ? n : (n + rate - 1) / rate;
    return COMMS_OK;
}
comms_status_t comms_upsample_out_len(size_t n, size_t rate, size_t* out_n) {
    COMMS_ARG(out_n != nullptr, "out_n is NULL");
    if (rate <= 1) {
        *out_n = n;
        return COMMS_OK;
    }
    COMMS_ARG(n <= SIZE_MAX / rate, "n * rate overflows");
    *out_n = n * rate;
    return COMMS_OK;
}

static bool elem_ok(size_t e) { return e == 1 || e == 2 || e == 4 || e == 8 || e == 16; }

#define COMMS_RESAMPLE_DISPATCH(KERNEL, elem, ...)                                              \
    switch (elem) {                                                                             \
        case 1: KERNEL<uint8_t><<<dim3(blocks), dim3(256), 0, s>>>(static_cast<const uint8_t*>(d_in), static_cast<uint8_t*>(d_out), __VA_ARGS__); break;   \
        case 2: KERNEL<uint16_t><<<dim3(blocks), dim3(256), 0, s>>>(static_cast<const uint16_t*>(d_in), static_cast<uint16_t*>(d_out), __VA_ARGS__); break; \
        case 4: KERNEL<uint32_t><<<dim3(blocks), dim3(256), 0, s>>>(static_cast<const uint32_t*>(d_in), static_cast<uint32_t*>(d_out), __VA_ARGS__); break; \
        case 8: KERNEL<uint2><<<dim3(blocks), dim3(256), 0, s>>>(static_cast<const uint2*>(d_in), static_cast<uint2*>(d_out), __VA_ARGS__); break;         \
        default: KERNEL<uint4><<<dim3(blocks), dim3(256), 0, s>>>(static_cast<const uint4*>(d_in), static_cast<uint4*>(d_out), __VA_ARGS__); break;        \
    }

comms_status_t comms_decimate_run_dev(const void* d_in, size_t n, size_t elem, size_t rate,
                                      void* d_out, size_t* out_n, int32_t device, void* stream) {
    COMMS_ARG(elem_ok(elem), "elem must be 1, 2, 4, 8 or 16 bytes (got %zu)", elem);
    COMMS_ARG((d_in && d_out) || !n, "NULL device pointer");
    COMMS_ARG(((reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_out)) % elem) == 0,
              "pointers must be aligned to elem");
    size_t n_out = 0;
    COMMS_TRY(comms_decimate_out_len(n, rate, &n_out));
    if (out_n) *out_n = n_out;
    COMMS_TRY(use_device(device));
    if (!n_out) return COMMS_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (rate <= 1) {
        if (d_in != d_out)
            COMMS_HIP_TRY(hipMemcpyAsync(d_out, d_in, n * elem, hipMemcpyDefault, s));
        return COMMS_OK;
    }
    COMMS_ARG(!ranges_overlap(d_in, n * elem, d_out, n_out * elem), "decimate cannot run in place");
    unsigned blocks = grid_for(n_out, 256, 8 * kNumCU);
    COMMS_RESAMPLE_DISPATCH(decimate_kernel, elem, n_out, rate)
    return launch_ok("decimate_kernel");
}

comms_status_t comms_upsample_run_dev(const void* d_in, size_t n, size_t elem, size_t rate,
                                      void* d_out, size_t* out_n, int32_t device, void* stream) {
    COMMS_ARG(elem_ok(elem), "elem must be 1, 2, 4, 8 or 16 bytes (got %zu)", elem);
    COMMS_ARG((d_in && d_out) || !n, "NULL device pointer");
    COMMS_ARG(((reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_out)) % elem) == 0,
              "pointers must be aligned to elem");
    size_t n_out = 0;
    COMMS_TRY(comms_upsample_out_len(n, rate, &n_out));
    if (out_n) *out_n = n_out;
    COMMS_TRY(use_device(device));
    if (!n_out) return COMMS_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (rate <= 1) {
        if (d_in != d_out)
            COMMS_HIP_TRY(hipMemcpyAsync(d_out, d_in, n * elem, hipMemcpyDefault, s));
        return COMMS_OK;
    }
    COMMS_ARG(!ranges_overlap(d_in, n * elem, d_out, n_out * elem), "upsample cannot run in place");
    if (elem < 16 && reinterpret_cast<uintptr_t>(d_out) % 16 == 0) {  // 16 output bytes per lane
        unsigned blocks = grid_for((n_out * elem + 15) / 16, 256, 8 * kNumCU);
        if (rate >= 16 / elem) {
            COMMS_RESAMPLE_DISPATCH(upsample_sparse_kernel, elem, n_out, rate)
            return launch_ok("upsample_sparse_kernel");
        }
        COMMS_RESAMPLE_DISPATCH(upsample_vec_kernel, elem, n_out, rate)
        return launch_ok("upsample_vec_kernel");
    }
    unsigned blocks = grid_for(n_out, 256, 8 * kNumCU);
    COMMS_RESAMPLE_DISPATCH(upsample_kernel, elem, n_out, rate)
    return launch_ok("upsample_kernel");
}

static comms_status_t resample_host(bool up, const void* in, size_t n, size_t elem, size_t rate,
                                    void* out, size_t* out_n, int32_t device) {
    COMMS_ARG(elem_ok(elem), "elem must be 1, 2, 4, 8 or 16 bytes (got %zu)", elem);
    COMMS_ARG((in && out) || !n, "NULL host pointer");
    size_t n_out = 0;
    COMMS_TRY(up ? comms_upsample_out_len(n, rate, &n_out) : comms_decimate_out_len(n, rate, &n_out));
    if (out_n) *out_n = n_out;
    COMMS_TRY(use_device(device));
    if (!n_out) return COMMS_OK;
    Handle* h = nullptr;  // these two nodes have no handle in the C ABI
    COMMS_TRY(thread_handle(device, &h));
    // (units: one element in, `rate` out -- or `rate` in, one out: a chunk of whole units keeps the batch's indexing,
    // src/util/resample_node.rs:53-65,120-131; the ragged tail of a decimated batch is the last chunk's)
    const size_t r = rate < 1 ? 1 : rate;
    return h->run_host_units(in, n * elem, up ? elem : r * elem, out, n_out * elem, up ? r * elem : elem, [&](void* d_in, void* d_out, size_t ib, size_t) {
        return up ? comms_upsample_run_dev(d_in, ib / elem, elem, rate, d_out, nullptr, device, h->stream)
                  : comms_decimate_run_dev(d_in, ib / elem, elem, rate, d_out, nullptr, device, h->stream);
    });
}

comms_status_t comms_decimate_run(const void* in, size_t n, size_t elem, size_t rate, void* out,
                                  size_t* out_n, int32_t device) {
    return resample_host(false, in, n, elem, rate, out, out_n, device);
}
comms_status_t comms_upsample_run(const void* in, size_t n, size_t elem, size_t rate, void* out,
                                  size_t* out_n, int32_t device) {
    return resample_host(true, in, n, elem, rate, out, out_n, device);
}

}  // extern "C"

// ------------------------------------------------------------------ FM demod handle
struct comms_fmdemod : Handle {
    float2* d_prev = nullptr;  // FM.prev (analog.rs:9), starts 0+0i: two words, d_prev[cur] is the current one
    int cur = 0;
};

extern "C" {

comms_status_t comms_fmdemod_create(int32_t device, comms_fmdemod_t** out) {
    COMMS_ARG(out != nullptr, "out is NULL");
    *out = nullptr;
    comms_fmdemod* h = new (std::nothrow) comms_fmdemod;
    COMMS_ARG(h != nullptr, "out of host memory");
    comms_status_t st = h->init(device);
    if (st != COMMS_OK) {
        delete h;
        return st;
    }
    hipError_t e = hipMalloc(&h->d_prev, 2 * sizeof(float2));
    if (e == hipSuccess) e = zero_device(h->d_prev, 2 * sizeof(float2));
    if (e != hipSuccess) {
        h->fini();
        delete h;
        return fail(COMMS_ERR_DEVICE, "fmdemod state alloc: %s", hipGetErrorString(e));
    }
    *out = h;
    return COMMS_OK;
}

comms_status_t comms_fmdemod_run_dev(comms_fmdemod_t* h, const comms_c32* d_in, size_t n,
                                     float* d_out, void* stream) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG((d_in && d_out) || !n, "NULL device pointer");
    COMMS_TRY(use_device(h->device));
    if (!n) return COMMS_OK;
    COMMS_ARG(!ranges_overlap(d_in, n * 8, d_out, n * 4), "fmdemod cannot run in place");
    hipStream_t s = nullptr;
    COMMS_TRY(h->enter(stream, &s));  // d_prev is read and advanced in stream order
    const float2* in = reinterpret_cast<const float2*>(d_in);
    unsigned blocks = grid_for((n + 3) / 4, 256, 8 * kNumCU);
    bool aligned = ((reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_out)) & 15) == 0;
    h->tic(s);
    if (aligned)
        fmdemod_kernel<true><<<dim3(blocks), dim3(256), 0, s>>>(in, h->d_prev + h->cur, h->d_prev + (h->cur ^ 1), d_out, n);
    else
        fmdemod_kernel<false><<<dim3(blocks), dim3(256), 0, s>>>(in, h->d_prev + h->cur, h->d_prev + (h->cur ^ 1), d_out, n);
    h->toc(s);
    COMMS_TRY(launch_ok("fmdemod_kernel"));
    h->cur ^= 1;  // the kernel stored the batch's last sample in the other word (no separate copy command)
    return COMMS_OK;
}

comms_status_t comms_fmdemod_run(comms_fmdemod_t* h, const comms_c32* in, size_t n, float* out) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG((in && out) || !n, "NULL host pointer");
    COMMS_TRY(use_device(h->device));
    if (!n) return COMMS_OK;
    return h->run_host_units(in, n * sizeof(comms_c32), sizeof(comms_c32), out, n * sizeof(float), sizeof(float), [&](void* d_in, void* d_out, size_t ib, size_t) {
        return comms_fmdemod_run_dev(h, static_cast<const comms_c32*>(d_in), ib / sizeof(comms_c32), static_cast<float*>(d_out), COMMS_STREAM_HANDLE);
    });
}

// FM.prev (src/modulation/analog.rs:9,31): the last input sample of the previous batch.  Getter and
// setter are the checkpoint hook and the 1-sample halo of a sharded stream.
comms_status_t comms_fmdemod_get_prev(comms_fmdemod_t* h, comms_c32* out_prev) {
    COMMS_ARG(h && out_prev, "NULL argument");
    COMMS_TRY(use_device(h->device));
    COMMS_TRY(h->quiesce());
    COMMS_HIP_TRY(hipMemcpy(out_prev, h->d_prev + h->cur, sizeof(float2), hipMemcpyDeviceToHost));
    return COMMS_OK;
}

comms_status_t comms_fmdemod_set_prev(comms_fmdemod_t* h, const comms_c32* prev) {
    COMMS_ARG(h && prev, "NULL argument");
    COMMS_TRY(use_device(h->device));
    COMMS_TRY(h->quiesce());
    COMMS_HIP_TRY(hipMemcpy(h->d_prev + h->cur, prev, sizeof(float2), hipMemcpyHostToDevice));
    return COMMS_OK;
}

comms_status_t comms_fmdemod_set_timer(comms_fmdemod_t* h, comms_timer_t* t) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    h->timer = t;
    return COMMS_OK;
}

comms_status_t comms_fmdemod_destroy(comms_fmdemod_t* h) {
    if (!h) return COMMS_OK;
    (void)use_device(h->device);
    if (h->d_prev) (void)hipFree(h->d_prev);
    h->fini();
    delete h;
    return COMMS_OK;
}

}  // extern "C"
