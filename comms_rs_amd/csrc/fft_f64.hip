// fft_f64.hip -- FFTBatchNode<f64> / FFTSampleNode<f64> and FMDemodNode<f64>.
//
// The reference's FFT nodes are generic over the sample type and compute in f64 whatever it is (src/fft/mod.rs:73-96:
// every input is converted to Complex<f64>, rustfft runs on f64, the result is cast back); its own doc examples
// instantiate FFTBatchNode<f64> / FFTSampleNode<f64> (src/fft/fft_node.rs:24,99).  FM::demod is generic over the float
// type too (src/modulation/analog.rs:8-48).  Every BASELINE config is f32 -- those are the tuned kernels of fft.hip and
// pointwise.hip -- but a graph that carries f64 samples must find its nodes (round 5; until then such graphs stayed on the
// CPU).  These are plain FP64 kernels, correct to f64 rounding (the tests hold them to 1e-12 relative), not roofline claims:
//   * powers of two up to 4096 points: one pass through LDS (bit-reversed load, log2 N radix-2 layers), several
//     transforms per workgroup;
//   * powers of two up to 2^24: four-step -- N2 column transforms of N1 points read in pieces of sixteen adjacent
//     columns, x W_N^{n2 k1} (two table look-ups and one product), rows of N2 points, transposed store;
//   * other lengths up to 4096: the DFT sum itself, inputs in LDS, roots from a table walked by (j k) mod N;
//   * other lengths above: Bluestein's chirp transform on the power-of-two kernels.
// Unnormalised, forward = e^{-2 pi i jk/N}, inverse = e^{+2 pi i jk/N} (rustfft 2.1.0 semantics), like comms_fft_*.
#include <cmath>
#include <vector>

#include "common.hpp"

namespace comms {

constexpr int F64_LDS_POINTS = 4096;  // double2 elements per workgroup (64 KiB)

__device__ __forceinline__ double2 zmul(double2 a, double2 b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 zconj_if(double2 a, bool c) { return c ? make_double2(a.x, -a.y) : a; }

struct F64Pass {
    const double2* in;
    double2* out;
    size_t in_so, in_st, in_sj;     // element (y, t, j) of the input at in + y in_so + t in_st + j in_sj
    size_t out_so, out_st, out_sk;  // output (y, t, k) at out + y out_so + t out_st + k out_sk
    unsigned T, N, logN, B;         // T transforms of N points per y; B of them per workgroup
    const double2* t4096;           // forward roots e^{-2 pi i e / 4096}
    const double2* twa;             // four-step twiddle of the column pass: W_P^{e} = twa[e >> h] * twb[e & (2^h - 1)], or null
    const double2* twb;
    unsigned h;
    int inverse;
};

// B transforms of N <= 4096 / B points through LDS: bit-reversed placement, log2 N in-place radix-2 layers.
__global__ __launch_bounds__(256) void fft_f64_lds_kernel(const F64Pass a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double2* buf = reinterpret_cast<double2*>(smem);
    const unsigned N = a.N, B = a.B, t0 = blockIdx.x * B;
    const unsigned nb = a.T - t0 < B ? a.T - t0 : B;
    const double2* in = a.in + static_cast<size_t>(blockIdx.y) * a.in_so;
    double2* out = a.out + static_cast<size_t>(blockIdx.y) * a.out_so;
    const bool inv = a.inverse != 0;
    // load: the unit-stride index runs fastest over the threads
    const bool in_t_fast = a.in_st == 1 && a.in_sj != 1;
    for (unsigned i = threadIdx.x; i < nb * N; i += 256) {
        const unsigned b = in_t_fast ? i % nb : i / N, j = in_t_fast ? i / nb : i % N;
        const unsigned r = a.logN ? __brev(j) >> (32 - a.logN) : 0u;
        buf[b * N + r] = in[static_cast<size_t>(t0 + b) * a.in_st + static_cast<size_t>(j) * a.in_sj];
    }
    __syncthreads();
    for (unsigned s = 0; s < a.logN; ++s) {
        const unsigned half = 1u << s;
        for (unsigned id = threadIdx.x; id < nb * (N >> 1); id += 256) {
            const unsigned b = id / (N >> 1), r = id % (N >> 1);
            const unsigned pos = r & (half - 1), i0 = b * N + ((r >> s) << (s + 1)) + pos, i1 = i0 + half;
            const double2 w = zconj_if(a.t4096[pos << (11 - s)], inv);
            const double2 u = buf[i0], v = zmul(buf[i1], w);
            buf[i0] = make_double2(u.x + v.x, u.y + v.y);
            buf[i1] = make_double2(u.x - v.x, u.y - v.y);
        }
        __syncthreads();
    }
    const bool out_t_fast = a.out_st == 1 && a.out_sk != 1;
    for (unsigned i = threadIdx.x; i < nb * N; i += 256) {
        const unsigned b = out_t_fast ? i % nb : i / N, k = out_t_fast ? i / nb : i % N;
        double2 x = buf[b * N + k];
        if (a.twa) {  // column pass of the four-step form: x W_P^{t k}
            const size_t e = static_cast<size_t>(t0 + b) * k;
            const double2 w = zmul(a.twa[e >> a.h], a.twb[e & ((static_cast<size_t>(1) << a.h) - 1)]);
            x = zmul(x, zconj_if(w, inv));
        }
        out[static_cast<size_t>(t0 + b) * a.out_st + static_cast<size_t>(k) * a.out_sk] = x;
    }
}

// The DFT sum for lengths that are not powers of two (N <= 4096): inputs in LDS, roots[(j k) mod N].
__global__ __launch_bounds__(256) void dft_f64_direct_kernel(const double2* __restrict__ in, double2* __restrict__ out, unsigned N,
                                                             const double2* __restrict__ roots, int inverse) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double2* x = reinterpret_cast<double2*>(smem);
    const double2* src = in + static_cast<size_t>(blockIdx.y) * N;
    for (unsigned i = threadIdx.x; i < N; i += 256) x[i] = src[i];
    __syncthreads();
    const unsigned k = blockIdx.x * 256 + threadIdx.x;
    if (k >= N) return;
    double re = 0.0, im = 0.0;
    unsigned idx = 0;
    for (unsigned j = 0; j < N; ++j) {
        const double2 w = zconj_if(roots[idx], inverse != 0);
        const double2 p = zmul(x[j], w);
        re += p.x;
        im += p.y;
        idx += k;
        if (idx >= N) idx -= N;
    }
    out[static_cast<size_t>(blockIdx.y) * N + k] = make_double2(re, im);
}

// Bluestein: a[n] = x[n] c[n] (n < N), 0 up to M;  A *= S;  X[k] = a'[k] c[k]
__global__ __launch_bounds__(256) void blue_in_kernel(const double2* __restrict__ x, const double2* __restrict__ c, double2* __restrict__ a,
                                                      size_t N, size_t M, int inverse) {
    const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= M) return;
    a[i] = i < N ? zmul(x[i], zconj_if(c[i], inverse != 0)) : make_double2(0.0, 0.0);
}
__global__ __launch_bounds__(256) void blue_mul_kernel(double2* __restrict__ a, const double2* __restrict__ s, size_t M, int inverse) {
    const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= M) return;
    // the spectrum of the conjugate chirp is the conjugate of the mirrored spectrum; the chirp is even, so conj(S[i]) serves
    a[i] = zmul(a[i], inverse ? make_double2(s[i].x, -s[i].y) : s[i]);
}
__global__ __launch_bounds__(256) void blue_out_kernel(const double2* __restrict__ a, const double2* __restrict__ c, double2* __restrict__ X,
                                                       size_t N, int inverse) {
    const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= N) return;
    X[i] = zmul(a[i], zconj_if(c[i], inverse != 0));
}

// FM::demod on Complex<f64> (src/modulation/analog.rs:22-35): theta = samp * prev.conj(); out = atan2(theta.im, theta.re)
__global__ __launch_bounds__(256) void fmdemod_f64_kernel(const double2* __restrict__ in, const double2* __restrict__ prev,
                                                          double2* __restrict__ prev_new, double* __restrict__ out, size_t n) {
    if (blockIdx.x == 0 && threadIdx.x == 0) prev_new[0] = in[n - 1];
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double2 x = in[i], p = i ? in[i - 1] : prev[0];
        const double pcr = p.x, pci = -p.y;
        out[i] = atan2(x.x * pci + x.y * pcr, x.x * pcr - x.y * pci);
    }
}

}  // namespace comms

using namespace comms;

struct comms_fft_f64 : Handle {
    size_t N = 0;
    bool inverse = false;
    int kind = 0;               // 0: one pass, 1: four-step, 2: DFT sum, 3: Bluestein
    size_t P = 0;               // the power-of-two length the passes run at (N, or Bluestein's M)
    unsigned logP = 0, logN1 = 0, logN2 = 0, h = 0;
    double2* d_t4096 = nullptr;
    double2* d_twa = nullptr;   // W_P^{e 2^h}
    double2* d_twb = nullptr;   // W_P^{e}, e < 2^h
    double2* d_roots = nullptr; // kind 2: W_N^e;  kind 3: the chirp c[n] = e^{-i pi n^2 / N}
    double2* d_spec = nullptr;  // kind 3: FFT_M of the wrapped conjugate chirp, / M
    Scratch s1, s2;
};

namespace {

const long double kPiL = 3.14159265358979323846264338327950288L;
double2 rootl(unsigned long long e, unsigned long long denom) {  // e^{-2 pi i e / denom}
    e %= denom;
    const long double a = 2.0L * kPiL * static_cast<long double>(e) / static_cast<long double>(denom);
    return make_double2(static_cast<double>(cosl(a)), static_cast<double>(-sinl(a)));
}
comms_status_t upload(const std::vector<double2>& v, double2** d) {
    COMMS_HIP_TRY(hipMalloc(d, v.size() * sizeof(double2)));
    COMMS_HIP_TRY(hipMemcpy(*d, v.data(), v.size() * sizeof(double2), hipMemcpyHostToDevice));
    return COMMS_OK;
}

comms_status_t lds_pass(const F64Pass& p, unsigned ny, hipStream_t s) {
    static DeviceOnce attr_once;
    if (attr_once.need())
        COMMS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fft_f64_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          static_cast<int>(F64_LDS_POINTS * sizeof(double2))));
    fft_f64_lds_kernel<<<dim3((p.T + p.B - 1) / p.B, ny), dim3(256), static_cast<size_t>(p.B) * p.N * sizeof(double2), s>>>(p);
    return launch_ok("fft_f64_lds_kernel");
}

// `batch` transforms of P = 2^logP points, back to back in `in`, to `out` (may be `in`); `inverse` overrides the handle's
// direction (Bluestein runs one transform each way)
comms_status_t pow2_run(comms_fft_f64* h, const double2* in, double2* out, size_t batch, bool inverse, hipStream_t s) {
    F64Pass p{};
    p.t4096 = h->d_t4096;
    p.inverse = inverse ? 1 : 0;
    if (h->logP <= 12) {
        p.in = in, p.out = out;
        p.N = static_cast<unsigned>(h->P), p.logN = h->logP;
        p.B = F64_LDS_POINTS / p.N < 16 ? F64_LDS_POINTS / p.N : 16;
        p.in_sj = p.out_sk = 1;
        p.in_st = p.out_st = h->P;
        for (size_t b0 = 0; b0 < batch; b0 += 1u << 20) {  // (grid.x in pieces: T is 32 bits)
            const size_t nb = batch - b0 < (1u << 20) ? batch - b0 : (1u << 20);
            p.in = in + b0 * h->P, p.out = out + b0 * h->P, p.T = static_cast<unsigned>(nb);
            COMMS_TRY(lds_pass(p, 1, s));
        }
        return COMMS_OK;
    }
    const size_t N1 = static_cast<size_t>(1) << h->logN1, N2 = static_cast<size_t>(1) << h->logN2;
    COMMS_TRY(h->s1.reserve(h->P * sizeof(double2) * (batch < 64 ? batch : 64)));
    double2* tmp = static_cast<double2*>(h->s1.p);
    for (size_t b0 = 0; b0 < batch; b0 += 64) {
        const unsigned ny = static_cast<unsigned>(batch - b0 < 64 ? batch - b0 : 64);
        // columns: N2 transforms of N1 points, x[n1 N2 + n2] -> tmp[k1 N2 + n2] x W_P^{n2 k1}
        F64Pass a = p;
        a.in = in + b0 * h->P, a.out = tmp;
        a.in_so = a.out_so = h->P;
        a.T = static_cast<unsigned>(N2), a.N = static_cast<unsigned>(N1), a.logN = h->logN1;
        a.B = F64_LDS_POINTS / a.N < 16 ? F64_LDS_POINTS / a.N : 16;
        a.in_st = 1, a.in_sj = N2, a.out_st = 1, a.out_sk = N2;
        a.twa = h->d_twa, a.twb = h->d_twb, a.h = h->h;
        COMMS_TRY(lds_pass(a, ny, s));
        // rows: N1 transforms of N2 points, tmp[k1 N2 + n2] -> out[k1 + N1 k2]
        F64Pass r = p;
        r.in = tmp, r.out = out + b0 * h->P;
        r.in_so = r.out_so = h->P;
        r.T = static_cast<unsigned>(N1), r.N = static_cast<unsigned>(N2), r.logN = h->logN2;
        r.B = F64_LDS_POINTS / r.N < 16 ? F64_LDS_POINTS / r.N : 16;
        r.in_st = N2, r.in_sj = 1, r.out_st = 1, r.out_sk = N1;
        COMMS_TRY(lds_pass(r, ny, s));
    }
    return COMMS_OK;
}

void free_fft_f64(comms_fft_f64* h) {
    (void)use_device(h->device);
    for (double2* q : {h->d_t4096, h->d_twa, h->d_twb, h->d_roots, h->d_spec})
        if (q) (void)hipFree(q);
    h->s1.release();
    h->s2.release();
    h->fini();
    delete h;
}

comms_status_t fft_f64_prepare(comms_fft_f64* h) {
    const size_t N = h->N;
    const bool pow2 = (N & (N - 1)) == 0;
    h->kind = pow2 ? (N <= 4096 ? 0 : 1) : (N <= 4096 ? 2 : 3);
    if (h->kind == 2) {
        std::vector<double2> r(N);
        for (size_t e = 0; e < N; ++e) r[e] = rootl(e, N);
        return upload(r, &h->d_roots);
    }
    size_t P = N;
    if (h->kind == 3) {
        P = 1;
        while (P < 2 * N - 1) P <<= 1;
    }
    COMMS_ARG(P <= (static_cast<size_t>(1) << 24), "FFT<f64>: %zu points need a %zu-point transform; the limit is 2^24", N, P);
    h->P = P;
    while ((static_cast<size_t>(1) << h->logP) < P) ++h->logP;
    {
        std::vector<double2> t(4096);
        for (unsigned e = 0; e < 4096; ++e) t[e] = rootl(e, 4096);
        COMMS_TRY(upload(t, &h->d_t4096));
    }
    if (h->logP > 12) {
        h->logN1 = (h->logP + 1) / 2, h->logN2 = h->logP - h->logN1;
        h->h = h->logP / 2;
        std::vector<double2> a((P >> h->h) + 1), b(static_cast<size_t>(1) << h->h);
        for (size_t e = 0; e < a.size(); ++e) a[e] = rootl(static_cast<unsigned long long>(e) << h->h, P);
        for (size_t e = 0; e < b.size(); ++e) b[e] = rootl(e, P);
        COMMS_TRY(upload(a, &h->d_twa));
        COMMS_TRY(upload(b, &h->d_twb));
    }
    if (h->kind == 3) {
        // chirp c[n] = e^{-i pi n^2 / N} = root(n^2 mod 2N, 2N); S = FFT_P(conj chirp, wrapped) / P
        std::vector<double2> c(N), w(P, make_double2(0.0, 0.0));
        for (size_t n = 0; n < N; ++n) {
            const unsigned long long q = (static_cast<unsigned long long>(n) * n) % (2ull * N);
            c[n] = rootl(q, 2ull * N);
            const double2 cc = make_double2(c[n].x / static_cast<double>(P), -c[n].y / static_cast<double>(P));
            w[n] = cc;
            if (n) w[P - n] = cc;
        }
        COMMS_TRY(upload(c, &h->d_roots));
        COMMS_TRY(upload(w, &h->d_spec));
        COMMS_TRY(pow2_run(h, h->d_spec, h->d_spec, 1, false, h->stream));
        COMMS_HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return COMMS_OK;
}

}  // namespace

extern "C" {

comms_status_t comms_fft_f64_create(size_t fft_size, int32_t inverse, int32_t device, comms_fft_f64_t** out) {
    COMMS_ARG(out != nullptr, "out is NULL");
    *out = nullptr;
    COMMS_ARG(fft_size >= 1, "fft_size must be >= 1");
    comms_fft_f64* h = new (std::nothrow) comms_fft_f64;
    COMMS_ARG(h != nullptr, "out of host memory");
    comms_status_t st = h->init(device);
    if (st != COMMS_OK) {
        delete h;
        return st;
    }
    h->N = fft_size;
    h->inverse = inverse != 0;
    st = fft_f64_prepare(h);
    if (st != COMMS_OK) {
        free_fft_f64(h);
        return st;
    }
    *out = h;
    return COMMS_OK;
}

comms_status_t comms_fft_f64_run_dev(comms_fft_f64_t* h, const comms_c64* d_in, size_t n, comms_c64* d_out, void* stream) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG((d_in && d_out) || !n, "NULL device pointer");
    // run_fft panics on any other length (src/fft/mod.rs:88: copy_from_slice of unequal lengths)
    COMMS_ARG(n % h->N == 0 && n >= h->N, "input length %zu is not a multiple of fft_size %zu", n, h->N);
    COMMS_TRY(use_device(h->device));
    const bool same = static_cast<const void*>(d_in) == static_cast<const void*>(d_out);
    COMMS_ARG(same || !ranges_overlap(d_in, n * 16, d_out, n * 16), "input and output overlap");
    COMMS_ARG(!(same && h->kind == 2), "the direct DFT (lengths that are not powers of two, up to 4096) cannot run in place");
    hipStream_t s = nullptr;
    COMMS_TRY(h->enter(stream, &s));
    const double2* in = reinterpret_cast<const double2*>(d_in);
    double2* o = reinterpret_cast<double2*>(d_out);
    const size_t batch = n / h->N;
    h->tic(s);
    if (h->kind <= 1) {
        COMMS_TRY(pow2_run(h, in, o, batch, h->inverse, s));
    } else if (h->kind == 2) {
        const unsigned N = static_cast<unsigned>(h->N);
        static DeviceOnce attr_once;
        if (attr_once.need())
            COMMS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&dft_f64_direct_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              static_cast<int>(F64_LDS_POINTS * sizeof(double2))));
        for (size_t b0 = 0; b0 < batch; b0 += 32768) {
            const unsigned nb = static_cast<unsigned>(batch - b0 < 32768 ? batch - b0 : 32768);
            dft_f64_direct_kernel<<<dim3((N + 255) / 256, nb), dim3(256), static_cast<size_t>(N) * sizeof(double2), s>>>(
                in + b0 * N, o + b0 * N, N, h->d_roots, h->inverse ? 1 : 0);
            COMMS_TRY(launch_ok("dft_f64_direct_kernel"));
        }
    } else {
        COMMS_TRY(h->s2.reserve(h->P * sizeof(double2)));
        double2* a = static_cast<double2*>(h->s2.p);
        const unsigned gm = static_cast<unsigned>((h->P + 255) / 256), gn = static_cast<unsigned>((h->N + 255) / 256);
        const int inv = h->inverse ? 1 : 0;
        for (size_t b = 0; b < batch; ++b) {
            blue_in_kernel<<<dim3(gm), dim3(256), 0, s>>>(in + b * h->N, h->d_roots, a, h->N, h->P, inv);
            COMMS_TRY(pow2_run(h, a, a, 1, false, s));
            blue_mul_kernel<<<dim3(gm), dim3(256), 0, s>>>(a, h->d_spec, h->P, inv);
            COMMS_TRY(pow2_run(h, a, a, 1, true, s));
            blue_out_kernel<<<dim3(gn), dim3(256), 0, s>>>(a, h->d_roots, o + b * h->N, h->N, inv);
            COMMS_TRY(launch_ok("blue_out_kernel"));
        }
    }
    h->toc(s);
    return COMMS_OK;
}

comms_status_t comms_fft_f64_run(comms_fft_f64_t* h, const comms_c64* in, size_t n, comms_c64* out) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG((in && out) || !n, "NULL host pointer");
    COMMS_ARG(n % h->N == 0 && n >= h->N, "input length %zu is not a multiple of fft_size %zu", n, h->N);
    COMMS_TRY(use_device(h->device));
    return h->run_host_units(in, n * sizeof(comms_c64), h->N * sizeof(comms_c64), out, n * sizeof(comms_c64), h->N * sizeof(comms_c64),
                             [&](void* d_in, void* d_out, size_t ib, size_t) {
                                 return comms_fft_f64_run_dev(h, static_cast<const comms_c64*>(d_in), ib / sizeof(comms_c64),
                                                              static_cast<comms_c64*>(d_out), COMMS_STREAM_HANDLE);
                             });
}

comms_status_t comms_fft_f64_destroy(comms_fft_f64_t* h) {
    if (!h) return COMMS_OK;
    free_fft_f64(h);
    return COMMS_OK;
}

}  // extern "C"

// ------------------------------------------------------------------ FMDemodNode<f64>
struct comms_fmdemod_f64 : Handle {
    double2* d_prev = nullptr;  // FM.prev (analog.rs:9), starts 0+0i: two words, d_prev[cur] is the current one
    int cur = 0;
};

extern "C" {

comms_status_t comms_fmdemod_f64_create(int32_t device, comms_fmdemod_f64_t** out) {
    COMMS_ARG(out != nullptr, "out is NULL");
    *out = nullptr;
    comms_fmdemod_f64* h = new (std::nothrow) comms_fmdemod_f64;
    COMMS_ARG(h != nullptr, "out of host memory");
    comms_status_t st = h->init(device);
    if (st != COMMS_OK) {
        delete h;
        return st;
    }
    hipError_t e = hipMalloc(&h->d_prev, 2 * sizeof(double2));
    if (e == hipSuccess) e = zero_device(h->d_prev, 2 * sizeof(double2));
    if (e != hipSuccess) {
        h->fini();
        delete h;
        return fail(COMMS_ERR_DEVICE, "fmdemod state alloc: %s", hipGetErrorString(e));
    }
    *out = h;
    return COMMS_OK;
}

comms_status_t comms_fmdemod_f64_run_dev(comms_fmdemod_f64_t* h, const comms_c64* d_in, size_t n, double* d_out, void* stream) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG((d_in && d_out) || !n, "NULL device pointer");
    COMMS_TRY(use_device(h->device));
    if (!n) return COMMS_OK;
    COMMS_ARG(!ranges_overlap(d_in, n * 16, d_out, n * 8), "fmdemod cannot run in place");
    hipStream_t s = nullptr;
    COMMS_TRY(h->enter(stream, &s));
    const size_t want = (n + 255) / 256;
    const unsigned blocks = static_cast<unsigned>(want < 8u * kNumCU ? want : 8u * kNumCU);
    h->tic(s);
    fmdemod_f64_kernel<<<dim3(blocks), dim3(256), 0, s>>>(reinterpret_cast<const double2*>(d_in), h->d_prev + h->cur,
                                                          h->d_prev + (h->cur ^ 1), d_out, n);
    h->toc(s);
    COMMS_TRY(launch_ok("fmdemod_f64_kernel"));
    h->cur ^= 1;
    return COMMS_OK;
}

comms_status_t comms_fmdemod_f64_run(comms_fmdemod_f64_t* h, const comms_c64* in, size_t n, double* out) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG((in && out) || !n, "NULL host pointer");
    COMMS_TRY(use_device(h->device));
    if (!n) return COMMS_OK;
    return h->run_host_units(in, n * sizeof(comms_c64), sizeof(comms_c64), out, n * sizeof(double), sizeof(double),
                             [&](void* d_in, void* d_out, size_t ib, size_t) {
                                 return comms_fmdemod_f64_run_dev(h, static_cast<const comms_c64*>(d_in), ib / sizeof(comms_c64),
                                                                  static_cast<double*>(d_out), COMMS_STREAM_HANDLE);
                             });
}

comms_status_t comms_fmdemod_f64_get_prev(comms_fmdemod_f64_t* h, comms_c64* out_prev) {
    COMMS_ARG(h && out_prev, "NULL argument");
    COMMS_TRY(use_device(h->device));
    COMMS_TRY(h->quiesce());
    COMMS_HIP_TRY(hipMemcpy(out_prev, h->d_prev + h->cur, sizeof(double2), hipMemcpyDeviceToHost));
    return COMMS_OK;
}

comms_status_t comms_fmdemod_f64_set_prev(comms_fmdemod_f64_t* h, const comms_c64* prev) {
    COMMS_ARG(h && prev, "NULL argument");
    COMMS_TRY(use_device(h->device));
    COMMS_TRY(h->quiesce());
    COMMS_HIP_TRY(hipMemcpy(h->d_prev + h->cur, prev, sizeof(double2), hipMemcpyHostToDevice));
    return COMMS_OK;
}

comms_status_t comms_fmdemod_f64_destroy(comms_fmdemod_f64_t* h) {
    if (!h) return COMMS_OK;
    (void)use_device(h->device);
    if (h->d_prev) (void)hipFree(h->d_prev);
    h->fini();
    delete h;
    return COMMS_OK;
}

}  // extern "C"
