// oracle.cpp -- CPU restatement of the comms-rs per-sample DSP hot path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under comms_rs_amd/ (the product) may
// include, link or call this file; only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg do, and only as the checker / reported baseline.
//
// Every function restates one reference function, cited as
// `<file>:<lines>` relative to the reference checkout.  The reference is Rust
// and cannot be built in this environment (no rustc/cargo), so parity is
// pinned by the reference's OWN golden vectors (tests/golden/reference_kats.json,
// transcribed from the reference's #[cfg(test)] blocks) -- see
// tests/test_oracle_golden.py.  Unpinned by any reference test (stated in
// DESIGN.md): FM demod, inverse FFT, f32 FIR, cross-call FIR state.
//
// Arithmetic fidelity rules:
//   * build with -ffp-contract=off: rustc/LLVM never contracts a*b+c into FMA;
//   * num-complex multiply is the plain 4-mul/2-add form
//         (a+bi)(c+di) = (a*c - b*d) + (a*d + b*c)i;
//   * `Iterator::sum` over Complex folds from Complex::zero(), k = 0..N-1.
//
// Third-party arithmetic not under the reference tree:
//   rustfft 2.1.0 (Cargo.lock) does ALL FFT arithmetic, in f64
//   (src/fft/mod.rs:86).  Its planner picks Radix4 for powers of two,
//   mixed-radix / Good-Thomas for composites, Rader's / naive DFT for primes
//   and hard-coded butterflies for tiny sizes; every one of them computes the
//   unnormalised DFT  X[k] = sum_j x[j] e^{-/+ 2 pi i jk/N}  to f64 rounding.
//   The restatement below computes the same DFT in f64 (iterative radix-2 for
//   powers of two, exact-index O(N^2) otherwise); after the final f64->f32 cast
//   (src/fft/mod.rs:89-94) the two agree to <= 1 f32 ulp.
//
// Layout: Complex<T> == interleaved {T re, T im} (num-complex is #[repr(C)]).

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

template <typename T>
struct Cx {
    T re, im;
};

// num-complex `impl Mul for Complex<T>`: plain 4-mul / 2-add form.
template <typename T>
inline Cx<T> cmul(Cx<T> a, Cx<T> b) {
    return Cx<T>{static_cast<T>(a.re * b.re - a.im * b.im),
                 static_cast<T>(a.re * b.im + a.im * b.re)};
}
template <typename T>
inline Cx<T> cadd(Cx<T> a, Cx<T> b) {
    return Cx<T>{static_cast<T>(a.re + b.re), static_cast<T>(a.im + b.im)};
}

// i16 arithmetic wraps in a release build (Cargo.toml:33-35 opt-level=3).
inline Cx<int16_t> cmul(Cx<int16_t> a, Cx<int16_t> b) {
    int32_t re = static_cast<int16_t>(a.re * b.re) - static_cast<int16_t>(a.im * b.im);
    int32_t im = static_cast<int16_t>(a.re * b.im) + static_cast<int16_t>(a.im * b.re);
    return Cx<int16_t>{static_cast<int16_t>(re), static_cast<int16_t>(im)};
}
inline Cx<int16_t> cadd(Cx<int16_t> a, Cx<int16_t> b) {
    return Cx<int16_t>{static_cast<int16_t>(a.re + b.re), static_cast<int16_t>(a.im + b.im)};
}

// src/filter/fir.rs:43-54 -- fir(): rotate state right by one, insert the new
// sample at [0], dot(taps, state) over zip(taps, state) (= min of the lengths).
template <typename T>
Cx<T> fir_step(Cx<T> x, const Cx<T>* taps, size_t n_taps, Cx<T>* state, size_t n_state) {
    if (n_state > 1) {  // slice::rotate_right(1)
        Cx<T> last = state[n_state - 1];
        std::memmove(state + 1, state, (n_state - 1) * sizeof(Cx<T>));
        state[0] = last;
    }
    state[0] = x;
    size_t n = n_taps < n_state ? n_taps : n_state;
    Cx<T> acc{0, 0};  // Sum for Complex folds from zero()
    for (size_t k = 0; k < n; ++k) acc = cadd(acc, cmul(taps[k], state[k]));
    return acc;
}

// src/filter/fir.rs:87-102 -- batch_fir(): fir() in a loop over the slice.
template <typename T>
void batch_fir(const Cx<T>* in, size_t n, const Cx<T>* taps, size_t n_taps, Cx<T>* state,
               size_t n_state, Cx<T>* out) {
    for (size_t i = 0; i < n; ++i) out[i] = fir_step(in[i], taps, n_taps, state, n_state);
}

// Same arithmetic, same summation order, no per-sample memmove: a linear
// history buffer replaces rotate_right.  Bit-identical to batch_fir (asserted
// in tests/test_oracle_golden.py); used where the literal form is too slow.
template <typename T>
void batch_fir_norotate(const Cx<T>* in, size_t n, const Cx<T>* taps, size_t n_taps,
                        Cx<T>* state, size_t n_state, Cx<T>* out) {
    size_t m = n_taps < n_state ? n_taps : n_state;
    // lin[j] for j in [0, n_state-1) = history oldest..newest, then the input.
    std::vector<Cx<T>> lin(n_state + n);
    for (size_t j = 0; j < n_state; ++j) lin[j] = state[n_state - 1 - j];
    std::memcpy(lin.data() + n_state, in, n * sizeof(Cx<T>));
    for (size_t i = 0; i < n; ++i) {
        const Cx<T>* newest = lin.data() + n_state + i;
        Cx<T> acc{0, 0};
        for (size_t k = 0; k < m; ++k) acc = cadd(acc, cmul(taps[k], newest[-(ptrdiff_t)k]));
        out[i] = acc;
    }
    // state after the batch: the last n_state elements, newest first.
    for (size_t j = 0; j < n_state; ++j) state[j] = lin[n_state + n - 1 - j];
}

const double kTwoPi = 2.0 * 3.14159265358979323846264338327950288;  // 2.0 * std::f64::consts::PI

// src/mixer.rs:73-84 -- Mixer::mix(): y = cast(cast_f64(x) * exp(i*phase));
// phase += dphase; if phase > 2pi { phase -= 2pi }.
template <typename T>
void mixer_run(const Cx<T>* in, size_t n, double* phase_io, double dphase, Cx<T>* out) {
    double phase = *phase_io;
    for (size_t i = 0; i < n; ++i) {
        Cx<double> inp{static_cast<double>(in[i].re), static_cast<double>(in[i].im)};
        // Complex::exp(Complex::new(0.0, phase)) == from_polar(exp(0.0), phase)
        //   == (1.0 * cos(phase), 1.0 * sin(phase))
        double r = std::exp(0.0);
        Cx<double> rot{r * std::cos(phase), r * std::sin(phase)};
        Cx<double> res = cmul(inp, rot);
        phase += dphase;
        if (phase > kTwoPi) phase -= kTwoPi;
        out[i] = Cx<T>{static_cast<T>(res.re), static_cast<T>(res.im)};
    }
    *phase_io = phase;
}

// f64 unnormalised DFT (see header: stands in for rustfft 2.1.0's f64 FFT).
void dft_f64(std::vector<Cx<double>>& a, bool inverse) {
    const size_t n = a.size();
    if (n <= 1) return;
    const double sgn = inverse ? 1.0 : -1.0;
    if ((n & (n - 1)) == 0) {
        // iterative radix-2 DIT, twiddles straight from sin/cos per index
        for (size_t i = 1, j = 0; i < n; ++i) {
            size_t bit = n >> 1;
            for (; j & bit; bit >>= 1) j ^= bit;
            j ^= bit;
            if (i < j) std::swap(a[i], a[j]);
        }
        // the "plan": twiddles of this length and direction, built once per thread and kept (rustfft plans once,
        // src/fft/fft_node.rs:66-67, and the node reuses the plan for every batch)
        static thread_local std::vector<Cx<double>> w;
        static thread_local size_t w_n = 0;
        static thread_local bool w_inv = false;
        if (w_n != n || w_inv != inverse) {
            w.resize(n / 2);
            for (size_t k = 0; k < n / 2; ++k) {
                double ang = sgn * kTwoPi * static_cast<double>(k) / static_cast<double>(n);
                w[k] = Cx<double>{std::cos(ang), std::sin(ang)};
            }
            w_n = n;
            w_inv = inverse;
        }
        for (size_t len = 2; len <= n; len <<= 1) {
            size_t half = len / 2, step = n / len;
            for (size_t i = 0; i < n; i += len)
                for (size_t k = 0; k < half; ++k) {
                    Cx<double> u = a[i + k];
                    Cx<double> v = cmul(a[i + k + half], w[k * step]);
                    a[i + k] = Cx<double>{u.re + v.re, u.im + v.im};
                    a[i + k + half] = Cx<double>{u.re - v.re, u.im - v.im};
                }
        }
        return;
    }
    // any other length: exact-index O(N^2) DFT with a long-double accumulator
    std::vector<Cx<long double>> w(n);
    for (size_t k = 0; k < n; ++k) {
        long double ang = (long double)sgn * 2.0L * 3.14159265358979323846264338327950288L *
                          (long double)k / (long double)n;
        w[k] = Cx<long double>{cosl(ang), sinl(ang)};
    }
    std::vector<Cx<double>> out(n);
    for (size_t k = 0; k < n; ++k) {
        long double sr = 0, si = 0;
        size_t idx = 0;
        for (size_t j = 0; j < n; ++j) {
            sr += a[j].re * w[idx].re - a[j].im * w[idx].im;
            si += a[j].re * w[idx].im + a[j].im * w[idx].re;
            idx += k;
            if (idx >= n) idx -= n;
        }
        out[k] = Cx<double>{(double)sr, (double)si};
    }
    a.swap(out);
}

// src/fft/mod.rs:73-96 -- BatchFFT::run_fft(): cast T->f64, FFT::process (f64,
// unnormalised; direction fixed by FFTplanner::new(ifft), fft_node.rs:66),
// cast f64->T.
template <typename T>
void fft_run(const Cx<T>* in, size_t n, bool inverse, Cx<T>* out) {
    std::vector<Cx<double>> buf(n);
    for (size_t i = 0; i < n; ++i)
        buf[i] = Cx<double>{static_cast<double>(in[i].re), static_cast<double>(in[i].im)};
    dft_f64(buf, inverse);
    for (size_t i = 0; i < n; ++i)
        out[i] = Cx<T>{static_cast<T>(buf[i].re), static_cast<T>(buf[i].im)};
}

// src/modulation/analog.rs:22-35 -- FM::demod(): theta = samp * prev.conj();
// out = theta.arg() = atan2(im, re); prev = samp.  prev persists.
template <typename T>
void fm_demod(const Cx<T>* in, size_t n, Cx<T>* prev_io, T* out) {
    Cx<T> prev = *prev_io;
    for (size_t i = 0; i < n; ++i) {
        Cx<T> pc{prev.re, static_cast<T>(-prev.im)};  // conj()
        Cx<T> theta = cmul(in[i], pc);
        out[i] = std::atan2(theta.im, theta.re);  // Complex::arg
        prev = in[i];
    }
    *prev_io = prev;
}

// src/util/math.rs:120-126
double sinc(double x) {
    const double PI = 3.14159265358979323846264338327950288;
    if (x != 0.0) return std::sin(PI * x) / (PI * x);
    return 1.0;
}

}  // namespace

extern "C" {

// ---------------------------------------------------------------- FIR
// src/filter/fir.rs:43-54
void orc_fir_f32(const float* x, const float* taps, size_t n_taps, float* state, size_t n_state,
                 float* out) {
    Cx<float> y = fir_step(*reinterpret_cast<const Cx<float>*>(x),
                           reinterpret_cast<const Cx<float>*>(taps), n_taps,
                           reinterpret_cast<Cx<float>*>(state), n_state);
    out[0] = y.re;
    out[1] = y.im;
}
void orc_fir_i16(const int16_t* x, const int16_t* taps, size_t n_taps, int16_t* state,
                 size_t n_state, int16_t* out) {
    Cx<int16_t> y = fir_step(*reinterpret_cast<const Cx<int16_t>*>(x),
                             reinterpret_cast<const Cx<int16_t>*>(taps), n_taps,
                             reinterpret_cast<Cx<int16_t>*>(state), n_state);
    out[0] = y.re;
    out[1] = y.im;
}
// src/filter/fir.rs:87-102 (literal: rotate_right per sample)
void orc_batch_fir_f32(const float* in, size_t n, const float* taps, size_t n_taps, float* state,
                       size_t n_state, float* out) {
    batch_fir(reinterpret_cast<const Cx<float>*>(in), n, reinterpret_cast<const Cx<float>*>(taps),
              n_taps, reinterpret_cast<Cx<float>*>(state), n_state,
              reinterpret_cast<Cx<float>*>(out));
}
void orc_batch_fir_f64(const double* in, size_t n, const double* taps, size_t n_taps,
                       double* state, size_t n_state, double* out) {
    batch_fir(reinterpret_cast<const Cx<double>*>(in), n,
              reinterpret_cast<const Cx<double>*>(taps), n_taps,
              reinterpret_cast<Cx<double>*>(state), n_state, reinterpret_cast<Cx<double>*>(out));
}
void orc_batch_fir_i16(const int16_t* in, size_t n, const int16_t* taps, size_t n_taps,
                       int16_t* state, size_t n_state, int16_t* out) {
    batch_fir(reinterpret_cast<const Cx<int16_t>*>(in), n,
              reinterpret_cast<const Cx<int16_t>*>(taps), n_taps,
              reinterpret_cast<Cx<int16_t>*>(state), n_state,
              reinterpret_cast<Cx<int16_t>*>(out));
}
// same results, no memmove ("circular-index" variant of BASELINE.md section 2)
void orc_batch_fir_norotate_f32(const float* in, size_t n, const float* taps, size_t n_taps,
                                float* state, size_t n_state, float* out) {
    batch_fir_norotate(reinterpret_cast<const Cx<float>*>(in), n,
                       reinterpret_cast<const Cx<float>*>(taps), n_taps,
                       reinterpret_cast<Cx<float>*>(state), n_state,
                       reinterpret_cast<Cx<float>*>(out));
}

// ---------------------------------------------------------------- pulse shaping
// src/pulse.rs:82-92 -- PulseNode::run(): per input symbol [fir(x), fir(0) x (sps-1)].
// `out` holds n_sym * sam_per_sym samples.  sam_per_sym == 0 underflows in the
// reference (`0..&self.sam_per_sym - 1` panics); callers must pass >= 1.
void orc_pulse_f32(const float* sym, size_t n_sym, const float* taps, size_t n_taps,
                   size_t sam_per_sym, float* state, float* out) {
    const Cx<float>* s = reinterpret_cast<const Cx<float>*>(sym);
    const Cx<float>* t = reinterpret_cast<const Cx<float>*>(taps);
    Cx<float>* st = reinterpret_cast<Cx<float>*>(state);
    Cx<float>* o = reinterpret_cast<Cx<float>*>(out);
    for (size_t i = 0; i < n_sym; ++i) {
        *o++ = fir_step(s[i], t, n_taps, st, n_taps);
        for (size_t j = 0; j + 1 < sam_per_sym; ++j)
            *o++ = fir_step(Cx<float>{0, 0}, t, n_taps, st, n_taps);
    }
}
void orc_pulse_f64(const double* sym, size_t n_sym, const double* taps, size_t n_taps,
                   size_t sam_per_sym, double* state, double* out) {
    const Cx<double>* s = reinterpret_cast<const Cx<double>*>(sym);
    const Cx<double>* t = reinterpret_cast<const Cx<double>*>(taps);
    Cx<double>* st = reinterpret_cast<Cx<double>*>(state);
    Cx<double>* o = reinterpret_cast<Cx<double>*>(out);
    for (size_t i = 0; i < n_sym; ++i) {
        *o++ = fir_step(s[i], t, n_taps, st, n_taps);
        for (size_t j = 0; j + 1 < sam_per_sym; ++j)
            *o++ = fir_step(Cx<double>{0, 0}, t, n_taps, st, n_taps);
    }
}
void orc_pulse_i16(const int16_t* sym, size_t n_sym, const int16_t* taps, size_t n_taps,
                   size_t sam_per_sym, int16_t* state, int16_t* out) {
    const Cx<int16_t>* s = reinterpret_cast<const Cx<int16_t>*>(sym);
    const Cx<int16_t>* t = reinterpret_cast<const Cx<int16_t>*>(taps);
    Cx<int16_t>* st = reinterpret_cast<Cx<int16_t>*>(state);
    Cx<int16_t>* o = reinterpret_cast<Cx<int16_t>*>(out);
    for (size_t i = 0; i < n_sym; ++i) {
        *o++ = fir_step(s[i], t, n_taps, st, n_taps);
        for (size_t j = 0; j + 1 < sam_per_sym; ++j)
            *o++ = fir_step(Cx<int16_t>{0, 0}, t, n_taps, st, n_taps);
    }
}

// ---------------------------------------------------------------- mixer
// src/mixer.rs:43-51 -- Mixer::new(): wrap dphase into [0, 2pi); phase untouched.
double orc_mixer_wrap_dphase(double dphase) {
    while (dphase >= kTwoPi) dphase -= kTwoPi;
    while (dphase < 0.0) dphase += kTwoPi;
    return dphase;
}
// src/mixer.rs:73-84 ; *phase is the persistent Mixer.phase; dphase already wrapped.
void orc_mixer_f32(const float* in, size_t n, double* phase, double dphase, float* out) {
    mixer_run(reinterpret_cast<const Cx<float>*>(in), n, phase, dphase,
              reinterpret_cast<Cx<float>*>(out));
}
void orc_mixer_f64(const double* in, size_t n, double* phase, double dphase, double* out) {
    mixer_run(reinterpret_cast<const Cx<double>*>(in), n, phase, dphase,
              reinterpret_cast<Cx<double>*>(out));
}

// ---------------------------------------------------------------- FFT
// src/fft/mod.rs:73-96 ; n == fft_size (a mismatch panics inside rustfft).
void orc_fft_f32(const float* in, size_t n, int inverse, float* out) {
    fft_run(reinterpret_cast<const Cx<float>*>(in), n, inverse != 0,
            reinterpret_cast<Cx<float>*>(out));
}
void orc_fft_f64(const double* in, size_t n, int inverse, double* out) {
    fft_run(reinterpret_cast<const Cx<double>*>(in), n, inverse != 0,
            reinterpret_cast<Cx<double>*>(out));
}

// ---------------------------------------------------------------- resampling
// src/util/resample_node.rs:53-65 -- decimate(): out[j] = in[j*R]; R in {0,1} copies.
// Returns the number of output elements; elem = sizeof(T) (T: Copy, any type).
size_t orc_decimate(const void* in, size_t n, size_t elem, size_t rate, void* out) {
    if (rate == 0 || rate == 1) {
        std::memcpy(out, in, n * elem);
        return n;
    }
    size_t j = 0;
    for (size_t ix = 0; ix < n; ix += rate, ++j)
        std::memcpy(static_cast<char*>(out) + j * elem, static_cast<const char*>(in) + ix * elem,
                    elem);
    return j;
}
// src/util/resample_node.rs:120-131 -- upsample(): out[j*R] = in[j], zeros between.
// T::zero() is all-zero bytes for every numeric T the reference uses.
size_t orc_upsample(const void* in, size_t n, size_t elem, size_t rate, void* out) {
    if (rate == 0 || rate == 1) {
        std::memcpy(out, in, n * elem);
        return n;
    }
    std::memset(out, 0, n * rate * elem);
    for (size_t j = 0; j < n; ++j)
        std::memcpy(static_cast<char*>(out) + j * rate * elem,
                    static_cast<const char*>(in) + j * elem, elem);
    return n * rate;
}

// ---------------------------------------------------------------- FM demod
// src/modulation/analog.rs:22-35 ; prev = {re, im} persists (starts 0+0i, :43-47).
void orc_fm_demod_f32(const float* in, size_t n, float* prev, float* out) {
    fm_demod(reinterpret_cast<const Cx<float>*>(in), n, reinterpret_cast<Cx<float>*>(prev), out);
}
void orc_fm_demod_f64(const double* in, size_t n, double* prev, double* out) {
    fm_demod(reinterpret_cast<const Cx<double>*>(in), n, reinterpret_cast<Cx<double>*>(prev),
             out);
}

// ---------------------------------------------------------------- tap design (f64)
// All return 0 on success, 1 = MathError::InvalidRolloffError (src/util/mod.rs:8-11).
// Outputs are the f64 values BEFORE the final T::from() cast; im is always 0.

// src/util/math.rs:48-55
int orc_rect_taps(size_t n_taps, double* out_re) {
    for (size_t i = 0; i < n_taps; ++i) out_re[i] = 1.0;
    return 0;
}
// src/util/math.rs:79-102
int orc_gaussian_taps(uint32_t n_taps, double sam_per_sym, double alpha, double* out_re) {
    const double PI = 3.14159265358979323846264338327950288;
    const double tsym = 1.0;
    const double fs = sam_per_sym / tsym;
    for (uint32_t i = 0; i < n_taps; ++i) {
        double t = ((double)i - (double)(n_taps - 1) / 2.0) / fs;
        out_re[i] = std::sqrt(alpha / PI) * std::exp(-alpha * (t * t));
    }
    return 0;
}
// src/util/math.rs:120-126
double orc_sinc(double x) { return sinc(x); }
// src/util/math.rs:151-196
int orc_rc_taps(uint32_t n_taps, double sam_per_sym, double beta, double* out_re) {
    const double PI = 3.14159265358979323846264338327950288;
    const double EPS = 2.220446049250313e-16;  // std::f64::EPSILON
    if (beta < 0.0 || beta > 1.0) return 1;
    const double tsym = 1.0;
    const double fs = sam_per_sym / tsym;
    const double zero_denom = (beta != 0.0) ? tsym / (2.0 * beta) : 0.0;
    for (uint32_t i = 0; i < n_taps; ++i) {
        double t = ((double)i - (double)(n_taps - 1) / 2.0) / fs;
        if (std::fabs(t - zero_denom) < EPS || std::fabs(t + zero_denom) < EPS) {
            out_re[i] = (PI / (4.0 * tsym)) * sinc(1.0 / (2.0 * beta));
        } else {
            double d = (2.0 * beta * t) / tsym;
            out_re[i] = (1.0 / tsym) * sinc(t / tsym) * std::cos((PI * beta * t) / tsym) /
                        (1.0 - d * d);
        }
    }
    return 0;
}
// src/util/math.rs:221-280
int orc_rrc_taps(uint32_t n_taps, double sam_per_sym, double beta, double* out_re) {
    const double PI = 3.14159265358979323846264338327950288;
    const double EPS = 2.220446049250313e-16;
    if (beta < 0.0 || beta > 1.0) return 1;
    const double tsym = 1.0;
    const double fs = sam_per_sym / tsym;
    const double zero_denom = (beta != 0.0) ? tsym / (4.0 * beta) : 0.0;
    for (uint32_t i = 0; i < n_taps; ++i) {
        double t = ((double)i - (double)(n_taps - 1) / 2.0) / fs;
        if (std::fabs(t) < EPS) {
            out_re[i] = (1.0 / tsym) * (1.0 + beta * (4.0 / PI - 1.0));
        } else if (std::fabs(t - zero_denom) < EPS || std::fabs(t + zero_denom) < EPS) {
            out_re[i] = (beta / (tsym * std::sqrt(2.0))) *
                        ((1.0 + 2.0 / PI) * std::sin(PI / (4.0 * beta)) +
                         (1.0 - (2.0 / PI)) * std::cos(PI / (4.0 * beta)));
        } else {
            double q = 4.0 * beta * (t / tsym);
            out_re[i] = (1.0 / tsym) *
                        (std::sin(PI * (t / tsym) * (1.0 - beta)) +
                         4.0 * beta * (t / tsym) * std::cos(PI * (t / tsym) * (1.0 + beta))) /
                        (PI * (t / tsym) * (1.0 - q * q));
        }
    }
    return 0;
}

// ---------------------------------------------------------------- raw IQ wire formats
// src/io/raw_iq.rs:16,50-51 (Complex<i16>, re then im) + src/util/math.rs:20-28 cast_complex
void orc_iq_i16_to_f32(const int16_t* in, size_t n, float scale, float* out) {
    for (size_t i = 0; i < 2 * n; ++i) out[i] = static_cast<float>(in[i]) * scale;
}
// examples/single_thread_bpsk.rs:40-44: `(8192.0 * x.re) as i16` -- Rust float->int `as`:
// truncate toward zero, saturate at the type's range, NaN -> 0
void orc_iq_f32_to_i16(const float* in, size_t n, float scale, int16_t* out) {
    for (size_t i = 0; i < 2 * n; ++i) {
        float v = scale * in[i];
        int16_t r;
        if (v != v) r = 0;
        else if (v >= 32767.0f) r = 32767;
        else if (v <= -32768.0f) r = -32768;
        else r = static_cast<int16_t>(static_cast<int32_t>(v));
        out[i] = r;
    }
}
// examples/fm_radio.rs:82-90: (x as f32 - 127.5) / 127.5
void orc_iq_u8_to_f32(const uint8_t* in, size_t n, float* out) {
    for (size_t i = 0; i < 2 * n; ++i) out[i] = (static_cast<float>(in[i]) - 127.5f) / 127.5f;
}

// ---------------------------------------------------------------- PRBS source (config C1 input)
// src/prns.rs:64-71 -- PrnGen<u8>::next_byte(): Fibonacci LFSR, left shift,
// output = MSB before the shift, feedback = parity(state & poly_mask).
void orc_prns_u8(uint8_t poly_mask, uint8_t* state_io, size_t n, uint8_t* out_bits) {
    uint8_t state = *state_io;
    for (size_t i = 0; i < n; ++i) {
        uint8_t fb = (uint8_t)(__builtin_popcount((unsigned)(state & poly_mask)) % 2);
        out_bits[i] = (uint8_t)(state >> 7);
        state = (uint8_t)(state << 1);
        state = (uint8_t)(state | fb);
    }
    *state_io = state;
}

}  // extern "C"
