// oracle_estimators.cpp -- CPU restatement of the block estimators.
// TEST INFRASTRUCTURE ONLY (see oracle.cpp).  f64 throughout, as in the reference;
// sums are sequential folds from zero (Iterator::sum).
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace {
struct Cd {
    double re, im;
};
inline Cd mul(Cd a, Cd b) { return Cd{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }

// num-complex Complex::powi(exp) for exp >= 0 == num_traits::pow::pow (exponentiation by
// squaring, exactly this operation order); powi(0) = 1.
Cd powi(Cd base, unsigned exp) {
    if (exp == 0) return Cd{1.0, 0.0};
    while ((exp & 1) == 0) {
        base = mul(base, base);
        exp >>= 1;
    }
    if (exp == 1) return base;
    Cd acc = base;
    while (exp > 1) {
        exp >>= 1;
        base = mul(base, base);
        if (exp & 1) acc = mul(acc, base);
    }
    return acc;
}
}  // namespace

extern "C" {

// src/demodulation/frequency_estimator.rs:27-42: arg( sum_n x[n+1] * conj(x[n]) )
double orc_frequency_offset_estimate(const double* x, size_t n) {
    const Cd* s = reinterpret_cast<const Cd*>(x);
    Cd acc{0.0, 0.0};
    for (size_t i = 0; i + 1 < n; ++i) {
        Cd p = mul(s[i + 1], Cd{s[i].re, -s[i].im});
        acc.re += p.re;
        acc.im += p.im;
    }
    return std::atan2(acc.im, acc.re);
}
// src/demodulation/phase_estimator.rs:26-33: arg( sum x^m ) / m
double orc_psk_phase_estimate(const double* x, size_t n, uint32_t m) {
    const Cd* s = reinterpret_cast<const Cd*>(x);
    Cd acc{0.0, 0.0};
    for (size_t i = 0; i < n; ++i) {
        Cd p = powi(s[i], m);
        acc.re += p.re;
        acc.im += p.im;
    }
    return std::atan2(acc.im, acc.re) / static_cast<double>(m);
}
// src/demodulation/phase_estimator.rs:58-65: arg( sum -1.0 * x^4 ) / 4
double orc_qam_phase_estimate(const double* x, size_t n) {
    const Cd* s = reinterpret_cast<const Cd*>(x);
    Cd acc{0.0, 0.0};
    for (size_t i = 0; i < n; ++i) {
        Cd p = powi(s[i], 4);
        acc.re += -1.0 * p.re;
        acc.im += -1.0 * p.im;
    }
    return std::atan2(acc.im, acc.re) / 4.0;
}

// src/util/math.rs:307-342 qfilt_taps; `out` holds n_taps | 1 values (even counts are incremented, :317-320)
int orc_qfilt_taps(uint32_t n_taps, double alpha, uint32_t sam_per_sym, double* out) {
    const double PI = 3.14159265358979323846264338327950288;
    if (alpha < 0.0 || alpha > 1.0) return 1;  // MathError::InvalidRolloffError
    uint32_t real_n_taps = n_taps;
    if (n_taps % 2 == 0) real_n_taps += 1;
    const int32_t d = static_cast<int32_t>(std::floor(static_cast<double>(real_n_taps) / 2.0));
    for (uint32_t x = 0; x < real_n_taps; ++x) {
        const double tt = static_cast<double>(static_cast<int32_t>(x) - d) / static_cast<double>(sam_per_sym);
        const double two_alpha_tt = 2.0 * alpha * tt;
        if (std::fabs(two_alpha_tt) == 1.0) {
            out[x] = std::sin(PI * alpha * tt) / (8.0 * tt);
        } else {
            const double numerator = alpha * std::cos(PI * alpha * tt);
            const double denominator = PI * (1.0 - (two_alpha_tt * two_alpha_tt));
            out[x] = numerator / denominator;
        }
    }
    return 0;
}

// src/demodulation/timing_estimator.rs:85-112 TimingEstimator::push, literally: mix, two
// batch_fir runs from zero state (src/filter/fir.rs:87-102, rotate + zip-sum), sequential
// sum of the products, -n * arg / (2 pi).
double orc_timing_push(const double* x, size_t len, uint32_t n, uint32_t d, double alpha) {
    const double PI = 3.14159265358979323846264338327950288;
    const Cd* s = reinterpret_cast<const Cd*>(x);
    const size_t nq = static_cast<size_t>(2) * n * d + 1, ndl = static_cast<size_t>(n) * d + 1;
    std::vector<double> qt(nq | 1);
    if (orc_qfilt_taps(static_cast<uint32_t>(nq), alpha, n, qt.data())) return NAN;
    std::vector<Cd> qfilt(nq), delay(ndl, Cd{0.0, 0.0});
    for (size_t k = 0; k < nq; ++k) qfilt[k] = Cd{qt[k], 0.0};
    delay[ndl - 1] = Cd{1.0, 0.0};
    std::vector<Cd> qstate(nq, Cd{0.0, 0.0}), dstate(ndl, Cd{0.0, 0.0});
    Cd sum{0.0, 0.0};
    for (size_t i = 0; i < len; ++i) {
        const double th = -PI * static_cast<double>(i) / static_cast<double>(n);
        const Cd r{1.0 * std::cos(th), 1.0 * std::sin(th)};  // Complex::exp = from_polar(e^0, th)
        const Cd qin = mul(Cd{s[i].re, -s[i].im}, r);
        const Cd din = mul(s[i], r);
        // fir(): rotate_right(1), state[0] = x, zip-sum from zero
        for (size_t k = nq - 1; k > 0; --k) qstate[k] = qstate[k - 1];
        qstate[0] = qin;
        Cd q{0.0, 0.0};
        for (size_t k = 0; k < nq; ++k) {
            const Cd p = mul(qfilt[k], qstate[k]);
            q.re += p.re;
            q.im += p.im;
        }
        for (size_t k = ndl - 1; k > 0; --k) dstate[k] = dstate[k - 1];
        dstate[0] = din;
        Cd dd{0.0, 0.0};
        for (size_t k = 0; k < ndl; ++k) {
            const Cd p = mul(delay[k], dstate[k]);
            dd.re += p.re;
            dd.im += p.im;
        }
        const Cd p = mul(q, dd);
        sum.re += p.re;
        sum.im += p.im;
    }
    return -static_cast<double>(n) * std::atan2(sum.im, sum.re) / (2.0 * PI);
}

// src/demodulation/nco.rs:41-50 Nco::new + :71-77 push, for a block of phase errors.
// `phase_io` carries the node's phase in and out.
void orc_nco_push(double dphase, double* phase_io, const double* perr, size_t n, double* out) {
    const double PI = 3.14159265358979323846264338327950288;
    while (dphase >= 2.0 * PI) dphase -= 2.0 * PI;
    while (dphase < 0.0) dphase += 2.0 * PI;
    double phase = *phase_io;
    for (size_t i = 0; i < n; ++i) {
        phase += dphase + perr[i];
        if (phase > 2.0 * PI) phase -= 2.0 * PI;
        out[2 * i] = 1.0 * std::cos(phase);  // Complex::exp(0 + i phase) = from_polar(e^0, phase)
        out[2 * i + 1] = 1.0 * std::sin(phase);
    }
    *phase_io = phase;
}

}  // extern "C"
