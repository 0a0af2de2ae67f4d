// oracle_estimators.cpp -- CPU restatement of the block estimators.
// TEST INFRASTRUCTURE ONLY (see oracle.cpp).  f64 throughout, as in the reference;
// sums are sequential folds from zero (Iterator::sum).
#include <cmath>
#include <cstddef>
#include <cstdint>

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

}  // extern "C"
