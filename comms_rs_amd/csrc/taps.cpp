// taps.cpp -- host-side filter design (f64, one-time), behind the C ABI.
//
// Replaces rect_taps / gaussian_taps / rc_taps / rrc_taps of the reference
// (src/util/math.rs:48-55, :79-102, :151-196, :221-280).  Design math stays on
// the CPU: it runs once per node and its output
// is the `taps` argument of comms_fir_create / comms_pulse_create.
// Tsym is 1 throughout, as in the reference; outputs are not normalised.
#include <cfloat>
#include <cmath>

#include "common.hpp"

namespace {

constexpr double kPi = 3.14159265358979323846264338327950288;

inline double sinc_pi(double x) { return x != 0.0 ? std::sin(kPi * x) / (kPi * x) : 1.0; }

// sample instant of tap i, in symbols
inline double tap_time(uint32_t i, uint32_t n_taps, double sam_per_sym) {
    return (static_cast<double>(i) - static_cast<double>(n_taps - 1) / 2.0) / sam_per_sym;
}

inline bool near(double a, double b) { return std::fabs(a - b) < DBL_EPSILON; }

inline void put(comms_c32* out, uint32_t i, double re) {
    out[i].re = static_cast<float>(re);
    out[i].im = 0.0f;
}

}  // namespace

extern "C" {

comms_status_t comms_rect_taps(size_t n_taps, comms_c32* out) {
    COMMS_ARG(out || !n_taps, "out is NULL");
    for (size_t i = 0; i < n_taps; ++i) {
        out[i].re = 1.0f;
        out[i].im = 0.0f;
    }
    return COMMS_OK;
}

comms_status_t comms_gaussian_taps(uint32_t n_taps, double sam_per_sym, double alpha,
                                   comms_c32* out) {
    COMMS_ARG(out || !n_taps, "out is NULL");
    const double gain = std::sqrt(alpha / kPi);
    for (uint32_t i = 0; i < n_taps; ++i) {
        const double t = tap_time(i, n_taps, sam_per_sym);
        put(out, i, gain * std::exp(-alpha * (t * t)));
    }
    return COMMS_OK;
}

comms_status_t comms_rc_taps(uint32_t n_taps, double sam_per_sym, double beta, comms_c32* out) {
    COMMS_ARG(out || !n_taps, "out is NULL");
    COMMS_ARG(beta >= 0.0 && beta <= 1.0, "InvalidRolloffError: beta=%g outside [0,1]", beta);
    // the closed form is singular at |t| = 1/(2 beta); the reference substitutes the limit there
    const double t_sing = beta != 0.0 ? 1.0 / (2.0 * beta) : 0.0;
    for (uint32_t i = 0; i < n_taps; ++i) {
        const double t = tap_time(i, n_taps, sam_per_sym);
        double re;
        if (near(t, t_sing) || near(t, -t_sing)) {
            re = (kPi / 4.0) * sinc_pi(1.0 / (2.0 * beta));
        } else {
            const double u = 2.0 * beta * t;
            re = sinc_pi(t) * std::cos(kPi * beta * t) / (1.0 - u * u);
        }
        put(out, i, re);
    }
    return COMMS_OK;
}

comms_status_t comms_rrc_taps(uint32_t n_taps, double sam_per_sym, double beta, comms_c32* out) {
    COMMS_ARG(out || !n_taps, "out is NULL");
    COMMS_ARG(beta >= 0.0 && beta <= 1.0, "InvalidRolloffError: beta=%g outside [0,1]", beta);
    // singular at t = 0 and |t| = 1/(4 beta); limits substituted as the reference does
    const double t_sing = beta != 0.0 ? 1.0 / (4.0 * beta) : 0.0;
    for (uint32_t i = 0; i < n_taps; ++i) {
        const double t = tap_time(i, n_taps, sam_per_sym);
        double re;
        if (std::fabs(t) < DBL_EPSILON) {
            re = 1.0 + beta * (4.0 / kPi - 1.0);
        } else if (near(t, t_sing) || near(t, -t_sing)) {
            const double a = kPi / (4.0 * beta);
            re = (beta / std::sqrt(2.0)) *
                 ((1.0 + 2.0 / kPi) * std::sin(a) + (1.0 - 2.0 / kPi) * std::cos(a));
        } else {
            const double u = 4.0 * beta * t;
            const double num = std::sin(kPi * t * (1.0 - beta)) + u * std::cos(kPi * t * (1.0 + beta));
            re = num / (kPi * t * (1.0 - u * u));
        }
        put(out, i, re);
    }
    return COMMS_OK;
}

}  // extern "C"
