// shard.cpp -- host arithmetic of the multi-GPU stream sharding: what a host written
// in C++ or Rust needs to cut a stream over several GPUs with the node handles of this library.  No device
// code, no communication: the hand-over itself is the caller's send / recv of a few hundred raw samples.
//
//   FIR / pulse (N taps)  the N-1 samples before the shard are the reference's `state` (fir_node.rs:193-211)
//                         -> comms_state_from_halo + comms_fir_set_state
//   mixer                 closed-form start phase                    -> comms_shard_mixer_phase
//   chains with FM demod  FM.prev is the decimated filter output before the shard (analog.rs:31): the rank runs
//                         its own chain over comms_chain_prefix_len raw samples first and drops the outputs
//   FFT batches           independent transforms                      -> comms_shard_range over transforms
#include <cmath>

#include "common.hpp"

using namespace comms;

extern "C" {

comms_status_t comms_shard_range(size_t total, uint32_t world, uint32_t rank, size_t* out_start, size_t* out_stop) {
    COMMS_ARG(out_start && out_stop, "NULL argument");
    COMMS_ARG(world >= 1 && rank < world, "rank %u outside world %u", rank, world);
    const size_t base = total / world, rem = total % world;
    const size_t start = rank * base + (rank < rem ? rank : rem);
    *out_start = start;
    *out_stop = start + base + (rank < rem ? 1 : 0);
    return COMMS_OK;
}

// time-ordered halo (oldest first) -> the reference's state layout (newest first, fir.rs:51-52)
comms_status_t comms_state_from_halo(const comms_c32* halo, size_t n, comms_c32* out_state) {
    COMMS_ARG((halo && out_state) || !n, "NULL argument");
    COMMS_ARG(halo != out_state || n < 2, "state_from_halo cannot run in place");
    for (size_t k = 0; k < n; ++k) out_state[k] = halo[n - 1 - k];
    return COMMS_OK;
}

comms_status_t comms_chain_prefix_len(size_t n_taps, size_t rate, int32_t fm_demod, size_t* out_len) {
    COMMS_ARG(out_len != nullptr, "out_len is NULL");
    COMMS_ARG(n_taps >= 1, "a chain has at least one tap");
    const size_t r = rate <= 1 ? 1 : rate;
    const size_t need = (n_taps - 1) + (fm_demod ? r : 0);
    *out_len = (need + r - 1) / r * r;
    return COMMS_OK;
}

// (phase0 + first_index * dphase) mod 2 pi in extended precision: the closed form of the reference's per-sample
// `phase += dphase` with wrap (src/mixer.rs:79-82).  first_index may be negative (a prefix starts before the shard).
comms_status_t comms_shard_mixer_phase(double phase0, double dphase, int64_t first_index, double* out_phase) {
    COMMS_ARG(out_phase != nullptr, "out_phase is NULL");
    COMMS_ARG(std::isfinite(phase0) && std::isfinite(dphase), "phase0 / dphase must be finite");
    // the reference wraps by the f64 constant 2.0 * PI (src/mixer.rs:80-81), so the modulus is fl64(2 pi), not 2 pi
    const long double two_pi = static_cast<long double>(2.0 * 3.14159265358979323846);
    long double ph = fmodl(static_cast<long double>(phase0) + static_cast<long double>(first_index) * static_cast<long double>(dphase), two_pi);
    if (ph < 0) ph += two_pi;
    *out_phase = static_cast<double>(ph);
    return COMMS_OK;
}

}  // extern "C"
