// nodes.hpp -- the hot-path DSP nodes of comms-rs, backed by libcomms_hip (MI355X).
//
// Same struct names, constructor arguments and public `input` / `output` fields as
// the reference, so a graph written against comms-rs
// reads the same here:
//   FirNode / BatchFirNode::new(taps, state)      src/filter/fir_node.rs:89, :193
//   FFTBatchNode / FFTSampleNode::new(size, ifft) src/fft/fft_node.rs:65, :142
//   MixerNode::new(dphase, phase)                 src/mixer.rs:128
//   PulseNode::new(taps, sam_per_sym)             src/pulse.rs:71
//   DecimateNode / UpsampleNode::new(rate)        src/util/resample_node.rs:23, :87
//   FMDemodNode::new()                            src/modulation/analog_node.rs:43
//   TimingEstimatorNode::new(n, d, alpha)         src/demodulation/timing_estimator.rs:123
//   NcoNode::new(dphase, phase) (block form)      src/demodulation/nco.rs:118
// Messages are host vectors (std::vector<Complex>), moved through the channels by
// value as in the reference; every run() goes H2D -> kernel -> D2H through the C
// ABI.  The *Dev variants at the bottom keep messages device-resident
// (DeviceBuf<T>, clone = refcount bump) -- what the roofline numbers use.
//
// Error convention: comms_status_t 1 -> NodeError::DataError, 2 -> PermanentError.
// Constructors throw std::runtime_error when a handle cannot be created (the
// reference panics in the same places); there is no CPU fallback.
#pragma once

#include <complex>
#include <cstring>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "../../../include/comms_hip.h"
#include "node.hpp"

namespace comms {

using Complex32 = std::complex<float>;
static_assert(sizeof(Complex32) == sizeof(comms_c32), "Complex<f32> must be interleaved {re, im}");

inline const comms_c32* c32(const Complex32* p) { return reinterpret_cast<const comms_c32*>(p); }
inline comms_c32* c32(Complex32* p) { return reinterpret_cast<comms_c32*>(p); }
using Complex64 = std::complex<double>;  // num::Complex<f64>
static_assert(sizeof(Complex64) == sizeof(comms_c64), "Complex<f64> must be interleaved {re, im}");
inline const comms_c64* c64(const Complex64* p) { return reinterpret_cast<const comms_c64*>(p); }
inline comms_c64* c64(Complex64* p) { return reinterpret_cast<comms_c64*>(p); }

inline void throw_on(comms_status_t st, const char* what) {
    if (st != COMMS_OK) throw std::runtime_error(std::string(what) + ": " + comms_last_error());
}
inline NodeError to_node_error(comms_status_t st) {
    return st == COMMS_ERR_ARG ? NodeError::DataError : NodeError::PermanentError;
}

// ---------------------------------------------------------------- device-resident message
template <class T>
class DeviceBuf {
public:
    DeviceBuf() = default;
    explicit DeviceBuf(size_t count, int device = 0) : count_(count) {
        throw_on(comms_buf_alloc(count * sizeof(T), device, &b_), "comms_buf_alloc");
    }
    DeviceBuf(const DeviceBuf& o) : b_(o.b_), count_(o.count_) {  // Clone = retain
        if (b_) comms_buf_retain(b_);
    }
    DeviceBuf(DeviceBuf&& o) noexcept : b_(o.b_), count_(o.count_) { o.b_ = nullptr; }
    DeviceBuf& operator=(DeviceBuf o) noexcept {
        std::swap(b_, o.b_);
        std::swap(count_, o.count_);
        return *this;
    }
    ~DeviceBuf() {
        if (b_) comms_buf_release(b_);
    }
    static DeviceBuf from_host(const std::vector<T>& v, int device = 0) {
        DeviceBuf d(v.size(), device);
        throw_on(comms_buf_upload(d.b_, 0, v.data(), v.size() * sizeof(T)), "comms_buf_upload");
        return d;
    }
    std::vector<T> to_host() const {
        std::vector<T> v(count_);
        throw_on(comms_buf_download(b_, 0, v.data(), count_ * sizeof(T)), "comms_buf_download");
        return v;
    }
    T* ptr() const { return static_cast<T*>(comms_buf_ptr(b_)); }
    comms_buf_t* raw() const { return b_; }  // for comms_buf_{wait_ready,record_use,record_ready}
    size_t size() const { return count_; }
    int device() const { return comms_buf_device(b_); }

private:
    comms_buf_t* b_ = nullptr;
    size_t count_ = 0;
};

// ---------------------------------------------------------------- FIR
class BatchFirNode : public DeriveNode<BatchFirNode> {
public:
    NodeReceiver<std::vector<Complex32>> input;
    NodeSender<std::vector<Complex32>> output;

    BatchFirNode(const std::vector<Complex32>& taps, const std::optional<std::vector<Complex32>>& state = std::nullopt,
                 int device = 0) {
        throw_on(comms_fir_create(c32(taps.data()), taps.size(), state ? c32(state->data()) : nullptr,
                                  state ? state->size() : 0, device, &h_),
                 "BatchFirNode::new");
    }
    BatchFirNode(BatchFirNode&& o) noexcept : input(std::move(o.input)), output(std::move(o.output)), h_(o.h_) { o.h_ = nullptr; }
    ~BatchFirNode() { comms_fir_destroy(h_); }

    Result<std::vector<Complex32>> run(const std::vector<Complex32>& in) {
        std::vector<Complex32> out(in.size());
        comms_status_t st = comms_fir_run(h_, c32(in.data()), in.size(), c32(out.data()));
        if (st != COMMS_OK) return to_node_error(st);
        return out;
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }
    comms_fir_t* handle() const { return h_; }

private:
    comms_fir_t* h_ = nullptr;
};

class FirNode : public DeriveNode<FirNode> {
public:
    NodeReceiver<Complex32> input;
    NodeSender<Complex32> output;

    FirNode(const std::vector<Complex32>& taps, const std::optional<std::vector<Complex32>>& state = std::nullopt,
            int device = 0) {
        throw_on(comms_fir_create(c32(taps.data()), taps.size(), state ? c32(state->data()) : nullptr,
                                  state ? state->size() : 0, device, &h_),
                 "FirNode::new");
    }
    FirNode(FirNode&& o) noexcept : input(std::move(o.input)), output(std::move(o.output)), h_(o.h_) { o.h_ = nullptr; }
    ~FirNode() { comms_fir_destroy(h_); }

    Result<Complex32> run(const Complex32& in) {
        Complex32 out;
        comms_status_t st = comms_fir_run(h_, c32(&in), 1, c32(&out));
        if (st != COMMS_OK) return to_node_error(st);
        return out;
    }
    // the samples already queued behind the first one, in one launch (DeriveNode::call): the
    // filter's state runs through the block exactly as through the same samples one by one
    Result<std::vector<Complex32>> run_block(const std::vector<Complex32>& ins) {
        std::vector<Complex32> out(ins.size());
        comms_status_t st = comms_fir_run(h_, c32(ins.data()), ins.size(), c32(out.data()));
        if (st != COMMS_OK) return to_node_error(st);
        return out;
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }

private:
    comms_fir_t* h_ = nullptr;
};

// ---------------------------------------------------------------- Complex<i16> instantiations
// FirNode<i16> / BatchFirNode<i16> / PulseNode<i16>: the reference's nodes are generic over the sample type and its own
// tests run on Complex<i16> (fir_node.rs:259-313, pulse.rs:129-183); wrapping arithmetic (comms_fir_i16_*).
using Complex16 = std::complex<int16_t>;
static_assert(sizeof(Complex16) == sizeof(comms_c16), "Complex<i16> must be interleaved {re, im}");
inline const comms_c16* c16(const Complex16* p) { return reinterpret_cast<const comms_c16*>(p); }
inline comms_c16* c16(Complex16* p) { return reinterpret_cast<comms_c16*>(p); }

class BatchFirNodeI16 : public DeriveNode<BatchFirNodeI16> {
public:
    NodeReceiver<std::vector<Complex16>> input;
    NodeSender<std::vector<Complex16>> output;
    BatchFirNodeI16(const std::vector<Complex16>& taps, const std::optional<std::vector<Complex16>>& state = std::nullopt,
                    int device = 0) {
        throw_on(comms_fir_i16_create(c16(taps.data()), taps.size(), state ? c16(state->data()) : nullptr,
                                      state ? state->size() : 0, device, &h_),
                 "BatchFirNode<i16>::new");
    }
    BatchFirNodeI16(BatchFirNodeI16&& o) noexcept : input(std::move(o.input)), output(std::move(o.output)), h_(o.h_) { o.h_ = nullptr; }
    ~BatchFirNodeI16() { comms_fir_i16_destroy(h_); }
    Result<std::vector<Complex16>> run(const std::vector<Complex16>& in) {
        std::vector<Complex16> out(in.size());
        comms_status_t st = comms_fir_i16_run(h_, c16(in.data()), in.size(), c16(out.data()));
        if (st != COMMS_OK) return to_node_error(st);
        return out;
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }

private:
    comms_fir_i16_t* h_ = nullptr;
};

class FirNodeI16 : public DeriveNode<FirNodeI16> {
public:
    NodeReceiver<Complex16> input;
    NodeSender<Complex16> output;
    FirNodeI16(const std::vector<Complex16>& taps, const std::optional<std::vector<Complex16>>& state = std::nullopt, int device = 0) {
        throw_on(comms_fir_i16_create(c16(taps.data()), taps.size(), state ? c16(state->data()) : nullptr,
                                      state ? state->size() : 0, device, &h_),
                 "FirNode<i16>::new");
    }
    FirNodeI16(FirNodeI16&& o) noexcept : input(std::move(o.input)), output(std::move(o.output)), h_(o.h_) { o.h_ = nullptr; }
    ~FirNodeI16() { comms_fir_i16_destroy(h_); }
    Result<Complex16> run(const Complex16& in) {
        Complex16 out;
        comms_status_t st = comms_fir_i16_run(h_, c16(&in), 1, c16(&out));
        if (st != COMMS_OK) return to_node_error(st);
        return out;
    }
    Result<std::vector<Complex16>> run_block(const std::vector<Complex16>& ins) {  // queued samples in one launch
        std::vector<Complex16> out(ins.size());
        comms_status_t st = comms_fir_i16_run(h_, c16(ins.data()), ins.size(), c16(out.data()));
        if (st != COMMS_OK) return to_node_error(st);
        return out;
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }

private:
    comms_fir_i16_t* h_ = nullptr;
};

class PulseNodeI16 : public DeriveNode<PulseNodeI16> {
public:
    NodeReceiver<Complex16> input;
    NodeSender<std::vector<Complex16>> output;
    PulseNodeI16(const std::vector<Complex16>& taps, size_t sam_per_sym, int device = 0) : sps_(sam_per_sym) {
        throw_on(comms_pulse_i16_create(c16(taps.data()), taps.size(), sam_per_sym, device, &h_), "PulseNode<i16>::new");
    }
    PulseNodeI16(PulseNodeI16&& o) noexcept : input(std::move(o.input)), output(std::move(o.output)), h_(o.h_), sps_(o.sps_) { o.h_ = nullptr; }
    ~PulseNodeI16() { comms_pulse_i16_destroy(h_); }
    Result<std::vector<Complex16>> run(const Complex16& sym) {
        std::vector<Complex16> out(sps_);
        comms_status_t st = comms_pulse_i16_run(h_, c16(&sym), 1, c16(out.data()));
        if (st != COMMS_OK) return to_node_error(st);
        return out;
    }
    Result<std::vector<std::vector<Complex16>>> run_block(const std::vector<Complex16>& syms) {
        std::vector<Complex16> flat(syms.size() * sps_);
        comms_status_t st = comms_pulse_i16_run(h_, c16(syms.data()), syms.size(), c16(flat.data()));
        if (st != COMMS_OK) return to_node_error(st);
        std::vector<std::vector<Complex16>> out(syms.size());
        for (size_t i = 0; i < syms.size(); ++i) out[i].assign(flat.begin() + i * sps_, flat.begin() + (i + 1) * sps_);
        return out;
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }

private:
    comms_pulse_i16_t* h_ = nullptr;
    size_t sps_;
};

// ---------------------------------------------------------------- Complex<f64> instantiations
// FirNode<f64> / BatchFirNode<f64> / PulseNode<f64>: the reference's own doc example of batch_fir (fir.rs:68-86) and its timing
// estimator (timing_estimator.rs:102-103) run on Complex<f64>; the reference's arithmetic operation for operation, outputs
// bit-identical to it (comms_fir_f64_*, round 5).

class BatchFirNodeF64 : public DeriveNode<BatchFirNodeF64> {
public:
    NodeReceiver<std::vector<Complex64>> input;
    NodeSender<std::vector<Complex64>> output;
    BatchFirNodeF64(const std::vector<Complex64>& taps, const std::optional<std::vector<Complex64>>& state = std::nullopt,
                    int device = 0) {
        throw_on(comms_fir_f64_create(c64(taps.data()), taps.size(), state ? c64(state->data()) : nullptr,
                                      state ? state->size() : 0, device, &h_),
                 "BatchFirNode<f64>::new");
    }
    BatchFirNodeF64(BatchFirNodeF64&& o) noexcept : input(std::move(o.input)), output(std::move(o.output)), h_(o.h_) { o.h_ = nullptr; }
    ~BatchFirNodeF64() { comms_fir_f64_destroy(h_); }
    Result<std::vector<Complex64>> run(const std::vector<Complex64>& in) {
        std::vector<Complex64> out(in.size());
        comms_status_t st = comms_fir_f64_run(h_, c64(in.data()), in.size(), c64(out.data()));
        if (st != COMMS_OK) return to_node_error(st);
        return out;
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }

private:
    comms_fir_f64_t* h_ = nullptr;
};

class FirNodeF64 : public DeriveNode<FirNodeF64> {
public:
    NodeReceiver<Complex64> input;
    NodeSender<Complex64> output;
    FirNodeF64(const std::vector<Complex64>& taps, const std::optional<std::vector<Complex64>>& state = std::nullopt, int device = 0) {
        throw_on(comms_fir_f64_create(c64(taps.data()), taps.size(), state ? c64(state->data()) : nullptr,
                                      state ? state->size() : 0, device, &h_),
                 "FirNode<f64>::new");
    }
    FirNodeF64(FirNodeF64&& o) noexcept : input(std::move(o.input)), output(std::move(o.output)), h_(o.h_) { o.h_ = nullptr; }
    ~FirNodeF64() { comms_fir_f64_destroy(h_); }
    Result<Complex64> run(const Complex64& in) {
        Complex64 out;
        comms_status_t st = comms_fir_f64_run(h_, c64(&in), 1, c64(&out));
        if (st != COMMS_OK) return to_node_error(st);
        return out;
    }
    Result<std::vector<Complex64>> run_block(const std::vector<Complex64>& ins) {  // queued samples in one launch
        std::vector<Complex64> out(ins.size());
        comms_status_t st = comms_fir_f64_run(h_, c64(ins.data()), ins.size(), c64(out.data()));
        if (st != COMMS_OK) return to_node_error(st);
        return out;
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }

private:
    comms_fir_f64_t* h_ = nullptr;
};

class PulseNodeF64 : public DeriveNode<PulseNodeF64> {
public:
    NodeReceiver<Complex64> input;
    NodeSender<std::vector<Complex64>> output;
    PulseNodeF64(const std::vector<Complex64>& taps, size_t sam_per_sym, int device = 0) : sps_(sam_per_sym) {
        throw_on(comms_pulse_f64_create(c64(taps.data()), taps.size(), sam_per_sym, device, &h_), "PulseNode<f64>::new");
    }
    PulseNodeF64(PulseNodeF64&& o) noexcept : input(std::move(o.input)), output(std::move(o.output)), h_(o.h_), sps_(o.sps_) { o.h_ = nullptr; }
    ~PulseNodeF64() { comms_pulse_f64_destroy(h_); }
    Result<std::vector<Complex64>> run(const Complex64& sym) {
        std::vector<Complex64> out(sps_);
        comms_status_t st = comms_pulse_f64_run(h_, c64(&sym), 1, c64(out.data()));
        if (st != COMMS_OK) return to_node_error(st);
        return out;
    }
    Result<std::vector<std::vector<Complex64>>> run_block(const std::vector<Complex64>& syms) {
        std::vector<Complex64> flat(syms.size() * sps_);
        comms_status_t st = comms_pulse_f64_run(h_, c64(syms.data()), syms.size(), c64(flat.data()));
        if (st != COMMS_OK) return to_node_error(st);
        std::vector<std::vector<Complex64>> out(syms.size());
        for (size_t i = 0; i < syms.size(); ++i) out[i].assign(flat.begin() + i * sps_, flat.begin() + (i + 1) * sps_);
        return out;
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }

private:
    comms_pulse_f64_t* h_ = nullptr;
    size_t sps_;
};

// ---------------------------------------------------------------- pulse shaping
class PulseNode : public DeriveNode<PulseNode> {
public:
    NodeReceiver<Complex32> input;
    NodeSender<std::vector<Complex32>> output;

    PulseNode(const std::vector<Complex32>& taps, size_t sam_per_sym, int device = 0) : sps_(sam_per_sym) {
        throw_on(comms_pulse_create(c32(taps.data()), taps.size(), sam_per_sym, device, &h_), "PulseNode::new");
    }
    PulseNode(PulseNode&& o) noexcept : input(std::move(o.input)), output(std::move(o.output)), h_(o.h_), sps_(o.sps_) { o.h_ = nullptr; }
    ~PulseNode() { comms_pulse_destroy(h_); }

    // transmit chain in one launch: the MixerNode::new(dphase, phase) that follows is fused in
    PulseNode& with_mixer(double dphase, std::optional<double> phase = std::nullopt) {
        throw_on(comms_pulse_set_mixer(h_, dphase, phase.value_or(0.0)), "PulseNode::with_mixer");
        return *this;
    }

    // (the i16 store stage, comms_pulse_set_output_format, is offered on the device-resident node below)
    Result<std::vector<Complex32>> run(const Complex32& sym) {
        std::vector<Complex32> out(sps_);
        comms_status_t st = comms_pulse_run(h_, c32(&sym), 1, c32(out.data()));
        if (st != COMMS_OK) return to_node_error(st);
        return out;
    }
    // queued symbols in one launch; still one Vec of sam_per_sym samples per symbol downstream
    Result<std::vector<std::vector<Complex32>>> run_block(const std::vector<Complex32>& syms) {
        std::vector<Complex32> flat(syms.size() * sps_);
        comms_status_t st = comms_pulse_run(h_, c32(syms.data()), syms.size(), c32(flat.data()));
        if (st != COMMS_OK) return to_node_error(st);
        std::vector<std::vector<Complex32>> out(syms.size());
        for (size_t i = 0; i < syms.size(); ++i) out[i].assign(flat.begin() + i * sps_, flat.begin() + (i + 1) * sps_);
        return out;
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }

private:
    comms_pulse_t* h_ = nullptr;
    size_t sps_;
};

// ---------------------------------------------------------------- mixer
class MixerNode : public DeriveNode<MixerNode> {
public:
    NodeReceiver<Complex32> input;
    NodeSender<Complex32> output;

    // NB (dphase, phase): the node's order, not Mixer::new's (mixer.rs:128 vs :43)
    explicit MixerNode(double dphase, std::optional<double> phase = std::nullopt, int device = 0) {
        throw_on(comms_mixer_create(dphase, phase.value_or(0.0), device, &h_), "MixerNode::new");
    }
    MixerNode(MixerNode&& o) noexcept : input(std::move(o.input)), output(std::move(o.output)), h_(o.h_) { o.h_ = nullptr; }
    ~MixerNode() { comms_mixer_destroy(h_); }

    Result<Complex32> run(const Complex32& in) {
        Complex32 out;
        comms_status_t st = comms_mixer_run(h_, c32(&in), 1, c32(&out));
        if (st != COMMS_OK) return to_node_error(st);
        return out;
    }
    Result<std::vector<Complex32>> run_block(const std::vector<Complex32>& ins) {  // see FirNode::run_block
        std::vector<Complex32> out(ins.size());
        comms_status_t st = comms_mixer_run(h_, c32(ins.data()), ins.size(), c32(out.data()));
        if (st != COMMS_OK) return to_node_error(st);
        return out;
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }

private:
    comms_mixer_t* h_ = nullptr;
};

// MixerNode<f64> (src/mixer.rs:93-148 with T = f64, the type of the reference's own mixer tests :160-336)
class MixerNode64 : public DeriveNode<MixerNode64> {
public:
    NodeReceiver<Complex64> input;
    NodeSender<Complex64> output;

    explicit MixerNode64(double dphase, std::optional<double> phase = std::nullopt, int device = 0) {
        throw_on(comms_mixer_create(dphase, phase.value_or(0.0), device, &h_), "MixerNode<f64>::new");
    }
    MixerNode64(MixerNode64&& o) noexcept : input(std::move(o.input)), output(std::move(o.output)), h_(o.h_) { o.h_ = nullptr; }
    ~MixerNode64() { comms_mixer_destroy(h_); }

    Result<Complex64> run(const Complex64& in) {
        Complex64 out;
        comms_status_t st = comms_mixer_run_f64(h_, c64(&in), 1, c64(&out));
        if (st != COMMS_OK) return to_node_error(st);
        return out;
    }
    Result<std::vector<Complex64>> run_block(const std::vector<Complex64>& ins) {  // see FirNode::run_block
        std::vector<Complex64> out(ins.size());
        comms_status_t st = comms_mixer_run_f64(h_, c64(ins.data()), ins.size(), c64(out.data()));
        if (st != COMMS_OK) return to_node_error(st);
        return out;
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }

private:
    comms_mixer_t* h_ = nullptr;
};

// The reference has no batch mixer node; this one mixes a whole Vec per message.
class BatchMixerNode : public DeriveNode<BatchMixerNode> {
public:
    NodeReceiver<std::vector<Complex32>> input;
    NodeSender<std::vector<Complex32>> output;

    explicit BatchMixerNode(double dphase, std::optional<double> phase = std::nullopt, int device = 0) {
        throw_on(comms_mixer_create(dphase, phase.value_or(0.0), device, &h_), "BatchMixerNode::new");
    }
    BatchMixerNode(BatchMixerNode&& o) noexcept : input(std::move(o.input)), output(std::move(o.output)), h_(o.h_) { o.h_ = nullptr; }
    ~BatchMixerNode() { comms_mixer_destroy(h_); }

    Result<std::vector<Complex32>> run(const std::vector<Complex32>& in) {
        std::vector<Complex32> out(in.size());
        comms_status_t st = comms_mixer_run(h_, c32(in.data()), in.size(), c32(out.data()));
        if (st != COMMS_OK) return to_node_error(st);
        return out;
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }

private:
    comms_mixer_t* h_ = nullptr;
};

// ---------------------------------------------------------------- decimate / upsample (T: Copy)
template <class T>
class DecimateNode : public DeriveNode<DecimateNode<T>> {
public:
    NodeReceiver<std::vector<T>> input;
    NodeSender<std::vector<T>> output;
    explicit DecimateNode(size_t dec_rate, int device = 0) : rate_(dec_rate), device_(device) {}

    Result<std::vector<T>> run(const std::vector<T>& signal) {
        std::vector<T> out;
        comms_status_t st = decimate(signal, out);
        if (st != COMMS_OK) return to_node_error(st);
        return out;
    }
    comms_status_t decimate(const std::vector<T>& data, std::vector<T>& out) const {
        size_t n_out = 0;
        comms_decimate_out_len(data.size(), rate_, &n_out);
        out.resize(n_out);
        return comms_decimate_run(data.data(), data.size(), sizeof(T), rate_, out.data(), nullptr, device_);
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }

private:
    size_t rate_;
    int device_;
};

template <class T>
class UpsampleNode : public DeriveNode<UpsampleNode<T>> {
public:
    NodeReceiver<std::vector<T>> input;
    NodeSender<std::vector<T>> output;
    explicit UpsampleNode(size_t ups_rate, int device = 0) : rate_(ups_rate), device_(device) {}

    Result<std::vector<T>> run(const std::vector<T>& signal) {
        std::vector<T> out;
        comms_status_t st = upsample(signal, out);
        if (st != COMMS_OK) return to_node_error(st);
        return out;
    }
    comms_status_t upsample(const std::vector<T>& data, std::vector<T>& out) const {
        size_t n_out = 0;
        comms_status_t st = comms_upsample_out_len(data.size(), rate_, &n_out);
        if (st != COMMS_OK) return st;
        out.resize(n_out);
        return comms_upsample_run(data.data(), data.size(), sizeof(T), rate_, out.data(), nullptr, device_);
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }

private:
    size_t rate_;
    int device_;
};

// ---------------------------------------------------------------- FM demod
class FMDemodNode : public DeriveNode<FMDemodNode> {
public:
    NodeReceiver<std::vector<Complex32>> input;
    NodeSender<std::vector<float>> output;

    explicit FMDemodNode(int device = 0) { throw_on(comms_fmdemod_create(device, &h_), "FMDemodNode::new"); }
    FMDemodNode(FMDemodNode&& o) noexcept : input(std::move(o.input)), output(std::move(o.output)), h_(o.h_) { o.h_ = nullptr; }
    ~FMDemodNode() { comms_fmdemod_destroy(h_); }

    Result<std::vector<float>> run(const std::vector<Complex32>& samples) {
        std::vector<float> out(samples.size());
        comms_status_t st = comms_fmdemod_run(h_, c32(samples.data()), samples.size(), out.data());
        if (st != COMMS_OK) return to_node_error(st);
        return out;
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }

private:
    comms_fmdemod_t* h_ = nullptr;
};

// ---------------------------------------------------------------- FFT
class FFTBatchNode : public DeriveNode<FFTBatchNode> {
public:
    NodeReceiver<std::vector<Complex32>> input;
    NodeSender<std::vector<Complex32>> output;

    FFTBatchNode(size_t fft_size, bool ifft, int device = 0) {
        throw_on(comms_fft_create(fft_size, ifft ? 1 : 0, device, &h_), "FFTBatchNode::new");
    }
    FFTBatchNode(FFTBatchNode&& o) noexcept : input(std::move(o.input)), output(std::move(o.output)), h_(o.h_) { o.h_ = nullptr; }
    ~FFTBatchNode() { comms_fft_destroy(h_); }

    Result<std::vector<Complex32>> run(const std::vector<Complex32>& data) {
        std::vector<Complex32> out(data.size());
        // a wrong length panics inside rustfft in the reference; here it is DataError
        comms_status_t st = comms_fft_run(h_, c32(data.data()), data.size(), c32(out.data()));
        if (st != COMMS_OK) return to_node_error(st);
        return out;
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }

private:
    comms_fft_t* h_ = nullptr;
};

// FFTBatchNode<f64> (the instantiation of the reference's doc examples, fft_node.rs:24) and FMDemodNode<f64>
// (analog_node.rs:20 with T = f64): plain FP64 kernels, correct to f64 rounding (comms_fft_f64_*, comms_fmdemod_f64_*)
class FFTBatchNodeF64 : public DeriveNode<FFTBatchNodeF64> {
public:
    NodeReceiver<std::vector<Complex64>> input;
    NodeSender<std::vector<Complex64>> output;

    FFTBatchNodeF64(size_t fft_size, bool ifft, int device = 0) {
        throw_on(comms_fft_f64_create(fft_size, ifft ? 1 : 0, device, &h_), "FFTBatchNode<f64>::new");
    }
    FFTBatchNodeF64(FFTBatchNodeF64&& o) noexcept : input(std::move(o.input)), output(std::move(o.output)), h_(o.h_) { o.h_ = nullptr; }
    ~FFTBatchNodeF64() { comms_fft_f64_destroy(h_); }

    Result<std::vector<Complex64>> run(const std::vector<Complex64>& data) {
        std::vector<Complex64> out(data.size());
        comms_status_t st = comms_fft_f64_run(h_, c64(data.data()), data.size(), c64(out.data()));
        if (st != COMMS_OK) return to_node_error(st);
        return out;
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }

private:
    comms_fft_f64_t* h_ = nullptr;
};

class FMDemodNodeF64 : public DeriveNode<FMDemodNodeF64> {
public:
    NodeReceiver<std::vector<Complex64>> input;
    NodeSender<std::vector<double>> output;

    explicit FMDemodNodeF64(int device = 0) { throw_on(comms_fmdemod_f64_create(device, &h_), "FMDemodNode<f64>::new"); }
    FMDemodNodeF64(FMDemodNodeF64&& o) noexcept : input(std::move(o.input)), output(std::move(o.output)), h_(o.h_) { o.h_ = nullptr; }
    ~FMDemodNodeF64() { comms_fmdemod_f64_destroy(h_); }

    Result<std::vector<double>> run(const std::vector<Complex64>& samples) {
        std::vector<double> out(samples.size());
        comms_status_t st = comms_fmdemod_f64_run(h_, c64(samples.data()), samples.size(), out.data());
        if (st != COMMS_OK) return to_node_error(st);
        return out;
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }

private:
    comms_fmdemod_f64_t* h_ = nullptr;
};

// #[aggregate]: run returns Some(vec) every fft_size pushes, None otherwise
class FFTSampleNode : public DeriveNode<FFTSampleNode> {
public:
    NodeReceiver<Complex32> input;
    NodeSender<std::vector<Complex32>> output;

    FFTSampleNode(size_t fft_size, bool ifft, int device = 0) : n_(fft_size) {
        throw_on(comms_fft_create(fft_size, ifft ? 1 : 0, device, &h_), "FFTSampleNode::new");
    }
    FFTSampleNode(FFTSampleNode&& o) noexcept
        : input(std::move(o.input)), output(std::move(o.output)), h_(o.h_), n_(o.n_), samples_(std::move(o.samples_)) {
        o.h_ = nullptr;
    }
    ~FFTSampleNode() { comms_fft_destroy(h_); }

    Result<std::optional<std::vector<Complex32>>> run(const Complex32& sample) {
        samples_.push_back(sample);
        if (samples_.size() != n_) return std::optional<std::vector<Complex32>>(std::nullopt);
        std::vector<Complex32> out(n_);
        comms_status_t st = comms_fft_run(h_, c32(samples_.data()), n_, c32(out.data()));
        samples_.clear();
        if (st != COMMS_OK) return to_node_error(st);
        return std::optional<std::vector<Complex32>>(std::move(out));
    }
    // queued samples at once: every completed run of fft_size samples becomes one output message
    // (all of them transformed by one batched launch); the remainder waits in `samples_` as before
    Result<std::vector<std::vector<Complex32>>> run_block(const std::vector<Complex32>& ins) {
        samples_.insert(samples_.end(), ins.begin(), ins.end());
        const size_t k = n_ ? samples_.size() / n_ : 0;
        std::vector<std::vector<Complex32>> outs;
        if (!k) return outs;
        std::vector<Complex32> flat(k * n_);
        comms_status_t st = comms_fft_run(h_, c32(samples_.data()), k * n_, c32(flat.data()));
        samples_.erase(samples_.begin(), samples_.begin() + k * n_);
        if (st != COMMS_OK) return to_node_error(st);
        outs.resize(k);
        for (size_t i = 0; i < k; ++i) outs[i].assign(flat.begin() + i * n_, flat.begin() + (i + 1) * n_);
        return outs;
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }

private:
    comms_fft_t* h_ = nullptr;
    size_t n_;
    std::vector<Complex32> samples_;
};

// ---------------------------------------------------------------- tap design
inline std::vector<Complex32> rrc_taps(uint32_t n_taps, double sam_per_sym, double beta) {
    std::vector<Complex32> t(n_taps);
    throw_on(comms_rrc_taps(n_taps, sam_per_sym, beta, c32(t.data())), "rrc_taps");  // MathError::InvalidRolloffError
    return t;
}
inline std::vector<Complex32> rc_taps(uint32_t n_taps, double sam_per_sym, double beta) {
    std::vector<Complex32> t(n_taps);
    throw_on(comms_rc_taps(n_taps, sam_per_sym, beta, c32(t.data())), "rc_taps");
    return t;
}
inline std::vector<Complex32> gaussian_taps(uint32_t n_taps, double sam_per_sym, double alpha) {
    std::vector<Complex32> t(n_taps);
    throw_on(comms_gaussian_taps(n_taps, sam_per_sym, alpha, c32(t.data())), "gaussian_taps");
    return t;
}
inline std::vector<Complex32> rect_taps(size_t n_taps) {
    std::vector<Complex32> t(n_taps);
    throw_on(comms_rect_taps(n_taps, c32(t.data())), "rect_taps");
    return t;
}

// ---------------------------------------------------------------- demodulation
using Complex64 = std::complex<double>;

// TimingEstimatorNode::new(n, d, alpha) -> Result<Self, MathError>; run(&[Complex<f64>]) -> f64
// (src/demodulation/timing_estimator.rs:116-136).  A bad alpha throws (the reference returns Err).
class TimingEstimatorNode : public DeriveNode<TimingEstimatorNode> {
public:
    NodeReceiver<std::vector<Complex64>> input;
    NodeSender<double> output;

    TimingEstimatorNode(uint32_t n, uint32_t d, double alpha, int device = 0) {
        throw_on(comms_timing_create(n, d, alpha, device, &h_), "TimingEstimatorNode::new");
    }
    TimingEstimatorNode(TimingEstimatorNode&& o) noexcept
        : input(std::move(o.input)), output(std::move(o.output)), h_(o.h_) { o.h_ = nullptr; }
    ~TimingEstimatorNode() { comms_timing_destroy(h_); }

    Result<double> run(const std::vector<Complex64>& samples) {
        double est = 0.0;
        comms_status_t st = comms_timing_push(h_, reinterpret_cast<const double*>(samples.data()), samples.size(), &est);
        if (st != COMMS_OK) return to_node_error(st);
        return est;
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }

private:
    comms_timing_t* h_ = nullptr;
};

// NcoNode::new(dphase, Option<phase>) (src/demodulation/nco.rs:118-133) in block form: one
// message is a vector of phase errors, the output is exp(i*phase) per sample.  (The reference
// node is per sample, f64 -> Complex<f64>; a closed loop runs it at block rate here.)
class BatchNcoNode : public DeriveNode<BatchNcoNode> {
public:
    NodeReceiver<std::vector<double>> input;
    NodeSender<std::vector<Complex64>> output;

    explicit BatchNcoNode(double dphase, double phase = 0.0, int device = 0) {
        throw_on(comms_nco_create(dphase, phase, device, &h_), "NcoNode::new");
    }
    BatchNcoNode(BatchNcoNode&& o) noexcept : input(std::move(o.input)), output(std::move(o.output)), h_(o.h_) { o.h_ = nullptr; }
    ~BatchNcoNode() { comms_nco_destroy(h_); }

    Result<std::vector<Complex64>> run(const std::vector<double>& perr) {
        std::vector<Complex64> out(perr.size());
        comms_status_t st = comms_nco_run(h_, perr.data(), perr.size(), reinterpret_cast<double*>(out.data()));
        if (st != COMMS_OK) return to_node_error(st);
        return out;
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }

private:
    comms_nco_t* h_ = nullptr;
};

inline std::vector<double> qfilt_taps(uint32_t n_taps, double alpha, uint32_t sam_per_sym) {
    std::vector<double> t(comms_qfilt_len(n_taps));
    throw_on(comms_qfilt_taps(n_taps, alpha, sam_per_sym, t.data()), "qfilt_taps");
    return t;
}

// ---------------------------------------------------------------- device-resident nodes
// Messages are DeviceBuf<T>: nothing crosses PCIe between nodes and nothing ever synchronises
// the device.  Every node owns a stream (DevStream); run() is
//     wait_ready(in)  ->  one asynchronous launch on the node's stream  ->  record_use(in),
//     record_ready(out)  ->  send(out)
// so the consumer's stream starts its own launch only after the producer's has finished, while the
// node threads themselves run ahead of the device.  An edge stays on ONE GPU: the buffer's events
// belong to its device and the kernels read it directly, so a message whose device is not the node's is a
// DataError (COMMS_ERR_ARG) -- between GPUs the host copies or sends the samples itself (sharding, above).  Output
// buffers come from the library's cache (no hipMalloc / hipFree in steady state); a buffer's
// memory is recycled only after the launches that read it.  to_host() waits for the producer.
class DevStream {
public:
    explicit DevStream(int device) : device_(device) { throw_on(comms_stream_create(device, &s_), "comms_stream_create"); }
    DevStream(DevStream&& o) noexcept : s_(o.s_), device_(o.device_) { o.s_ = nullptr; }
    DevStream(const DevStream&) = delete;
    ~DevStream() {
        if (!s_) return;
        comms_stream_synchronize(device_, s_);
        comms_stream_destroy(device_, s_);
    }
    // `launch(stream)` is the node's *_run_dev call
    template <class TI, class TO, class F>
    comms_status_t run(const DeviceBuf<TI>& in, DeviceBuf<TO>& out, F&& launch) {
        if (in.device() != device_ || out.device() != device_) return COMMS_ERR_ARG;  // no cross-device edges (see above)
        comms_status_t st = comms_buf_wait_ready(in.raw(), s_);
        if (st == COMMS_OK) st = launch(s_);
        if (st == COMMS_OK) st = comms_buf_record_use(in.raw(), s_);
        if (st == COMMS_OK) st = comms_buf_record_ready(out.raw(), s_);
        return st;
    }
    int device() const { return device_; }

private:
    void* s_ = nullptr;
    int device_;
};

class BatchFirNodeDev : public DeriveNode<BatchFirNodeDev> {
public:
    NodeReceiver<DeviceBuf<Complex32>> input;
    NodeSender<DeviceBuf<Complex32>> output;
    BatchFirNodeDev(const std::vector<Complex32>& taps, const std::optional<std::vector<Complex32>>& state = std::nullopt,
                    int device = 0)
        : device_(device), st_(device) {
        throw_on(comms_fir_create(c32(taps.data()), taps.size(), state ? c32(state->data()) : nullptr,
                                  state ? state->size() : 0, device, &h_),
                 "BatchFirNodeDev::new");
    }
    BatchFirNodeDev(BatchFirNodeDev&& o) noexcept
        : input(std::move(o.input)), output(std::move(o.output)), h_(o.h_), device_(o.device_), st_(std::move(o.st_)) { o.h_ = nullptr; }
    ~BatchFirNodeDev() { comms_fir_destroy(h_); }
    Result<DeviceBuf<Complex32>> run(const DeviceBuf<Complex32>& in) {
        DeviceBuf<Complex32> out(in.size(), device_);
        comms_status_t st = st_.run(in, out, [&](void* s) { return comms_fir_run_dev(h_, c32(in.ptr()), in.size(), c32(out.ptr()), s); });
        if (st != COMMS_OK) return to_node_error(st);
        return out;
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }

private:
    comms_fir_t* h_ = nullptr;
    int device_;
    DevStream st_;
};

class BatchMixerNodeDev : public DeriveNode<BatchMixerNodeDev> {
public:
    NodeReceiver<DeviceBuf<Complex32>> input;
    NodeSender<DeviceBuf<Complex32>> output;
    explicit BatchMixerNodeDev(double dphase, std::optional<double> phase = std::nullopt, int device = 0) : device_(device), st_(device) {
        throw_on(comms_mixer_create(dphase, phase.value_or(0.0), device, &h_), "BatchMixerNodeDev::new");
    }
    BatchMixerNodeDev(BatchMixerNodeDev&& o) noexcept
        : input(std::move(o.input)), output(std::move(o.output)), h_(o.h_), device_(o.device_), st_(std::move(o.st_)) { o.h_ = nullptr; }
    ~BatchMixerNodeDev() { comms_mixer_destroy(h_); }
    Result<DeviceBuf<Complex32>> run(const DeviceBuf<Complex32>& in) {
        DeviceBuf<Complex32> out(in.size(), device_);
        comms_status_t st = st_.run(in, out, [&](void* s) { return comms_mixer_run_dev(h_, c32(in.ptr()), in.size(), c32(out.ptr()), s); });
        if (st != COMMS_OK) return to_node_error(st);
        return out;
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }

private:
    comms_mixer_t* h_ = nullptr;
    int device_;
    DevStream st_;
};

class DecimateNodeDev : public DeriveNode<DecimateNodeDev> {
public:
    NodeReceiver<DeviceBuf<Complex32>> input;
    NodeSender<DeviceBuf<Complex32>> output;
    explicit DecimateNodeDev(size_t dec_rate, int device = 0) : rate_(dec_rate), device_(device), st_(device) {}
    Result<DeviceBuf<Complex32>> run(const DeviceBuf<Complex32>& in) {
        size_t n_out = 0;
        comms_decimate_out_len(in.size(), rate_, &n_out);
        DeviceBuf<Complex32> out(n_out, device_);
        comms_status_t st = st_.run(in, out, [&](void* s) {
            return comms_decimate_run_dev(in.ptr(), in.size(), sizeof(Complex32), rate_, out.ptr(), nullptr, device_, s);
        });
        if (st != COMMS_OK) return to_node_error(st);
        return out;
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }

private:
    size_t rate_;
    int device_;
    DevStream st_;
};

// One macro-free pattern for the remaining device-resident nodes: own the C handle, move-only,
// run() takes the output DeviceBuf from the cache and launches on the node's stream.
class FFTBatchNodeDev : public DeriveNode<FFTBatchNodeDev> {
public:
    NodeReceiver<DeviceBuf<Complex32>> input;
    NodeSender<DeviceBuf<Complex32>> output;
    FFTBatchNodeDev(size_t fft_size, bool ifft, int device = 0) : device_(device), st_(device) {
        throw_on(comms_fft_create(fft_size, ifft ? 1 : 0, device, &h_), "FFTBatchNodeDev::new");
    }
    FFTBatchNodeDev(FFTBatchNodeDev&& o) noexcept
        : input(std::move(o.input)), output(std::move(o.output)), h_(o.h_), device_(o.device_), st_(std::move(o.st_)) { o.h_ = nullptr; }
    ~FFTBatchNodeDev() { comms_fft_destroy(h_); }
    // the message may hold any whole number of transforms (the reference: exactly one)
    Result<DeviceBuf<Complex32>> run(const DeviceBuf<Complex32>& in) {
        DeviceBuf<Complex32> out(in.size(), device_);
        comms_status_t st = st_.run(in, out, [&](void* s) { return comms_fft_run_dev(h_, c32(in.ptr()), in.size(), c32(out.ptr()), s); });
        if (st != COMMS_OK) return to_node_error(st);
        return out;
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }

private:
    comms_fft_t* h_ = nullptr;
    int device_;
    DevStream st_;
};

class FMDemodNodeDev : public DeriveNode<FMDemodNodeDev> {
public:
    NodeReceiver<DeviceBuf<Complex32>> input;
    NodeSender<DeviceBuf<float>> output;
    explicit FMDemodNodeDev(int device = 0) : device_(device), st_(device) {
        throw_on(comms_fmdemod_create(device, &h_), "FMDemodNodeDev::new");
    }
    FMDemodNodeDev(FMDemodNodeDev&& o) noexcept
        : input(std::move(o.input)), output(std::move(o.output)), h_(o.h_), device_(o.device_), st_(std::move(o.st_)) { o.h_ = nullptr; }
    ~FMDemodNodeDev() { comms_fmdemod_destroy(h_); }
    Result<DeviceBuf<float>> run(const DeviceBuf<Complex32>& in) {
        DeviceBuf<float> out(in.size(), device_);
        comms_status_t st = st_.run(in, out, [&](void* s) { return comms_fmdemod_run_dev(h_, c32(in.ptr()), in.size(), out.ptr(), s); });
        if (st != COMMS_OK) return to_node_error(st);
        return out;
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }

private:
    comms_fmdemod_t* h_ = nullptr;
    int device_;
    DevStream st_;
};

// PulseNode over whole symbol blocks: n symbols in, n * sam_per_sym samples out
class BatchPulseNodeDev : public DeriveNode<BatchPulseNodeDev> {
public:
    NodeReceiver<DeviceBuf<Complex32>> input;
    NodeSender<DeviceBuf<Complex32>> output;
    BatchPulseNodeDev(const std::vector<Complex32>& taps, size_t sam_per_sym, int device = 0)
        : sps_(sam_per_sym), device_(device), st_(device) {
        throw_on(comms_pulse_create(c32(taps.data()), taps.size(), sam_per_sym, device, &h_), "BatchPulseNodeDev::new");
    }
    BatchPulseNodeDev(BatchPulseNodeDev&& o) noexcept
        : input(std::move(o.input)), output(std::move(o.output)), h_(o.h_), sps_(o.sps_), device_(o.device_), st_(std::move(o.st_)) { o.h_ = nullptr; }
    ~BatchPulseNodeDev() { comms_pulse_destroy(h_); }
    BatchPulseNodeDev& with_mixer(double dphase, std::optional<double> phase = std::nullopt) {
        throw_on(comms_pulse_set_mixer(h_, dphase, phase.value_or(0.0)), "BatchPulseNodeDev::with_mixer");
        return *this;
    }
    Result<DeviceBuf<Complex32>> run(const DeviceBuf<Complex32>& sym) {
        DeviceBuf<Complex32> out(sym.size() * sps_, device_);
        comms_status_t st = st_.run(sym, out, [&](void* s) { return comms_pulse_run_dev(h_, c32(sym.ptr()), sym.size(), c32(out.ptr()), s); });
        if (st != COMMS_OK) return to_node_error(st);
        return out;
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }

private:
    comms_pulse_t* h_ = nullptr;
    size_t sps_;
    int device_;
    DevStream st_;
};

class UpsampleNodeDev : public DeriveNode<UpsampleNodeDev> {
public:
    NodeReceiver<DeviceBuf<Complex32>> input;
    NodeSender<DeviceBuf<Complex32>> output;
    explicit UpsampleNodeDev(size_t ups_rate, int device = 0) : rate_(ups_rate), device_(device), st_(device) {}
    Result<DeviceBuf<Complex32>> run(const DeviceBuf<Complex32>& in) {
        size_t n_out = 0;
        comms_upsample_out_len(in.size(), rate_, &n_out);
        DeviceBuf<Complex32> out(n_out, device_);
        comms_status_t st = st_.run(in, out, [&](void* s) {
            return comms_upsample_run_dev(in.ptr(), in.size(), sizeof(Complex32), rate_, out.ptr(), nullptr, device_, s);
        });
        if (st != COMMS_OK) return to_node_error(st);
        return out;
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }

private:
    size_t rate_;
    int device_;
    DevStream st_;
};

// ---------------------------------------------------------------- stream shards
// One long stream over several GPUs, one node set per GPU: contiguous shards and one hand-over of the raw samples
// in front of each shard.  Thin wrappers of the C entries (csrc/shard.cpp), so that a C++ host cuts a stream exactly
// as bench.py / sharding.py do.
inline std::pair<size_t, size_t> shard_range(size_t total, unsigned world, unsigned rank) {
    size_t a = 0, b = 0;
    throw_on(comms_shard_range(total, world, rank, &a, &b), "shard_range");
    return {a, b};
}
// the n samples before a shard, time order -> BatchFirNode::new(taps, Some(state)) (fir_node.rs:193-211)
inline std::vector<Complex32> state_from_halo(const std::vector<Complex32>& halo) {
    std::vector<Complex32> st(halo.size());
    throw_on(comms_state_from_halo(c32(halo.data()), halo.size(), c32(st.data())), "state_from_halo");
    return st;
}
// oscillator phase of stream sample first_index: MixerNode::new(dphase, Some(phase)) of the shard's node
inline double shard_mixer_phase(double phase0, double dphase, long long first_index) {
    double ph = 0.0;
    throw_on(comms_shard_mixer_phase(phase0, dphase, first_index, &ph), "shard_mixer_phase");
    return ph;
}
// raw samples a chain shard runs through first (outputs dropped): FIR history + the sample FM.prev comes from
inline size_t chain_prefix_len(size_t n_taps, size_t rate, bool fm_demod) {
    size_t n = 0;
    throw_on(comms_chain_prefix_len(n_taps, rate, fm_demod ? 1 : 0, &n), "chain_prefix_len");
    return n;
}

// mixer / FIR / decimate [/ FM demod] as ONE node (comms_chain_*; an additional node, the results of
// the reference nodes in series).  Out = Complex32 without FM demod, float with it.
template <class Out>
class ChainNodeDev : public DeriveNode<ChainNodeDev<Out>> {
public:
    static constexpr bool kFm = std::is_same<Out, float>::value;
    NodeReceiver<DeviceBuf<Complex32>> input;
    NodeSender<DeviceBuf<Out>> output;
    ChainNodeDev(double dphase, double phase, const std::vector<Complex32>& taps, size_t rate, bool mixer_after_fir = false,
                 int device = 0)
        : rate_(rate), device_(device), st_(device) {
        const int32_t flags = (kFm ? COMMS_CHAIN_FM_DEMOD : 0) | (mixer_after_fir ? COMMS_CHAIN_MIXER_AFTER_FIR : 0);
        throw_on(comms_chain_create_ex(dphase, phase, c32(taps.data()), taps.size(), rate, flags, device, &h_),
                 "ChainNodeDev::new");
    }
    ChainNodeDev(ChainNodeDev&& o) noexcept
        : input(std::move(o.input)), output(std::move(o.output)), h_(o.h_), rate_(o.rate_), device_(o.device_), st_(std::move(o.st_)) { o.h_ = nullptr; }
    ~ChainNodeDev() { comms_chain_destroy(h_); }
    Result<DeviceBuf<Out>> run(const DeviceBuf<Complex32>& in) {
        if (rate_ == 0 || in.size() % rate_) return NodeError::DataError;
        DeviceBuf<Out> out(in.size() / rate_, device_);
        comms_status_t st = st_.run(in, out, [&](void* s) { return comms_chain_run_dev(h_, c32(in.ptr()), in.size(), out.ptr(), s); });
        if (st != COMMS_OK) return to_node_error(st);
        return out;
    }
    int fused_kind() const {  // 0 four kernels, 1 overlap-save fusion, 2 time-domain decimating kernel, 3 its any-rate form, 4 the polyphase
                              // frequency-domain kernel (what the last call ran on: rates 4, 8, 12 ... 64)
        int32_t f = 0;
        comms_chain_is_fused(h_, &f);
        return f;
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }

private:
    comms_chain_t* h_ = nullptr;
    size_t rate_;
    int device_;
    DevStream st_;
};

}  // namespace comms
