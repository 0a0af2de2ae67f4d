// chain.hip -- mixer / FIR / decimate-by-R [/ FM demod] as ONE node.
//
// An ADDITIONAL node (the reference has no fused nodes): same results as the
// reference nodes in series -- MixerNode -> BatchFirNode -> DecimateNode
// [-> FMDemodNode] (BASELINE config 3; examples/fm_radio.rs:146-148 order with a
// mixer in front) or BatchFirNode -> MixerNode -> DecimateNode (the BASELINE
// metric's chain).  Up to 257 taps it is ONE launch: 8 B read per input sample and
// 8/R (or 4/R) B written, instead of 16 + 16 + 9 + 1.5 B for the four nodes --
//   * fir_decim_kernel (fir_decim.hip): time domain, computes only the kept outputs;
//     used when taps/rate is small enough to beat the FFT (rates 2,3,4,5,6,8,10,12,16);
//   * fir_os1024_kernel<.., MODE> (fir.hip): the 1024-point overlap-save kernel with
//     the extra stages fused in registers, for everything else.
// Longer filters run the four device kernels back to back through HBM temporaries.
#include <cmath>
#include <vector>

#include "common.hpp"
#include "fir_handle.hpp"

using namespace comms;

// four-kernel path with the mixer in front: the FIR node's own history holds MIXED samples, so the
// chain keeps the last N RAW input samples beside it (what comms_chain_{get,set}_fir_state speak)
__global__ __launch_bounds__(256) void chain_raw_hist_kernel(const float2* __restrict__ old_hist,
                                                             const float2* __restrict__ in, size_t n,
                                                             float2* __restrict__ new_hist, int HL) {
    hist_advance(old_hist, in, n, new_hist, HL);
}

struct comms_chain : Handle {
    bool fused = false;
    bool decim = false;  // fused on the time-domain decimating kernel
    bool decim_any = false;  // ... on its any-rate form (fir_decim_any.hip; a mixer in front is folded into the taps)
    bool poly8 = false;      // COMMS_CHAIN_POLYPHASE: always the polyphase frequency-domain kernel (fir_poly8.hip)
    bool fm_separate = false;  // fused mixer / FIR / decimate launch, FM demod as its own (small) kernel behind it
    bool os_dec = false;       // fused: the 4096-point overlap-save kernel with mixer and decimator in its store stage (258 ... 1537 taps)
    bool os_dec16 = false;     // ... the 16384-point kernel (1538 ... 4097 taps; Complex<f32> input: raw formats take the conversion pass)
    bool pre_as_post = false;  // series of launches, mixer in front folded into the taps: runs as the mixer-behind form
    int mode = 0;
    // fused path state
    comms_fir_t* fir = nullptr;
    uint64_t turns = 0, frac = 0;
    float2* d_prev[2] = {nullptr, nullptr};
    int cur = 0;
    // unfused path
    comms_mixer_t* mixer = nullptr;
    comms_fmdemod_t* fm = nullptr;
    size_t rate = 1;
    double dphase = 0.0;  // wrapped, as the mixer steps it
    bool fm_demod = false, mixer_after = false;
    Scratch t1, t2, t3;
    float2* raw_hist[2] = {nullptr, nullptr};  // unfused + mixer first: last n_eff raw inputs, time order
    int raw_cur = 0;
    std::vector<comms_c32> pending_raw;        // a user state not yet mixed into the FIR node's history
    int in_fmt = COMMS_IQ_C32;                 // wire format of d_in (comms_chain_set_input_format)
    float in_scale = 1.0f;
    Scratch t0;                                // converted input, for the paths that read Complex<f32> only
};

// four-kernel path with the mixer in front: the FIR node keeps MIXED samples, so a raw user history is
// mixed with the phases the oscillator had at samples -1, -2, ... (Mixer::mix arithmetic: f64 product
// rounded once, src/mixer.rs:77-78).  Done at the next run, so that set_fir_state and set_phase may
// come in either order.
static comms_status_t chain_flush_raw_state(comms_chain* h) {
    if (h->pending_raw.empty()) return COMMS_OK;
    double ph = 0.0;
    COMMS_TRY(comms_mixer_get_phase(h->mixer, &ph));
    const size_t n_state = h->pending_raw.size();
    std::vector<comms_c32> mixed(n_state);
    for (size_t k = 0; k < n_state; ++k) {
        const double a = ph - static_cast<double>(k + 1) * h->dphase;
        const double c = std::cos(a), s = std::sin(a);
        const double re = h->pending_raw[k].re, im = h->pending_raw[k].im;
        mixed[k].re = static_cast<float>(re * c - im * s);
        mixed[k].im = static_cast<float>(re * s + im * c);
    }
    h->pending_raw.clear();
    return comms_fir_set_state(h->fir, mixed.data(), n_state);
}

static void free_chain(comms_chain* h) {
    if (h->mixer) comms_mixer_destroy(h->mixer);
    if (h->fir) comms_fir_destroy(h->fir);
    if (h->fm) comms_fmdemod_destroy(h->fm);
    (void)use_device(h->device);
    if (h->d_prev[0]) (void)hipFree(h->d_prev[0]);
    if (h->d_prev[1]) (void)hipFree(h->d_prev[1]);
    if (h->raw_hist[0]) (void)hipFree(h->raw_hist[0]);
    if (h->raw_hist[1]) (void)hipFree(h->raw_hist[1]);
    h->t1.release();
    h->t2.release();
    h->t3.release();
    h->t0.release();
    h->fini();
    delete h;
}

extern "C" {

comms_status_t comms_chain_create_ex(double dphase, double phase, const comms_c32* taps, size_t n_taps,
                                     size_t rate, int32_t flags, int32_t device, comms_chain_t** out) {
    COMMS_ARG(out != nullptr, "out is NULL");
    *out = nullptr;
    COMMS_ARG(rate >= 1, "rate must be >= 1");
    COMMS_ARG(std::isfinite(dphase) && std::isfinite(phase), "dphase/phase must be finite");
    comms_chain* h = new (std::nothrow) comms_chain;
    COMMS_ARG(h != nullptr, "out of host memory");
    comms_status_t st = h->init(device);
    if (st != COMMS_OK) {
        delete h;
        return st;
    }
    h->rate = rate;
    h->dphase = mix_wrap_dphase(dphase);
    h->fm_demod = (flags & COMMS_CHAIN_FM_DEMOD) != 0;
    h->mixer_after = (flags & COMMS_CHAIN_MIXER_AFTER_FIR) != 0;
    st = comms_fir_create(taps, n_taps, nullptr, 0, device, &h->fir);
    const bool can_fuse_nofm = !(flags & COMMS_CHAIN_UNFUSED) && n_taps <= 257 && rate <= (1u << 20);
    // FM chains on the overlap-save path: mixer / FIR / decimate as the one fused launch, the demodulator as its
    // own kernel over the n / rate decimated samples.  That beats demodulating inside the overlap-save kernel at
    // every rate (2^24 samples, 127 taps: /2 85.8 against 95.9 us, /3 68 against 83, /8 60 against 82 -- the fused
    // form needs `rate` more halo samples per segment and an atan2 per output in a kernel short of issue slots) and
    // has no limit on taps + rate; COMMS_CHAIN_FM_SEPARATE=0 brings the in-kernel form back for comparison.
    static const int fm_sep = diag_knob("COMMS_CHAIN_FM_SEPARATE", 1);
    const bool can_fuse = can_fuse_nofm && (!h->fm_demod || (!fm_sep && rate <= 64 && n_taps + rate <= 257));
    const bool can_hybrid = h->fm_demod && can_fuse_nofm && !can_fuse;
    // the time-domain kernel against what would run otherwise: an overlap-save fusion, or the four kernels in
    // series, which it beats up to many more MACs per input sample
    const int decim_ok = st == COMMS_OK && rate <= 16
                             ? comms_fir_decim_supported_for(h->fir, static_cast<uint32_t>(rate), h->fm_demod ? 1 : 0,
                                                             can_fuse || can_hybrid ? 1 : 0)
                             : 0;
    // (rate 4 with a long filter: too many MACs for the time-domain kernel, but its chain kind is the one that reaches the polyphase
    // frequency-domain kernel, which takes every call from 64 taps -- fir_poly8.hip)
    const bool poly_pref = decim_ok >= 1 && !(flags & COMMS_CHAIN_TIME_DOMAIN) &&
                           comms_fir_poly8_supported(h->fir, static_cast<uint32_t>(rate), COMMS_CHAIN_DEC | (h->fm_demod ? COMMS_CHAIN_FM : 0),
                                                     static_cast<size_t>(1) << 26) == 2;
    // FM chains where that kernel runs without its demodulator (rates 12 ... 64; rate 4 beyond 249 taps): mixer / FIR / decimate on
    // it and the demodulator as its own small launch over the n / rate kept samples, where the kernel takes EVERY call of this
    // filter (asked with the shortest batch) -- rate 4, 255 taps, 2^24 samples: ~50 us against 70 for the overlap-save launch +
    // demodulator (up to 249 taps the demodulator runs in the kernel: poly_pref above)
    const bool poly_sep = h->fm_demod && st == COMMS_OK && rate != 8 && !poly_pref && !(flags & (COMMS_CHAIN_TIME_DOMAIN | COMMS_CHAIN_FREQ_DOMAIN | COMMS_CHAIN_UNFUSED)) &&
                          comms_fir_poly8_supported(h->fir, static_cast<uint32_t>(rate), COMMS_CHAIN_DEC, rate) == 2;
    const bool can_decim = !(flags & (COMMS_CHAIN_UNFUSED | COMMS_CHAIN_FREQ_DOMAIN)) &&
                           (decim_ok == 2 || poly_pref || (poly_sep && decim_ok >= 1) || (decim_ok == 1 && (flags & COMMS_CHAIN_TIME_DOMAIN)));
    // Rates the per-rate kernel is not built for (17 and up; 11 / 13 / 15 with complex taps): the any-rate kernel (half
    // a wave per output).  Against the overlap-save launch it replaces (58-61 us at 2^24 samples whatever the rate): 255
    // taps 60 us at rate 17, 53 at 20, 39 at 32, 30 at 100, 11 at 1000; 127 taps 49 at 17; 63 taps 45 at 17
    // (profiles/r03_bench_chain_rates.txt) -- so from rate 17; taps up to 512, where the alternative is four kernels in
    // series.  COMMS_CHAIN_TIME_DOMAIN forces it wherever it can run.
    static const int any_min_rate = diag_knob("COMMS_ANY_MIN_RATE", 17);
    const size_t any_from = static_cast<size_t>(any_min_rate);
    const bool can_any = st == COMMS_OK && !can_decim && !(flags & (COMMS_CHAIN_UNFUSED | COMMS_CHAIN_FREQ_DOMAIN)) &&
                         comms_fir_decim_any_supported(h->fir, static_cast<uint32_t>(rate)) &&
                         ((flags & COMMS_CHAIN_TIME_DOMAIN) || rate >= any_from);
    const int32_t poly_mode = (h->mixer_after ? COMMS_CHAIN_POST : COMMS_CHAIN_PRE) | COMMS_CHAIN_DEC | (h->fm_demod ? COMMS_CHAIN_FM : 0);
    const bool force_poly8 = st == COMMS_OK && (flags & COMMS_CHAIN_POLYPHASE) && !(flags & COMMS_CHAIN_UNFUSED) &&
                             comms_fir_poly8_supported(h->fir, static_cast<uint32_t>(rate), poly_mode, static_cast<size_t>(1) << 26) != 0;
    // Filters too long for the other fusions (258 ... 513 taps; until round 5 these chains ran as overlap-save FIR + mixer-decimator
    // [+ demodulator]: 80 us at 2^24 samples and rate 8): the polyphase kernel with five to eight halo rows, where it takes every
    // call of the filter (asked with the shortest batch); the demodulator in the kernel where it has one (rates 8 and 4, 505
    // taps), as its own launch over the kept samples otherwise.
    bool poly_long = false, poly_long_fm = false;
    if (st == COMMS_OK && n_taps > 257 && !force_poly8 &&
        !(flags & (COMMS_CHAIN_UNFUSED | COMMS_CHAIN_TIME_DOMAIN | COMMS_CHAIN_FREQ_DOMAIN))) {
        poly_long_fm = h->fm_demod && comms_fir_poly8_supported(h->fir, static_cast<uint32_t>(rate), poly_mode, rate) == 2;
        poly_long = poly_long_fm || comms_fir_poly8_supported(h->fir, static_cast<uint32_t>(rate), poly_mode & ~COMMS_CHAIN_FM, rate) == 2;
    }
    if (force_poly8 || poly_long) {
        h->fused = true;
        h->poly8 = true;
        h->mode = poly_long && !poly_long_fm ? (poly_mode & ~COMMS_CHAIN_FM) : poly_mode;
        h->fm_separate = poly_long && h->fm_demod && !poly_long_fm;
        if (h->fm_separate) st = comms_fmdemod_create(device, &h->fm);
        h->frac = mix_to_turns(mix_wrap_dphase(dphase));
        h->turns = mix_to_turns(phase);
        for (int i = 0; i < 2 && st == COMMS_OK; ++i) {
            hipError_t e = hipMalloc(&h->d_prev[i], sizeof(float2));
            if (e == hipSuccess) e = zero_device(h->d_prev[i], sizeof(float2));
            if (e != hipSuccess) st = fail(COMMS_ERR_DEVICE, "chain state alloc: %s", hipGetErrorString(e));
        }
        if (st != COMMS_OK) {
            free_chain(h);
            return st;
        }
        *out = h;
        return COMMS_OK;
    }
    if (can_any && !h->mixer_after) {
        // mixer in front: sum_k h[k] x[n-k] e^{i phi(n-k)} = e^{i phi(n)} sum_k (h[k] e^{-i k dphi}) x[n-k] -- the kernel
        // filters the RAW samples with modulated (complex) taps and mixes the kept outputs (the roundings fall
        // elsewhere than in the reference's order, inside the parity tolerance: test_chain_any_rate)
        std::vector<comms_c32> mod(n_taps);
        for (size_t k = 0; k < n_taps; ++k) {
            const double ang = -h->dphase * static_cast<double>(k);
            const double cr = std::cos(ang), ci = std::sin(ang);
            const double tr = taps[k].re, ti = taps[k].im;
            mod[k].re = static_cast<float>(tr * cr - ti * ci);
            mod[k].im = static_cast<float>(tr * ci + ti * cr);
        }
        comms_fir_destroy(h->fir);
        h->fir = nullptr;
        st = comms_fir_create(mod.data(), n_taps, nullptr, 0, device, &h->fir);
    }
    if (st == COMMS_OK && can_any) {
        h->fused = true;
        h->decim_any = true;
        h->fm_separate = poly_sep;
        h->mode = COMMS_CHAIN_POST | COMMS_CHAIN_DEC | (h->fm_demod && !poly_sep ? COMMS_CHAIN_FM : 0);
        if (h->fm_separate) st = comms_fmdemod_create(device, &h->fm);
        h->frac = mix_to_turns(mix_wrap_dphase(dphase));
        h->turns = mix_to_turns(phase);
        for (int i = 0; i < 2 && st == COMMS_OK; ++i) {
            hipError_t e = hipMalloc(&h->d_prev[i], sizeof(float2));
            if (e == hipSuccess) e = zero_device(h->d_prev[i], sizeof(float2));
            if (e != hipSuccess) st = fail(COMMS_ERR_DEVICE, "chain state alloc: %s", hipGetErrorString(e));
        }
    } else if (st == COMMS_OK && (can_fuse || can_decim || can_hybrid)) {
        h->fused = true;
        h->decim = can_decim;
        h->fm_separate = (!can_decim && !can_fuse) || (can_decim && poly_sep);
        h->mode = (h->mixer_after ? COMMS_CHAIN_POST : COMMS_CHAIN_PRE) | COMMS_CHAIN_DEC |
                  (h->fm_demod && !h->fm_separate ? COMMS_CHAIN_FM : 0);
        if (h->fm_separate) st = comms_fmdemod_create(device, &h->fm);
        h->frac = mix_to_turns(mix_wrap_dphase(dphase));
        h->turns = mix_to_turns(phase);
        for (int i = 0; i < 2 && st == COMMS_OK; ++i) {
            hipError_t e = hipMalloc(&h->d_prev[i], sizeof(float2));
            if (e == hipSuccess) e = zero_device(h->d_prev[i], sizeof(float2));
            if (e != hipSuccess) st = fail(COMMS_ERR_DEVICE, "chain state alloc: %s", hipGetErrorString(e));
        }
    } else if (st == COMMS_OK) {
        // 258 ... 1537 taps at the rates the polyphase kernel does not run: the 4096-point overlap-save kernel keeps, mixes and stores
        // every rate-th output itself (one launch instead of FIR + mixer-decimator; 383 taps at rate 5, 2^24 samples: 96 -> ~60 us);
        // FM demod follows as its own launch over the kept samples
        const uint32_t rate32 = static_cast<uint32_t>(rate < (1u << 21) ? rate : 0);
        const bool os_ok = !(flags & (COMMS_CHAIN_UNFUSED | COMMS_CHAIN_TIME_DOMAIN));
        const bool os_dec16 = os_ok && comms_fir_os16k_decim_supported(h->fir, rate32) != 0;  // (1538 ... 4097 taps: the 16384-point kernel)
        const bool os_dec = os_dec16 || (os_ok && comms_fir_os4096_decim_supported(h->fir, rate32) != 0);
        if (!h->mixer_after && !(flags & COMMS_CHAIN_UNFUSED)) {
            // Mixer in front of a long filter: sum_k h[k] x[n-k] e^{i phi_(n-k)} = e^{i phi_n} sum_k (h[k] e^{-i k dphi}) x[n-k],
            // so the chain runs as FIR (modulated taps, raw samples) -> mixer + decimator in one pass over the kept
            // samples, instead of a full-rate mixer pass in front of the FIR (511 taps / 16 at 2^24: 135 -> 95 us).  The
            // roundings fall elsewhere than in the reference's order (inside the parity tolerance); COMMS_CHAIN_UNFUSED
            // keeps the literal four nodes.
            std::vector<comms_c32> mod(n_taps);
            for (size_t k = 0; k < n_taps; ++k) {
                const double ang = -h->dphase * static_cast<double>(k);
                const double cr = std::cos(ang), ci = std::sin(ang);
                const double tr = taps[k].re, ti = taps[k].im;
                mod[k].re = static_cast<float>(tr * cr - ti * ci);
                mod[k].im = static_cast<float>(tr * ci + ti * cr);
            }
            comms_fir_destroy(h->fir);
            h->fir = nullptr;
            st = comms_fir_create(mod.data(), n_taps, nullptr, 0, device, &h->fir);
            h->pre_as_post = st == COMMS_OK;
        }
        if (st == COMMS_OK && os_dec) {
            h->fused = true;
            h->os_dec = !os_dec16;
            h->os_dec16 = os_dec16;
            h->fm_separate = h->fm_demod;
            h->mode = COMMS_CHAIN_POST | COMMS_CHAIN_DEC;
            h->frac = mix_to_turns(mix_wrap_dphase(dphase));
            h->turns = mix_to_turns(phase);
        } else if (st == COMMS_OK) {
            st = comms_mixer_create(dphase, phase, device, &h->mixer);
        }
        if (st == COMMS_OK && h->fm_demod) st = comms_fmdemod_create(device, &h->fm);
        for (int i = 0; i < 2 && st == COMMS_OK && !h->mixer_after && !h->pre_as_post && !h->os_dec && !h->os_dec16; ++i) {
            const size_t bytes = static_cast<size_t>(h->fir->n_eff) * sizeof(float2);
            hipError_t e = hipMalloc(&h->raw_hist[i], bytes);
            if (e == hipSuccess) e = zero_device(h->raw_hist[i], bytes);
            if (e != hipSuccess) st = fail(COMMS_ERR_DEVICE, "chain state alloc: %s", hipGetErrorString(e));
        }
    }
    if (st != COMMS_OK) {
        free_chain(h);
        return st;
    }
    if (h->fir) h->fir->no_poly8 = (flags & COMMS_CHAIN_TIME_DOMAIN) != 0;
    *out = h;
    return COMMS_OK;
}

comms_status_t comms_chain_create(double dphase, double phase, const comms_c32* taps,
                                  size_t n_taps, size_t rate, int32_t fm_demod, int32_t device,
                                  comms_chain_t** out) {
    return comms_chain_create_ex(dphase, phase, taps, n_taps, rate, fm_demod ? COMMS_CHAIN_FM_DEMOD : 0, device, out);
}

comms_status_t comms_chain_is_fused(const comms_chain_t* h, int32_t* out_fused) {
    COMMS_ARG(h && out_fused, "NULL argument");
    *out_fused = h->fused ? (h->poly8 || ((h->decim || h->decim_any) && h->fir && h->fir->last_poly8) ? 4 : h->decim_any ? 3 : h->decim ? 2 : 1) : 0;
    return COMMS_OK;
}

comms_status_t comms_chain_run_dev(comms_chain_t* h, const comms_c32* d_in_any, size_t n,
                                   void* d_out, void* stream) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG((d_in_any && d_out) || !n, "NULL device pointer");
    const comms_c32* d_in = d_in_any;  // n samples in the chain's input format
    COMMS_ARG(n % h->rate == 0, "n (%zu) must be a multiple of the decimation rate %zu", n, h->rate);
    COMMS_TRY(use_device(h->device));
    if (!n) return COMMS_OK;
    const size_t in_elem = h->in_fmt == COMMS_IQ_I16 ? 4 : h->in_fmt == COMMS_IQ_U8 ? 2 : 8;
    COMMS_ARG(!ranges_overlap(d_in, n * in_elem, d_out,
                              (n / h->rate) * (h->fm_demod ? sizeof(float) : sizeof(comms_c32))),
              "the chain cannot run in place");
    COMMS_ARG((reinterpret_cast<uintptr_t>(d_in) & (in_elem - 1)) == 0, "input must be aligned to one IQ sample");
    hipStream_t hs = nullptr;
    COMMS_TRY(h->enter(stream, &hs));  // the stages' state (history, prev) advances in stream order
    void* s = static_cast<void*>(hs);
    if (h->in_fmt != COMMS_IQ_C32 && !(h->fused && (h->decim || h->decim_any || h->poly8 || h->os_dec))) {
        // only the time-domain kernel reads wire formats in its load stage; everything else gets one
        // conversion pass first (same arithmetic, iqformat.hip)
        COMMS_TRY(h->t0.reserve(n * sizeof(comms_c32)));
        comms_c32* c = static_cast<comms_c32*>(h->t0.p);
        if (h->in_fmt == COMMS_IQ_I16)
            COMMS_TRY(comms_iq_i16_to_c32_dev(reinterpret_cast<const int16_t*>(d_in), n, h->in_scale, c, h->device, s));
        else
            COMMS_TRY(comms_iq_u8_to_c32_dev(reinterpret_cast<const uint8_t*>(d_in), n, c, h->device, s));
        d_in = c;
    }
    const size_t n_dec = n / h->rate;
    if (h->fused) {
        void* stage_out = d_out;
        if (h->fm_separate) {  // decimated filter output to scratch, the demodulator reads it
            COMMS_TRY(h->t3.reserve(n_dec * sizeof(comms_c32)));
            stage_out = h->t3.p;
        }
        if (h->os_dec)
            COMMS_TRY(comms_fir_run_os4096_decim_dev(h->fir, d_in, n, stage_out, h->turns, h->frac, static_cast<uint32_t>(h->rate), s));
        else if (h->os_dec16)
            COMMS_TRY(comms_fir_run_os16k_decim_dev(h->fir, d_in, n, stage_out, h->turns, h->frac, static_cast<uint32_t>(h->rate), s));
        else if (h->poly8)
            COMMS_TRY(comms_fir_run_poly8_dev(h->fir, d_in, n, stage_out, h->mode, h->turns, h->frac, static_cast<uint32_t>(h->rate), h->d_prev[h->cur], h->d_prev[h->cur ^ 1], s));
        else if (h->decim_any)
            COMMS_TRY(comms_fir_run_decim_any_dev(h->fir, d_in, n, stage_out, h->mode, h->turns, h->frac, static_cast<uint32_t>(h->rate),
                                                  h->d_prev[h->cur], h->d_prev[h->cur ^ 1], s));
        else if (h->decim)
            COMMS_TRY(comms_fir_run_decim_dev(h->fir, d_in, n, stage_out, h->mode, h->turns, h->frac, static_cast<uint32_t>(h->rate),
                                              h->d_prev[h->cur], h->d_prev[h->cur ^ 1], s));
        else
            COMMS_TRY(comms_fir_run_fused_dev(h->fir, d_in, n, stage_out, h->mode, h->turns, h->frac, static_cast<uint32_t>(h->rate),
                                              h->d_prev[h->cur], h->d_prev[h->cur ^ 1], s));
        h->turns += static_cast<uint64_t>(n) * h->frac;
        if (h->fm_separate)
            return comms_fmdemod_run_dev(h->fm, static_cast<const comms_c32*>(stage_out), n_dec, static_cast<float*>(d_out), s);
        if (h->fm_demod) h->cur ^= 1;
        return COMMS_OK;
    }
    COMMS_TRY(h->t1.reserve(n * sizeof(comms_c32)));
    comms_c32* a = static_cast<comms_c32*>(h->t1.p);
    if (h->mixer_after || h->pre_as_post) {  // FIR, then mixer + decimate in one pass over the kept samples only
        COMMS_TRY(comms_fir_run_dev(h->fir, d_in, n, a, s));
        comms_c32* dst = static_cast<comms_c32*>(d_out);
        if (h->fm_demod) {
            COMMS_TRY(h->t3.reserve(n_dec * sizeof(comms_c32)));
            dst = static_cast<comms_c32*>(h->t3.p);
        }
        COMMS_TRY(comms_mixer_run_decim_dev(h->mixer, a, n, h->rate, dst, s));
        if (!h->fm_demod) return COMMS_OK;
        return comms_fmdemod_run_dev(h->fm, dst, n_dec, static_cast<float*>(d_out), s);
    }
    COMMS_TRY(h->t2.reserve(n * sizeof(comms_c32)));
    comms_c32* b = static_cast<comms_c32*>(h->t2.p);
    {
        COMMS_TRY(chain_flush_raw_state(h));
        COMMS_TRY(comms_mixer_run_dev(h->mixer, d_in, n, a, s));
        COMMS_TRY(comms_fir_run_dev(h->fir, a, n, b, s));
        chain_raw_hist_kernel<<<dim3(1), dim3(256), 0, hs>>>(h->raw_hist[h->raw_cur], reinterpret_cast<const float2*>(d_in), n,
                                                             h->raw_hist[h->raw_cur ^ 1], h->fir->n_eff);
        COMMS_TRY(launch_ok("chain_raw_hist_kernel"));
        h->raw_cur ^= 1;
    }
    if (!h->fm_demod)
        return comms_decimate_run_dev(b, n, sizeof(comms_c32), h->rate, d_out, nullptr, h->device, s);
    COMMS_TRY(h->t3.reserve(n_dec * sizeof(comms_c32)));
    comms_c32* c = static_cast<comms_c32*>(h->t3.p);
    COMMS_TRY(comms_decimate_run_dev(b, n, sizeof(comms_c32), h->rate, c, nullptr, h->device, s));
    return comms_fmdemod_run_dev(h->fm, c, n_dec, static_cast<float*>(d_out), s);
}

comms_status_t comms_chain_run(comms_chain_t* h, const comms_c32* in, size_t n, void* out) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG((in && out) || !n, "NULL host pointer");
    COMMS_ARG(n % h->rate == 0, "n (%zu) must be a multiple of the decimation rate %zu", n, h->rate);
    COMMS_TRY(use_device(h->device));
    if (!n) return COMMS_OK;
    const size_t out_bytes = (n / h->rate) * (h->fm_demod ? sizeof(float) : sizeof(comms_c32));
    const size_t in_elem = h->in_fmt == COMMS_IQ_I16 ? 4 : h->in_fmt == COMMS_IQ_U8 ? 2 : 8;
    // (chunks of whole groups of `rate` samples: DecimateNode restarts its index with every batch, src/util/resample_node.rs:53-65,
    // and a chunk that starts on a multiple of the rate keeps the batch's indexing)
    const size_t out_elem = h->fm_demod ? sizeof(float) : sizeof(comms_c32);
    COMMS_TRY(h->run_host_units(in, n * in_elem, h->rate * in_elem, out, out_bytes, out_elem, [&](void* d_in, void* d_out, size_t ib, size_t) {
        return comms_chain_run_dev(h, static_cast<const comms_c32*>(d_in), ib / in_elem, d_out, COMMS_STREAM_HANDLE);
    }));
    // (a long filter runs the 16384-point FIR kernel inside the series of launches: its failure is this call's)
    return h->fir ? fir_check_sticky(h->fir) : COMMS_OK;
}

// d_in of the run entries then points to raw IQ samples of that format; the conversion (iqformat.hip's
// arithmetic, bit for bit) happens in the load stage of the time-domain chain kernel -- HBM sees 4 or 2
// bytes per input sample -- and as one extra pass in front of the other chain forms.
comms_status_t comms_chain_set_input_format(comms_chain_t* h, int32_t format, float scale) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG(format == COMMS_IQ_C32 || format == COMMS_IQ_I16 || format == COMMS_IQ_U8, "unknown sample format %d", format);
    COMMS_ARG(format != COMMS_IQ_I16 || std::isfinite(scale), "scale must be finite");
    h->in_fmt = format;
    h->in_scale = format == COMMS_IQ_I16 ? scale : 1.0f;
    if (h->fused && (h->decim || h->decim_any || h->poly8 || h->os_dec)) COMMS_TRY(comms_fir_set_input_format(h->fir, format, scale));
    return COMMS_OK;
}

comms_status_t comms_chain_set_timer(comms_chain_t* h, comms_timer_t* t) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    return comms_fir_set_timer(h->fir, t);
}

comms_status_t comms_chain_set_fir_state(comms_chain_t* h, const comms_c32* state, size_t n_state) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG(state || !n_state, "state is NULL");
    COMMS_TRY(use_device(h->device));
    COMMS_TRY(h->quiesce());
    if (h->fused || h->mixer_after || h->pre_as_post) return comms_fir_set_state(h->fir, state, n_state);
    COMMS_ARG(n_state == static_cast<size_t>(h->fir->n_eff), "state must hold exactly the %d effective taps", h->fir->n_eff);
    {
        std::vector<float2> ring(n_state);
        for (size_t k = 0; k < n_state; ++k) ring[n_state - 1 - k] = make_float2(state[k].re, state[k].im);
        COMMS_HIP_TRY(hipMemcpy(h->raw_hist[h->raw_cur], ring.data(), n_state * sizeof(float2), hipMemcpyHostToDevice));
    }
    h->pending_raw.assign(state, state + n_state);  // mixed into the FIR node's history at the next run
    return COMMS_OK;
}

comms_status_t comms_chain_get_fir_state(comms_chain_t* h, comms_c32* state, size_t n_state) {
    COMMS_ARG(h && state, "NULL argument");
    COMMS_TRY(use_device(h->device));
    COMMS_TRY(h->quiesce());
    if (h->fused || h->mixer_after || h->pre_as_post) return comms_fir_get_state(h->fir, state, n_state);
    const size_t N = static_cast<size_t>(h->fir->n_eff);
    COMMS_ARG(n_state <= N, "n_state %zu exceeds the %zu effective taps", n_state, N);
    std::vector<float2> ring(N);
    COMMS_HIP_TRY(hipMemcpy(ring.data(), h->raw_hist[h->raw_cur], N * sizeof(float2), hipMemcpyDeviceToHost));
    for (size_t k = 0; k < n_state; ++k) {
        state[k].re = ring[N - 1 - k].x;
        state[k].im = ring[N - 1 - k].y;
    }
    return COMMS_OK;
}

// Oscillator phase of the next input sample (radians, as comms_mixer_get_phase).
comms_status_t comms_chain_get_phase(comms_chain_t* h, double* out_phase) {
    COMMS_ARG(h && out_phase, "NULL argument");
    if (!h->fused) return comms_mixer_get_phase(h->mixer, out_phase);
    *out_phase = static_cast<double>(h->turns >> 11) * (kMixT * 0x1.0p-53);
    return COMMS_OK;
}

comms_status_t comms_chain_set_phase(comms_chain_t* h, double phase) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG(std::isfinite(phase), "phase must be finite");
    if (!h->fused) return comms_mixer_set_phase(h->mixer, phase);
    h->turns = mix_to_turns(phase);
    return COMMS_OK;
}

// FM.prev of the chain's demodulator (src/modulation/analog.rs:9,31): the last DECIMATED filter
// output of the previous batch.  COMMS_ERR_ARG for a chain without FM demod.
comms_status_t comms_chain_get_fm_prev(comms_chain_t* h, comms_c32* out_prev) {
    COMMS_ARG(h && out_prev, "NULL argument");
    COMMS_ARG(h->fm_demod, "this chain has no FM demodulator");
    if (!h->fused || h->fm_separate) return comms_fmdemod_get_prev(h->fm, out_prev);
    COMMS_TRY(use_device(h->device));
    COMMS_TRY(h->quiesce());
    COMMS_HIP_TRY(hipMemcpy(out_prev, h->d_prev[h->cur], sizeof(float2), hipMemcpyDeviceToHost));
    return COMMS_OK;
}

comms_status_t comms_chain_set_fm_prev(comms_chain_t* h, const comms_c32* prev) {
    COMMS_ARG(h && prev, "NULL argument");
    COMMS_ARG(h->fm_demod, "this chain has no FM demodulator");
    if (!h->fused || h->fm_separate) return comms_fmdemod_set_prev(h->fm, prev);
    COMMS_TRY(use_device(h->device));
    COMMS_TRY(h->quiesce());
    COMMS_HIP_TRY(hipMemcpy(h->d_prev[h->cur], prev, sizeof(float2), hipMemcpyHostToDevice));
    return COMMS_OK;
}

comms_status_t comms_chain_destroy(comms_chain_t* h) {
    if (!h) return COMMS_OK;
    free_chain(h);
    return COMMS_OK;
}

}  // extern "C"
