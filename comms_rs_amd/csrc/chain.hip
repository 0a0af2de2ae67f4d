// chain.hip -- mixer -> FIR -> decimate-by-R [-> FM demod] as ONE node.
//
// An ADDITIONAL node (the reference has no fused nodes): same results as
// MixerNode -> BatchFirNode -> DecimateNode [-> FMDemodNode] in series
// (examples/fm_radio.rs:146-148 order, mixer of BASELINE config 3 in front).
// This first version chains the four device kernels through two handle-owned
// HBM temporaries on one stream; state (mixer phase, FIR history, FM prev)
// lives in the sub-handles.
#include "common.hpp"

using namespace comms;

struct comms_chain : Handle {
    comms_mixer_t* mixer = nullptr;
    comms_fir_t* fir = nullptr;
    comms_fmdemod_t* fm = nullptr;
    size_t rate = 1;
    bool fm_demod = false;
    Scratch t1, t2, t3;
};

static void free_chain(comms_chain* h) {
    if (h->mixer) comms_mixer_destroy(h->mixer);
    if (h->fir) comms_fir_destroy(h->fir);
    if (h->fm) comms_fmdemod_destroy(h->fm);
    (void)use_device(h->device);
    h->t1.release();
    h->t2.release();
    h->t3.release();
    h->fini();
    delete h;
}

extern "C" {

comms_status_t comms_chain_create(double dphase, double phase, const comms_c32* taps,
                                  size_t n_taps, size_t rate, int32_t fm_demod, int32_t device,
                                  comms_chain_t** out) {
    COMMS_ARG(out != nullptr, "out is NULL");
    *out = nullptr;
    COMMS_ARG(rate >= 1, "rate must be >= 1");
    comms_chain* h = new (std::nothrow) comms_chain;
    COMMS_ARG(h != nullptr, "out of host memory");
    comms_status_t st = h->init(device);
    if (st != COMMS_OK) {
        delete h;
        return st;
    }
    h->rate = rate;
    h->fm_demod = fm_demod != 0;
    st = comms_mixer_create(dphase, phase, device, &h->mixer);
    if (st == COMMS_OK) st = comms_fir_create(taps, n_taps, nullptr, 0, device, &h->fir);
    if (st == COMMS_OK && h->fm_demod) st = comms_fmdemod_create(device, &h->fm);
    if (st != COMMS_OK) {
        free_chain(h);
        return st;
    }
    *out = h;
    return COMMS_OK;
}

comms_status_t comms_chain_run_dev(comms_chain_t* h, const comms_c32* d_in, size_t n,
                                   void* d_out, void* stream) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    COMMS_ARG((d_in && d_out) || !n, "NULL device pointer");
    COMMS_ARG(n % h->rate == 0, "n (%zu) must be a multiple of the decimation rate %zu", n, h->rate);
    COMMS_TRY(use_device(h->device));
    if (!n) return COMMS_OK;
    void* s = stream == COMMS_STREAM_HANDLE ? static_cast<void*>(h->stream) : stream;
    const size_t n_dec = n / h->rate;
    COMMS_TRY(h->t1.reserve(n * sizeof(comms_c32)));
    COMMS_TRY(h->t2.reserve(n * sizeof(comms_c32)));
    comms_c32* a = static_cast<comms_c32*>(h->t1.p);
    comms_c32* b = static_cast<comms_c32*>(h->t2.p);
    COMMS_TRY(comms_mixer_run_dev(h->mixer, d_in, n, a, s));
    COMMS_TRY(comms_fir_run_dev(h->fir, a, n, b, s));
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
    COMMS_TRY(h->in_scratch.reserve(n * sizeof(comms_c32)));
    COMMS_TRY(h->out_scratch.reserve(out_bytes));
    COMMS_HIP_TRY(hipMemcpyAsync(h->in_scratch.p, in, n * sizeof(comms_c32), hipMemcpyHostToDevice, h->stream));
    COMMS_TRY(comms_chain_run_dev(h, static_cast<comms_c32*>(h->in_scratch.p), n, h->out_scratch.p, COMMS_STREAM_HANDLE));
    COMMS_HIP_TRY(hipMemcpyAsync(out, h->out_scratch.p, out_bytes, hipMemcpyDeviceToHost, h->stream));
    COMMS_HIP_TRY(hipStreamSynchronize(h->stream));
    return COMMS_OK;
}

comms_status_t comms_chain_set_timer(comms_chain_t* h, comms_timer_t* t) {
    COMMS_ARG(h != nullptr, "handle is NULL");
    return comms_fir_set_timer(h->fir, t);
}

comms_status_t comms_chain_destroy(comms_chain_t* h) {
    if (!h) return COMMS_OK;
    free_chain(h);
    return COMMS_OK;
}

}  // extern "C"
