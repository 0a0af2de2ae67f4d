// common.hpp -- shared plumbing of libcomms_hip.so (gfx950 only; no CPU fallback).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "../../include/comms_hip.h"

// fused-chain stage bits shared by fir.hip and chain.hip (internal)
#define COMMS_CHAIN_PRE 1  /* mixer before the FIR */
#define COMMS_CHAIN_POST 2 /* mixer after the FIR */
#define COMMS_CHAIN_DEC 4
#define COMMS_CHAIN_FM 8
#ifdef COMMS_DIAG  // the diagnostic build lets scripts/trace_fir.py, stamp_fir.py reach these
#define COMMS_INTERNAL
#else
#define COMMS_INTERNAL __attribute__((visibility("hidden")))
#endif
extern "C" COMMS_INTERNAL int32_t comms_fir_os4096_decim_supported(const comms_fir_t* h, uint32_t rate);
extern "C" COMMS_INTERNAL int32_t comms_fir_os16k_decim_supported(const comms_fir_t* h, uint32_t rate);
extern "C" COMMS_INTERNAL comms_status_t comms_fir_run_os16k_decim_dev(comms_fir_t* h, const comms_c32* d_in, size_t n, void* d_out, uint64_t turns0,
                                                                      uint64_t frac, uint32_t rate, void* stream);
extern "C" COMMS_INTERNAL comms_status_t comms_fir_run_os4096_decim_dev(comms_fir_t* h, const void* d_in, size_t n, void* d_out, uint64_t turns0,
                                                                       uint64_t frac, uint32_t rate, void* stream);
extern "C" COMMS_INTERNAL comms_status_t comms_fir_run_fused_dev(comms_fir_t* h, const comms_c32* d_in, size_t n, void* d_out,
                                                  int32_t mode, uint64_t turns0, uint64_t frac, uint32_t rate,
                                                  const void* fm_prev, void* fm_prev_new, void* stream);

// the same chain on the time-domain decimating kernel (fir_decim.hip), where it applies
extern "C" COMMS_INTERNAL comms_status_t comms_mixer_run_decim_dev(comms_mixer_t* h, const comms_c32* d_in, size_t n, size_t rate,
                                                                 comms_c32* d_out, void* stream);
extern "C" COMMS_INTERNAL int32_t comms_fir_decim_supported(const comms_fir_t* h, uint32_t rate);
extern "C" COMMS_INTERNAL int32_t comms_fir_decim_supported_for(const comms_fir_t* h, uint32_t rate, int32_t fm_demod, int32_t can_fuse);
extern "C" COMMS_INTERNAL comms_status_t comms_fir_run_decim_dev(comms_fir_t* h, const void* d_in, size_t n, void* d_out,
                                                  int32_t mode, uint64_t turns0, uint64_t frac, uint32_t rate,
                                                  const void* fm_prev, void* fm_prev_new, void* stream);

// ... and on the any-rate time-domain kernel (fir_decim_any.hip): mixer behind the FIR only
extern "C" COMMS_INTERNAL int32_t comms_fir_decim_any_supported(const comms_fir_t* h, uint32_t rate);
extern "C" COMMS_INTERNAL comms_status_t comms_fir_run_decim_any_dev(comms_fir_t* h, const void* d_in, size_t n, void* d_out,
                                                  int32_t mode, uint64_t turns0, uint64_t frac, uint32_t rate,
                                                  const void* fm_prev, void* fm_prev_new, void* stream);

// ... and, at rate 8, as eight polyphase branches in the frequency domain (fir_poly8.hip): long filters on long batches
extern "C" COMMS_INTERNAL int32_t comms_fir_poly8_supported(const comms_fir_t* h, uint32_t rate, int32_t mode, size_t n);
extern "C" COMMS_INTERNAL comms_status_t comms_fir_run_poly8_dev(comms_fir_t* h, const void* d_in, size_t n, void* d_out,
                                                  int32_t mode, uint64_t turns0, uint64_t frac, uint32_t rate,
                                                  const void* fm_prev, void* fm_prev_new, void* stream);

namespace comms {

// ---- tuning knobs ------------------------------------------------------------
// Kernel selectors and sweep parameters (COMMS_OS1024_*, COMMS_DECIM_*, COMMS_FFT_*, ...) exist in the DIAGNOSTIC build
// only (`make diag`, -DCOMMS_DIAG: scripts/ load it through scripts/with_lib.py), where they are read once from the
// environment.  In the product build every knob is its default at compile time: no getenv on any launch path, nothing
// in the environment changes which kernel runs or what it computes.  (The two documented runtime limits,
// COMMS_ZERO_COPY_BYTES and COMMS_BUF_POOL_MB in runtime.hip, are resources, not kernel selectors.)
#ifdef COMMS_DIAG
inline int diag_knob(const char* name, int dflt) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}
#else
constexpr int diag_knob(const char*, int dflt) { return dflt; }
#endif

// ---- thread-local last error -------------------------------------------------
inline char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
inline comms_status_t fail(comms_status_t code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define COMMS_HIP_TRY(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess)                                                            \
            return ::comms::fail(COMMS_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr,       \
                                 hipGetErrorString(_e), __FILE__, __LINE__);             \
    } while (0)

#define COMMS_TRY(expr)                       \
    do {                                      \
        comms_status_t _s = (expr);           \
        if (_s != COMMS_OK) return _s;        \
    } while (0)

#define COMMS_ARG(cond, ...)                                             \
    do {                                                                 \
        if (!(cond)) return ::comms::fail(COMMS_ERR_ARG, __VA_ARGS__);   \
    } while (0)

// Every entry point runs on the handle's device: the current device is
// thread-local in HIP and a node is created on one thread and run on another
// (src/node/mod.rs:279-281 in the reference).
comms_status_t use_device(int32_t device);

// Kernel launch check (launch-time errors only; execution stays async).
inline comms_status_t launch_ok(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(COMMS_ERR_DEVICE, "launch of %s failed: %s", what, hipGetErrorString(e));
    return COMMS_OK;
}

// hipFuncSetAttribute applies to the current device only: a `static DeviceOnce` per call site
// remembers which devices have had it (a process may drive several GPUs through the C ABI).
// Node threads of one process reach the same call site concurrently, hence atomics; two threads
// racing on the first use both set the (idempotent) attribute, which is harmless.
struct DeviceOnce {
    std::atomic<bool> done[64] = {};
    bool need() {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
        return !done[dev].exchange(true, std::memory_order_acq_rel);
    }
};

// Grow-only device scratch used by the host-pointer (`*_run`) entry points.
struct Scratch {
    void* p = nullptr;
    size_t cap = 0;
    comms_status_t reserve(size_t bytes) {
        if (bytes <= cap) return COMMS_OK;
        if (p) {
            COMMS_HIP_TRY(hipFree(p));
            p = nullptr;
            cap = 0;
        }
        size_t want = bytes + bytes / 4 + 256;
        COMMS_HIP_TRY(hipMalloc(&p, want));
        cap = want;
        return COMMS_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// Host-mapped pinned staging for SHORT host-pointer calls: the kernel reads its input from, and
// writes its output to, pinned host memory directly, which replaces two DMA submissions by two
// small CPU copies (mixer: 32 -> 18 us per call at 64 samples, 44 -> 31 us at 16384; measured
// crossover with the DMA route between 512 KiB and 2 MiB per direction).
struct Pinned {
    void* h = nullptr;  // host address
    void* d = nullptr;  // the same memory as the device sees it
    size_t cap = 0;
    comms_status_t reserve(size_t bytes) {
        if (bytes <= cap) return COMMS_OK;
        release();
        const size_t want = bytes < 65536 ? 65536 : bytes;
        // coherent (fine-grained) on purpose, not by the HIP_HOST_COHERENT default: the GPU must not keep lines of a
        // staging buffer in its L2 from one call to the next, the CPU rewrites the buffer between them
        COMMS_HIP_TRY(hipHostMalloc(&h, want, hipHostMallocMapped | hipHostMallocCoherent));
        hipError_t e = hipHostGetDevicePointer(&d, h, 0);
        if (e != hipSuccess) {
            (void)hipHostFree(h);
            h = d = nullptr;
            return fail(COMMS_ERR_DEVICE, "hipHostGetDevicePointer failed: %s", hipGetErrorString(e));
        }
        cap = want;
        return COMMS_OK;
    }
    void release() {
        if (h) (void)hipHostFree(h);
        h = d = nullptr;
        cap = 0;
    }
};
// Create-time zeroing of device state (FIR history, FM.prev).  hipMemset on device memory returns before the fill
// has run, on the legacy stream -- which the handles' own non-blocking streams (and PyTorch's side streams) do not
// wait for: a first launch could read the allocation's old bytes.  So the fill is waited for here, once per create.
inline hipError_t zero_device(void* p, size_t bytes) {
    hipError_t e = hipMemsetAsync(p, 0, bytes, nullptr);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    return e;
}

// calls moving at most this many bytes each way take the zero-copy route (COMMS_ZERO_COPY_BYTES)
size_t zero_copy_limit();

}  // namespace comms

// Kernel timer: a pool of hipEvent pairs recorded around a node's dominant kernel, device slots that the kernels
// themselves stamp (below), or both (bench / profiling).
struct comms_timer {
    int32_t device = 0;
    size_t n = 0;        // event pairs (0: a stamps-only timer)
    size_t next = 0;     // bracketed launches so far (the pairs wrap modulo n)
    size_t stride = 1;   // events bracket every stride-th launch of the attached node
    size_t seq = 0;      // launches seen since the last reset
    hipEvent_t* start = nullptr;
    hipEvent_t* stop = nullptr;
    size_t ns = 0;       // launches with stamp slots (stamping stops there until the next reset)
    size_t snext = 0;
    unsigned long long* d_begin = nullptr;  // [ns][kStampSlots], initialised to ~0
    unsigned long long* d_end = nullptr;    // [ns][kStampSlots], initialised to 0
};

// In-kernel begin / end stamps of a launch: s_memrealtime (the 100 MHz counter every CU reads alike) taken by the
// first thread of every workgroup when it starts and by every wave once its last store has been acknowledged,
// reduced into kStampSlots slots per launch by atomic min / max (workgroups b and b + kStampSlots share a slot; the
// host reduces the slots).  Unlike an event pair this idles nothing: the launch is an ordinary one, back to back with
// its neighbours in the stream, so the figure is the kernel's duration IN the stream (first wave in to last wave
// out), measured on every launch.  A null KStamp costs a scalar compare at either end of the kernel.
constexpr int kStampSlots = 256;
struct KStamp {
    unsigned long long* begin;
    unsigned long long* end;
};
__device__ __forceinline__ void kstamp_begin(const KStamp& k) {
    if (k.begin != nullptr && threadIdx.x == 0)
        atomicMin(k.begin + (blockIdx.x & (kStampSlots - 1)), static_cast<unsigned long long>(__builtin_amdgcn_s_memrealtime()));
}
__device__ __forceinline__ void kstamp_end(const KStamp& k) {
    if (k.end != nullptr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if ((threadIdx.x & 63) == 0)
            atomicMax(k.end + (blockIdx.x & (kStampSlots - 1)), static_cast<unsigned long long>(__builtin_amdgcn_s_memrealtime()));
    }
}

namespace comms {

// Base of every node handle: device + own stream + scratch for host-pointer runs.
// Streams of handles and host-graph nodes come from a per-device pool and go back to it instead of being destroyed
// (runtime.hip): a HIP event keeps referring to the stream it was last recorded on, and the events that travel with
// pooled buffers outlive the node -- and so the stream -- that recorded them.  On this runtime, synchronising such an
// event after hipStreamDestroy intermittently failed with "operation not permitted when stream is capturing".
COMMS_INTERNAL comms_status_t stream_acquire(int32_t device, hipStream_t* out);
COMMS_INTERNAL void stream_release(int32_t device, hipStream_t s);
// Node handles alive per device (the per-thread handles of the handle-less entry points excepted: they only ever
// follow their own stream).  A handle remembers the last stream it launched on, possibly a pooled one that its owner
// has since released: comms_stream_pool_trim must not destroy streams while such a handle exists.
COMMS_INTERNAL void handle_count(int32_t device, int delta);
// Handles that currently FOLLOW a stream of the device's pool other than their own (a node thread's stream, handed to
// run_dev): comms_stream_pool_trim refuses while there is one -- its owner may have released that stream to the pool,
// and the handle's next drain would synchronise a destroyed stream.  pool_owns: did the pool create `s`?
COMMS_INTERNAL void follower_count(int32_t device, int delta);
COMMS_INTERNAL bool pool_owns(int32_t device, hipStream_t s);
// in + out bytes from which a host-pointer call is pipelined in chunks (COMMS_HOST_PIPE_BYTES, a documented runtime limit
// like COMMS_ZERO_COPY_BYTES; 0 = never)
COMMS_INTERNAL size_t host_pipe_bytes();
COMMS_INTERNAL size_t host_chunk_bytes();  // COMMS_HOST_CHUNK_BYTES: bytes of the larger side per chunk (default 8 MiB)

struct Handle {
    int32_t device = 0;
    hipStream_t stream = nullptr;
    Scratch in_scratch, out_scratch;
    Pinned pin_in, pin_out;
    comms_timer* timer = nullptr;

    // Event bracketing of the dominant kernel launch (no-ops without an attached timer that holds event pairs).
    // Every timed launch site runs exactly one of: tic ... toc, or take_events; each counts the launch once and
    // brackets it only when it is the timer's stride-th.
    bool has_events() const { return timer && timer->n; }
    bool timed() const { return has_events() && timer->seq % timer->stride == 0; }
    void tic(hipStream_t s) {
        if (timed()) (void)hipEventRecord(timer->start[timer->next % timer->n], s);
    }
    void toc(hipStream_t s) {
        if (!has_events()) return;
        if (timed()) {
            (void)hipEventRecord(timer->stop[timer->next % timer->n], s);
            ++timer->next;
        }
        ++timer->seq;
    }
    // For launches made with hipExtLaunchKernelGGL: the pair is updated with the kernel's own begin / end
    // timestamps (no dispatch gap), which is what rocprofv3 reports as its duration.  False: launch plainly.
    bool take_events(hipEvent_t& a, hipEvent_t& b) {
        if (!has_events()) return false;
        const bool on = timed();
        ++timer->seq;
        if (on) {
            a = timer->start[timer->next % timer->n];
            b = timer->stop[timer->next % timer->n];
            ++timer->next;
        }
        return on;
    }
    // In-kernel stamps: the slots of the launch about to be made, or a null KStamp when the attached timer has none
    // (left).  Only kernels that take a KStamp are timed this way.
    KStamp next_stamp() {
        if (!timer || !timer->d_begin || timer->snext >= timer->ns) return KStamp{nullptr, nullptr};
        const size_t i = timer->snext++;
        return KStamp{timer->d_begin + i * kStampSlots, timer->d_end + i * kStampSlots};
    }

    bool counted = false;
    comms_status_t init(int32_t dev, bool count_me = true) {
        COMMS_TRY(use_device(dev));
        device = dev;
        COMMS_TRY(stream_acquire(dev, &stream));
        counted = count_me;
        if (counted) handle_count(dev, +1);
        return COMMS_OK;
    }
    // `stream` arguments of the C ABI are passed through as HIP does: NULL is the
    // legacy default stream; COMMS_STREAM_HANDLE selects the handle's own stream.
    hipStream_t pick(void* s) const {
        return s == COMMS_STREAM_HANDLE ? stream : reinterpret_cast<hipStream_t>(s);
    }
    // Device-resident node state (FIR history ping-pong, FM prev, FFT work buffers) is advanced by
    // the launches themselves, in stream order.  A handle therefore follows ONE stream at a time:
    // `enter` is the stream pick of every stateful run_dev -- when the caller moves the node to a
    // different stream, the work still pending on the previous one is drained first (host wait,
    // rare) so that launches on the two streams can never race on the state; `quiesce` is what
    // the state getters / setters call before touching the state from the host.
    hipStream_t last_stream = nullptr;
    bool launched = false;
    bool follows_pooled = false;  // last_stream is a pooled stream that is not this handle's own
    void set_following(hipStream_t s) {
        const bool f = s != nullptr && s != stream && pool_owns(device, s);  // (looked up on a CHANGE of stream only)
        if (f != follows_pooled) follower_count(device, f ? +1 : -1);
        follows_pooled = f;
    }
    comms_status_t enter(void* s_arg, hipStream_t* out) {
        hipStream_t s = pick(s_arg);
        if (launched && s != last_stream) COMMS_TRY(drain());
        if (s != last_stream) set_following(s);
        last_stream = s;
        launched = true;
        *out = s;
        return COMMS_OK;
    }
    comms_status_t quiesce() {
        if (launched) COMMS_TRY(drain());
        return COMMS_OK;
    }
    // Host wait for the launches pending on the stream the handle followed last.  Whatever the outcome the handle
    // forgets that stream: a caller-owned stream that was destroyed meanwhile (against the header's lifetime rule)
    // fails this ONE call and the handle goes on with the next stream it is given.
    comms_status_t drain() {
        hipStream_t s = last_stream;
        launched = false;
        last_stream = nullptr;
        hipError_t e = hipStreamSynchronize(s);
        set_following(nullptr);
        if (e != hipSuccess)
            return fail(COMMS_ERR_DEVICE, "draining the handle's previous stream: %s", hipGetErrorString(e));
        return COMMS_OK;
    }
    // The host-pointer form of a node: `launch(d_in, d_out)` runs the device form on this handle's
    // stream.  Short calls work on host-mapped pinned staging (two CPU copies instead of two DMA
    // submissions), longer ones go through device scratch.  Synchronous.
    template <class F>
    comms_status_t run_host(const void* in, size_t in_bytes, void* out, size_t out_bytes, F&& launch) {
        if (in_bytes <= zero_copy_limit() && out_bytes <= zero_copy_limit()) {
            COMMS_TRY(pin_in.reserve(in_bytes));
            COMMS_TRY(pin_out.reserve(out_bytes));
            std::memcpy(pin_in.h, in, in_bytes);
            COMMS_TRY(launch(pin_in.d, pin_out.d));
            COMMS_HIP_TRY(hipStreamSynchronize(stream));
            std::memcpy(out, pin_out.h, out_bytes);
            return COMMS_OK;
        }
        COMMS_TRY(in_scratch.reserve(in_bytes));
        COMMS_TRY(out_scratch.reserve(out_bytes));
        COMMS_HIP_TRY(hipMemcpyAsync(in_scratch.p, in, in_bytes, hipMemcpyHostToDevice, stream));
        COMMS_TRY(launch(in_scratch.p, out_scratch.p));
        COMMS_HIP_TRY(hipMemcpyAsync(out, out_scratch.p, out_bytes, hipMemcpyDeviceToHost, stream));
        COMMS_HIP_TRY(hipStreamSynchronize(stream));
        return COMMS_OK;
    }

    // ---- long host-pointer calls, pipelined (round 5) ----------------------------------------------------------------
    // The drop-in path of a comms-rs graph is `run(&[Complex<T>]) -> Vec<Complex<T>>` (src/filter/fir_node.rs:215-220):
    // host memory in, host memory out.  Until round 4 that was one copy in, the launch, one copy out, one direction of
    // the PCIe link at a time.  Above kHostPipeBytes (and where both directions carry bytes) the batch is now cut into chunks of whole UNITS (a unit = in_u input
    // bytes that yield out_u output bytes: a sample, `rate` samples of a decimator, one transform): the calling thread
    // feeds chunk after chunk -- copy in, launch(es), an event -- and a helper thread copies each chunk's outputs back
    // on a second stream as soon as its event has fired, so that the two directions of the link run together.  Stateful
    // nodes stream across the chunks exactly as they stream across calls (same launches, same order, one stream), so the
    // samples are bit-identical to the single-shot path (tests/test_gpu_host_pipeline.py).
    hipStream_t out_stream = nullptr;
    std::vector<hipEvent_t> pipe_events;
    static constexpr size_t kHostPipeBytes = 64u << 20;   // in + out bytes from which a call is pipelined
    static constexpr size_t kHostChunkBytes = 16u << 20;  // of the larger side, per chunk
    template <class F>
    comms_status_t run_host_units(const void* in, size_t in_bytes, size_t in_u, void* out, size_t out_bytes, size_t out_u, F&& launch) {
        const size_t big_u = in_u > out_u ? in_u : out_u;
        size_t per = big_u ? host_chunk_bytes() / big_u : 0;
        if (per < 1) per = 1;
        const size_t n_units = in_u ? (in_bytes + in_u - 1) / in_u : 1;
        const size_t n_chunks = (n_units + per - 1) / per;
        // Only where BOTH directions carry a real share of the bytes: what the pipeline buys is the two directions of the
        // link at once (2^24 samples through the FIR node: 4.84 -> 3.26 ms); a decimating chain's output is an eighth of
        // its input, and its one big copy in is faster whole than in chunks (2.74 against 2.82 - 3.20 ms) --
        // scripts/bench_host_path.py, profiles/r05_host_pipeline.txt
        const bool both_ways = out_bytes * 4 >= in_bytes && in_bytes * 4 >= out_bytes;
        if (in_bytes + out_bytes < host_pipe_bytes() || n_chunks < 2 || !in_u || !out_u || !both_ways)
            return run_host(in, in_bytes, out, out_bytes, [&](void* d_in, void* d_out) { return launch(d_in, d_out, in_bytes, out_bytes); });
        COMMS_TRY(in_scratch.reserve(in_bytes));
        COMMS_TRY(out_scratch.reserve(out_bytes));
        if (!out_stream) COMMS_TRY(stream_acquire(device, &out_stream));
        while (pipe_events.size() < n_chunks) {
            hipEvent_t e = nullptr;
            COMMS_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            pipe_events.push_back(e);
        }
        struct Shared {
            std::mutex m;
            std::condition_variable cv;
            size_t recorded = 0;   // chunks whose event has been recorded
            bool abort = false;
            hipError_t err = hipSuccess;
        } sh;
        const char* cin = static_cast<const char*>(in);
        char* cout_ = static_cast<char*>(out);
        char* din = static_cast<char*>(in_scratch.p);
        char* dout = static_cast<char*>(out_scratch.p);
        auto out_range = [&](size_t k, size_t& o0, size_t& o1) {
            o0 = k * per * out_u;
            o1 = (k + 1) * per * out_u;
            if (o0 > out_bytes) o0 = out_bytes;
            if (o1 > out_bytes || k + 1 == n_chunks) o1 = out_bytes;
        };
        std::thread back([&] {
            if (hipSetDevice(device) != hipSuccess) {
                std::lock_guard<std::mutex> lk(sh.m);
                sh.err = hipErrorInvalidDevice;
                return;
            }
            for (size_t k = 0; k < n_chunks; ++k) {
                {
                    std::unique_lock<std::mutex> lk(sh.m);
                    sh.cv.wait(lk, [&] { return sh.recorded > k || sh.abort; });
                    if (sh.abort) return;
                }
                size_t o0, o1;
                out_range(k, o0, o1);
                hipError_t e = hipStreamWaitEvent(out_stream, pipe_events[k], 0);
                if (e == hipSuccess && o1 > o0) e = hipMemcpyAsync(cout_ + o0, dout + o0, o1 - o0, hipMemcpyDeviceToHost, out_stream);
                if (e != hipSuccess) {
                    std::lock_guard<std::mutex> lk(sh.m);
                    sh.err = e;
                    return;
                }
            }
            hipError_t e = hipStreamSynchronize(out_stream);
            if (e != hipSuccess) {
                std::lock_guard<std::mutex> lk(sh.m);
                sh.err = e;
            }
        });
        comms_status_t st = COMMS_OK;
        for (size_t k = 0; k < n_chunks && st == COMMS_OK; ++k) {
            size_t i0 = k * per * in_u, i1 = (k + 1) * per * in_u;
            if (i1 > in_bytes || k + 1 == n_chunks) i1 = in_bytes;
            size_t o0, o1;
            out_range(k, o0, o1);
            hipError_t e = hipMemcpyAsync(din + i0, cin + i0, i1 - i0, hipMemcpyHostToDevice, stream);
            if (e != hipSuccess) st = fail(COMMS_ERR_DEVICE, "host pipeline: copy in failed: %s", hipGetErrorString(e));
            if (st == COMMS_OK) st = launch(din + i0, dout + o0, i1 - i0, o1 - o0);
            if (st == COMMS_OK && (e = hipEventRecord(pipe_events[k], stream)) != hipSuccess)
                st = fail(COMMS_ERR_DEVICE, "host pipeline: event record failed: %s", hipGetErrorString(e));
            std::lock_guard<std::mutex> lk(sh.m);
            if (st == COMMS_OK) sh.recorded = k + 1;
            else sh.abort = true;
            sh.cv.notify_all();
        }
        back.join();
        if (st != COMMS_OK) {
            (void)hipStreamSynchronize(stream);  // nothing of this call stays in flight behind an error
            return st;
        }
        if (sh.err != hipSuccess) return fail(COMMS_ERR_DEVICE, "host pipeline: copy out failed: %s", hipGetErrorString(sh.err));
        COMMS_HIP_TRY(hipStreamSynchronize(stream));
        return COMMS_OK;
    }

    void fini() {
        set_following(nullptr);
        for (hipEvent_t e : pipe_events) (void)hipEventDestroy(e);
        pipe_events.clear();
        if (out_stream) stream_release(device, out_stream);
        out_stream = nullptr;
        in_scratch.release();
        out_scratch.release();
        pin_in.release();
        pin_out.release();
        if (stream) stream_release(device, stream);
        stream = nullptr;
        if (counted) handle_count(device, -1);
        counted = false;
    }
};

// Handle-less host-pointer entry points (resampling, IQ formats, estimators) borrow a per-thread,
// per-device handle (stream + staging), created on first use and kept for the thread's life:
// no hipMalloc / hipFree -- which also synchronise the device -- per call.  The handles END with their thread
// (comms-rs starts one thread per node, src/node/mod.rs:276-284: a graph that is torn down and rebuilt must not leave
// a pooled stream, device scratch and pinned staging behind per retired node thread).
struct ThreadHandles {
    Handle* h[64] = {};
    ~ThreadHandles() {
        for (Handle*& p : h) {
            if (!p) continue;
            (void)use_device(p->device);
            p->fini();  // scratch and staging freed, the stream back to the device's pool
            delete p;
            p = nullptr;
        }
    }
};
inline comms_status_t thread_handle(int32_t device, Handle** out) {
    static thread_local ThreadHandles th;
    Handle** tl = th.h;
    COMMS_ARG(device >= 0 && device < 64, "device index out of range");
    if (!tl[device]) {
        Handle* nh = new (std::nothrow) Handle;
        COMMS_ARG(nh != nullptr, "out of host memory");
        comms_status_t st = nh->init(device, false);
        if (st != COMMS_OK) {
            delete nh;
            return st;
        }
        tl[device] = nh;
    }
    *out = tl[device];
    return COMMS_OK;
}

inline bool ranges_overlap(const void* a, size_t na, const void* b, size_t nb) {
    const char* pa = static_cast<const char*>(a);
    const char* pb = static_cast<const char*>(b);
    return pa < pb + nb && pb < pa + na;
}

constexpr int kNumCU = 256;  // MI355X: 8 XCD x 32 CU

// Mixer phase bookkeeping shared by the mixer node and the fused chain: phases are
// 64-bit fixed-point fractions ("turns") of T = fl(2*pi), the constant the reference
// wraps with (src/mixer.rs:79-82).
constexpr double kMixT = 2.0 * 3.14159265358979323846264338327950288;
inline uint64_t mix_to_turns(double angle) {
    long double r = fmodl(static_cast<long double>(angle), static_cast<long double>(kMixT));
    if (r < 0) r += static_cast<long double>(kMixT);
    long double t = r / static_cast<long double>(kMixT) * 18446744073709551616.0L;
    if (t >= 18446744073709551616.0L) return 0;
    return static_cast<uint64_t>(t);
}
inline void mix_host_rotor(uint64_t turns, double& c, double& s) {
    double ang = static_cast<double>(turns >> 11) * (kMixT * 0x1.0p-53);
    c = cos(ang);
    s = sin(ang);
}
// Mixer::new (src/mixer.rs:43-51): dphase wrapped into [0, 2pi)
inline double mix_wrap_dphase(double dphase) {
    if (fabs(dphase) > 64.0 * kMixT) dphase = fmod(dphase, kMixT);
    while (dphase >= kMixT) dphase -= kMixT;
    while (dphase < 0.0) dphase += kMixT;
    return dphase;
}

}  // namespace comms
