// common.hpp -- shared plumbing of libcomms_hip.so (gfx950 only; no CPU fallback).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/comms_hip.h"

namespace comms {

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

}  // namespace comms

// Pool of hipEvent pairs recorded around a node's dominant kernel (bench/profiling).
struct comms_timer {
    int32_t device = 0;
    size_t n = 0;
    size_t next = 0;  // launches recorded so far (wraps modulo n)
    hipEvent_t* start = nullptr;
    hipEvent_t* stop = nullptr;
};

namespace comms {

// Base of every node handle: device + own stream + scratch for host-pointer runs.
struct Handle {
    int32_t device = 0;
    hipStream_t stream = nullptr;
    Scratch in_scratch, out_scratch;
    comms_timer* timer = nullptr;

    // bracket the dominant kernel launch; no-ops without an attached timer
    void tic(hipStream_t s) {
        if (timer && timer->n) (void)hipEventRecord(timer->start[timer->next % timer->n], s);
    }
    void toc(hipStream_t s) {
        if (timer && timer->n) {
            (void)hipEventRecord(timer->stop[timer->next % timer->n], s);
            ++timer->next;
        }
    }

    comms_status_t init(int32_t dev) {
        COMMS_TRY(use_device(dev));
        device = dev;
        COMMS_HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        return COMMS_OK;
    }
    // `stream` arguments of the C ABI are passed through as HIP does: NULL is the
    // legacy default stream; COMMS_STREAM_HANDLE selects the handle's own stream.
    hipStream_t pick(void* s) const {
        return s == COMMS_STREAM_HANDLE ? stream : reinterpret_cast<hipStream_t>(s);
    }
    void fini() {
        in_scratch.release();
        out_scratch.release();
        if (stream) (void)hipStreamDestroy(stream);
        stream = nullptr;
    }
};

inline bool ranges_overlap(const void* a, size_t na, const void* b, size_t nb) {
    const char* pa = static_cast<const char*>(a);
    const char* pb = static_cast<const char*>(b);
    return pa < pb + nb && pb < pa + na;
}

constexpr int kNumCU = 256;  // MI355X: 8 XCD x 32 CU

}  // namespace comms
