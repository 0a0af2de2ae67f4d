// runtime.hip -- library-wide plumbing: errors, device selection, ref-counted
// device buffers (the DeviceBuf<T> message type), synthetic IQ generator.
#include <atomic>
#include <mutex>
#include <vector>

#include "common.hpp"

namespace comms {

size_t zero_copy_limit() {
    static const size_t lim = [] {
        const char* v = getenv("COMMS_ZERO_COPY_BYTES");
        return static_cast<size_t>(v && *v ? atol(v) : (1u << 20));
    }();
    return lim;
}


size_t host_pipe_bytes() {
    static const size_t lim = [] {
        const char* v = getenv("COMMS_HOST_PIPE_BYTES");
        const long long x = v && *v ? atoll(v) : static_cast<long long>(Handle::kHostPipeBytes);
        return x <= 0 ? ~static_cast<size_t>(0) : static_cast<size_t>(x);
    }();
    return lim;
}

size_t host_chunk_bytes() {
    static const size_t b = [] {
        const char* v = getenv("COMMS_HOST_CHUNK_BYTES");
        const long long x = v && *v ? atoll(v) : static_cast<long long>(Handle::kHostChunkBytes);
        return x < 4096 ? static_cast<size_t>(4096) : static_cast<size_t>(x);
    }();
    return b;
}

comms_status_t use_device(int32_t device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(COMMS_ERR_DEVICE,
                    "no usable HIP device (%s); libcomms_hip has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= n)
        return fail(COMMS_ERR_ARG, "device %d out of range [0,%d)", device, n);
    COMMS_HIP_TRY(hipSetDevice(device));
    return COMMS_OK;
}

// SplitMix64 finaliser on a counter: stateless, so any index range can be
// produced anywhere.  Word k of the stream = mix(seed + (k+1)*golden).
__host__ __device__ inline uint64_t splitmix(uint64_t seed, uint64_t k) {
    uint64_t z = seed + (k + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ inline float unit_pm1(uint64_t w) {
    // top 24 bits -> [0, 2) in steps of 2^-23, minus 1 -> [-1, 1)
    return static_cast<float>(static_cast<uint32_t>(w >> 40)) * (1.0f / 8388608.0f) - 1.0f;
}

__global__ void synth_iq_kernel(float2* out, size_t n, uint64_t first, uint64_t seed) {
    size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t g = first + i;
        out[i] = make_float2(unit_pm1(splitmix(seed, 2 * g)), unit_pm1(splitmix(seed, 2 * g + 1)));
    }
}

// ---- stream pool (common.hpp: why streams are recycled rather than destroyed)
static std::mutex g_stream_m;
static std::vector<hipStream_t> g_free_streams[64];

static std::atomic<long> g_streams_created[64];  // (what the pool has ever created: the diagnostic build reports it)
static std::vector<hipStream_t> g_pool_streams[64];  // every stream the pool created and has not destroyed (under g_stream_m)
comms_status_t stream_acquire(int32_t device, hipStream_t* out) {
    COMMS_ARG(device >= 0 && device < 64, "device index out of range");
    COMMS_TRY(use_device(device));
    {
        std::lock_guard<std::mutex> lk(g_stream_m);
        auto& fl = g_free_streams[device];
        if (!fl.empty()) {
            *out = fl.back();
            fl.pop_back();
            return COMMS_OK;
        }
    }
    COMMS_HIP_TRY(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
    g_streams_created[device].fetch_add(1);
    {
        std::lock_guard<std::mutex> lk(g_stream_m);
        g_pool_streams[device].push_back(*out);
    }
    return COMMS_OK;
}

static std::atomic<long> g_live_handles[64];
static std::atomic<long> g_followers[64];
void follower_count(int32_t device, int delta) {
    if (device >= 0 && device < 64) g_followers[device].fetch_add(delta);
}
bool pool_owns(int32_t device, hipStream_t s) {
    if (device < 0 || device >= 64) return false;
    std::lock_guard<std::mutex> lk(g_stream_m);
    for (hipStream_t q : g_pool_streams[device])
        if (q == s) return true;
    return false;
}
void handle_count(int32_t device, int delta) {
    if (device >= 0 && device < 64) g_live_handles[device].fetch_add(delta);
}

void stream_release(int32_t device, hipStream_t s) {
    if (!s || device < 0 || device >= 64) return;
    if (use_device(device) == COMMS_OK) (void)hipStreamSynchronize(s);  // the next owner starts on an idle stream
    std::lock_guard<std::mutex> lk(g_stream_m);
    g_free_streams[device].push_back(s);
}

long streams_created(int32_t device) { return device >= 0 && device < 64 ? g_streams_created[device].load() : -1; }

}  // namespace comms

using namespace comms;

#ifdef COMMS_DIAG
// diagnostic build: streams the pool of `device` has created so far (tests/test_gpu_host_threads.py: node threads that
// come and go must reuse them)
extern "C" long comms_debug_streams_created(int32_t device) { return comms::streams_created(device); }
#endif

// Device-resident message.  Besides the allocation it carries what lets node threads hand it
// from stream to stream without ever synchronising the device:
//   ready -- recorded by the producer after the launch that fills the buffer; a consumer's
//            stream waits on it (hipStreamWaitEvent) before its own launch;
//   uses  -- one event per consumer launch that reads the buffer; the memory goes back to the
//            pool at refcount 0 and is handed out again only once these have completed.
struct comms_buf {
    void* ptr;
    size_t bytes;      // what the caller asked for
    size_t cap;        // size class actually allocated
    int32_t device;
    std::atomic<int> refs;
    std::mutex m;      // clones of one buffer live on different node threads
    hipEvent_t ready = nullptr;
    bool ready_set = false;
    std::vector<hipEvent_t> uses;
};

namespace {

// Per-device cache of released allocations by size class and of events, so that a steady-state
// graph does no hipMalloc / hipFree / hipEventCreate per message (hipFree synchronises the whole
// device).  Size classes: powers of two from 256 B up to 1 MiB, four per octave above (a block is
// at most 25 % larger than the request; a plain power of two could double a multi-GiB message).
// COMMS_BUF_POOL_MB caps the cached bytes per device (default 8192; 0 disables caching).
struct PoolBlock {
    void* ptr;
    std::vector<hipEvent_t> uses;  // readers that may still be running
};
constexpr int kMaxBufLog2 = 40;    // 1 TiB: beyond any device; larger requests are argument errors
constexpr int kNumClasses = 13 + 4 * (kMaxBufLog2 - 20) + 1;
struct DevicePool {
    std::mutex m;
    std::vector<PoolBlock> free_blocks[kNumClasses];
    std::vector<hipEvent_t> free_events;
    size_t cached_bytes = 0;
    std::atomic<long> live_bufs{0};  // comms_buf objects alive on this device (stream-pool trim refuses while > 0)
};
DevicePool g_pools[64];
size_t pool_cap_bytes() {
    static const size_t cap = [] {
        const char* v = getenv("COMMS_BUF_POOL_MB");
        return static_cast<size_t>(v && *v ? atol(v) : 8192) << 20;
    }();
    return cap;
}
// bytes in [1, 2^kMaxBufLog2] (checked by the caller) -> class index and the class's size
int size_class(size_t bytes, size_t* cap) {
    if (bytes <= (static_cast<size_t>(1) << 20)) {
        int c = 8;
        while (c < 20 && (static_cast<size_t>(1) << c) < bytes) ++c;
        *cap = static_cast<size_t>(1) << c;
        return c - 8;  // 0 .. 12
    }
    int o = 20;  // octave: 2^o < bytes <= 2^(o+1)
    while (o + 1 < kMaxBufLog2 && (static_cast<size_t>(1) << (o + 1)) < bytes) ++o;
    const size_t step = static_cast<size_t>(1) << (o - 2);
    const size_t q = (bytes - (static_cast<size_t>(1) << o) + step - 1) / step;  // 1 .. 4
    *cap = (static_cast<size_t>(1) << o) + q * step;
    return 13 + 4 * (o - 20) + static_cast<int>(q) - 1;
}
hipEvent_t pool_event(DevicePool& p) {  // caller holds no lock; current device is the pool's
    {
        std::lock_guard<std::mutex> lk(p.m);
        if (!p.free_events.empty()) {
            hipEvent_t e = p.free_events.back();
            p.free_events.pop_back();
            return e;
        }
    }
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    return e;
}
void pool_return_events(DevicePool& p, std::vector<hipEvent_t>& evs) {
    std::lock_guard<std::mutex> lk(p.m);
    for (hipEvent_t e : evs) p.free_events.push_back(e);
    evs.clear();
}

}  // namespace

extern "C" {

const char* comms_version(void) { return "comms_hip 0.1.0 (gfx950)"; }
const char* comms_last_error(void) { return err_buf(); }

comms_status_t comms_device_count(int32_t* out_n) {
    COMMS_ARG(out_n != nullptr, "out_n is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *out_n = 0;
        return fail(COMMS_ERR_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *out_n = n;
    return COMMS_OK;
}

comms_status_t comms_device_info(int32_t device, char* name, size_t name_cap, int32_t* out_cus,
                                 uint64_t* out_hbm_bytes) {
    COMMS_TRY(use_device(device));
    hipDeviceProp_t p;
    COMMS_HIP_TRY(hipGetDeviceProperties(&p, device));
    if (name && name_cap) snprintf(name, name_cap, "%s (%s)", p.name, p.gcnArchName);
    if (out_cus) *out_cus = p.multiProcessorCount;
    if (out_hbm_bytes) *out_hbm_bytes = p.totalGlobalMem;
    return COMMS_OK;
}

comms_status_t comms_buf_alloc(size_t bytes, int32_t device, comms_buf_t** out) {
    COMMS_ARG(out != nullptr, "out is NULL");
    *out = nullptr;
    COMMS_ARG(device >= 0 && device < 64, "device index out of range");
    COMMS_ARG(bytes <= (static_cast<size_t>(1) << kMaxBufLog2), "%zu bytes is not a device allocation (limit 2^%d)",
              bytes, kMaxBufLog2);
    COMMS_TRY(use_device(device));
    comms_buf* b = new (std::nothrow) comms_buf;
    COMMS_ARG(b != nullptr, "out of host memory");
    b->ptr = nullptr;
    b->bytes = bytes;
    b->cap = 0;
    b->device = device;
    b->refs.store(1);
    if (bytes) {
        const int cls = size_class(bytes, &b->cap);
        DevicePool& p = g_pools[device];
        PoolBlock blk{nullptr, {}};
        {
            std::lock_guard<std::mutex> lk(p.m);
            auto& fl = p.free_blocks[cls];
            if (!fl.empty()) {
                blk = std::move(fl.back());
                fl.pop_back();
                p.cached_bytes -= b->cap;
            }
        }
        if (blk.ptr) {
            // readers of the block's previous life: normally long finished, so this does not block.
            // A block whose users never recorded anything (callers that pass raw pointers to
            // *_run_dev on streams of their own) gets what hipFree used to give them: a device sync.
            hipError_t e = blk.uses.empty() ? hipDeviceSynchronize() : hipSuccess;
            for (hipEvent_t ev : blk.uses)
                if (e == hipSuccess) e = hipEventSynchronize(ev);
            pool_return_events(p, blk.uses);
            if (e != hipSuccess) {
                (void)hipFree(blk.ptr);
                delete b;
                return fail(COMMS_ERR_DEVICE, "waiting for a recycled buffer's readers: %s", hipGetErrorString(e));
            }
            b->ptr = blk.ptr;
        } else {
            hipError_t e = hipMalloc(&b->ptr, b->cap);
            if (e != hipSuccess) {
                comms_buf_pool_trim(device);  // give the cache back and try once more
                e = hipMalloc(&b->ptr, b->cap);
            }
            if (e != hipSuccess) {
                delete b;
                return fail(COMMS_ERR_DEVICE, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
            }
        }
    }
    g_pools[device].live_bufs.fetch_add(1);
    *out = b;
    return COMMS_OK;
}
comms_status_t comms_buf_retain(comms_buf_t* b) {
    COMMS_ARG(b != nullptr, "buffer is NULL");
    b->refs.fetch_add(1);
    return COMMS_OK;
}
comms_status_t comms_buf_release(comms_buf_t* b) {
    COMMS_ARG(b != nullptr, "buffer is NULL");
    if (b->refs.fetch_sub(1) == 1) {
        DevicePool& p = g_pools[b->device];
        if (b->ready) {
            if (b->ready_set && b->uses.empty() && b->ptr) {
                // filled by a launch that no reader followed: that launch is the block's last user
                b->uses.push_back(b->ready);
            } else {  // the producer's launch precedes every reader's: the readers' events cover it
                std::lock_guard<std::mutex> lk(p.m);
                p.free_events.push_back(b->ready);
            }
            b->ready = nullptr;
        }
        if (b->ptr) {
            bool cached = false;
            size_t cap = 0;
            const int cls = size_class(b->cap, &cap);
            {
                std::lock_guard<std::mutex> lk(p.m);
                if (p.cached_bytes + b->cap <= pool_cap_bytes()) {
                    p.free_blocks[cls].push_back(PoolBlock{b->ptr, std::move(b->uses)});
                    p.cached_bytes += b->cap;
                    cached = true;
                }
            }
            if (!cached) {
                COMMS_TRY(use_device(b->device));
                COMMS_HIP_TRY(hipFree(b->ptr));  // synchronises the device: every reader is done
                pool_return_events(p, b->uses);
            }
        }
        p.live_bufs.fetch_sub(1);
        delete b;
    }
    return COMMS_OK;
}
// Frees every cached allocation of `device` (the events stay pooled).
comms_status_t comms_buf_pool_trim(int32_t device) {
    COMMS_ARG(device >= 0 && device < 64, "device index out of range");
    DevicePool& p = g_pools[device];
    std::vector<PoolBlock> all;
    {
        std::lock_guard<std::mutex> lk(p.m);
        for (auto& fl : p.free_blocks) {
            for (auto& blk : fl) all.push_back(std::move(blk));
            fl.clear();
        }
        p.cached_bytes = 0;
    }
    if (all.empty()) return COMMS_OK;
    COMMS_TRY(use_device(device));
    for (auto& blk : all) {
        (void)hipFree(blk.ptr);
        pool_return_events(p, blk.uses);
    }
    return COMMS_OK;
}

// ---- ordering that travels with the buffer (device-resident graphs; nodes.hpp *Dev nodes)
comms_status_t comms_buf_record_ready(comms_buf_t* b, void* stream) {
    COMMS_ARG(b != nullptr, "buffer is NULL");
    COMMS_TRY(use_device(b->device));
    std::lock_guard<std::mutex> lk(b->m);
    if (!b->ready) {
        b->ready = pool_event(g_pools[b->device]);
        if (!b->ready) return fail(COMMS_ERR_DEVICE, "hipEventCreate failed");
    }
    COMMS_HIP_TRY(hipEventRecord(b->ready, reinterpret_cast<hipStream_t>(stream)));
    b->ready_set = true;
    return COMMS_OK;
}
comms_status_t comms_buf_wait_ready(comms_buf_t* b, void* stream) {
    COMMS_ARG(b != nullptr, "buffer is NULL");
    std::lock_guard<std::mutex> lk(b->m);
    if (!b->ready_set) return COMMS_OK;  // filled synchronously (upload) or never written
    COMMS_HIP_TRY(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), b->ready, 0));
    return COMMS_OK;
}
comms_status_t comms_buf_record_use(comms_buf_t* b, void* stream) {
    COMMS_ARG(b != nullptr, "buffer is NULL");
    COMMS_TRY(use_device(b->device));
    hipEvent_t e = pool_event(g_pools[b->device]);
    if (!e) return fail(COMMS_ERR_DEVICE, "hipEventCreate failed");
    hipError_t err = hipEventRecord(e, reinterpret_cast<hipStream_t>(stream));
    std::lock_guard<std::mutex> lk(b->m);
    b->uses.push_back(e);
    if (err != hipSuccess) return fail(COMMS_ERR_DEVICE, "hipEventRecord: %s", hipGetErrorString(err));
    return COMMS_OK;
}
comms_status_t comms_buf_sync(comms_buf_t* b) {
    COMMS_ARG(b != nullptr, "buffer is NULL");
    std::lock_guard<std::mutex> lk(b->m);
    if (b->ready_set) COMMS_HIP_TRY(hipEventSynchronize(b->ready));
    return COMMS_OK;
}

// ---- plain streams for host graph nodes (one per node thread); recycled, see stream_acquire
comms_status_t comms_stream_create(int32_t device, void** out_stream) {
    COMMS_ARG(out_stream != nullptr, "out_stream is NULL");
    *out_stream = nullptr;
    hipStream_t s = nullptr;
    COMMS_TRY(comms::stream_acquire(device, &s));
    *out_stream = s;
    return COMMS_OK;
}
comms_status_t comms_stream_synchronize(int32_t device, void* stream) {
    COMMS_TRY(use_device(device));
    COMMS_HIP_TRY(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
    return COMMS_OK;
}
comms_status_t comms_stream_destroy(int32_t device, void* stream) {
    if (!stream) return COMMS_OK;
    COMMS_ARG(device >= 0 && device < 64, "device index out of range");
    comms::stream_release(device, reinterpret_cast<hipStream_t>(stream));
    return COMMS_OK;
}
// Destroys the idle streams of `device`'s pool.  Streams are recycled rather than destroyed because events
// that travel with comms_buf allocations keep referring to the stream that recorded them; so this first
// requires that no comms_buf is alive on the device, then frees the buffer cache, drains the device and
// destroys the pooled events (all of them are free at that point) before the streams go.
comms_status_t comms_stream_pool_trim(int32_t device) {
    COMMS_ARG(device >= 0 && device < 64, "device index out of range");
    DevicePool& p = g_pools[device];
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> events;
    {
        // the checks and the swaps under both locks: no buffer can be allocated (its events come from this pool) and no
        // stream handed out between them
        std::lock_guard<std::mutex> ls(comms::g_stream_m);
        std::lock_guard<std::mutex> lp(p.m);
        COMMS_ARG(p.live_bufs.load() == 0, "%ld comms_buf objects are alive on device %d: release them first",
                  p.live_bufs.load(), device);
        // (round 5: not "any live handle" -- a long-running host that retired node threads could then never trim without
        // tearing its graph down -- but handles that FOLLOW a pooled stream other than their own: only those can be left
        // holding a stream this call destroys.  A state getter / setter, or the handle's next launch on another stream,
        // detaches it.)
        COMMS_ARG(comms::g_followers[device].load() == 0,
                  "%ld node handles on device %d still follow a pooled stream that is not their own (quiesce them -- any state "
                  "getter does -- or destroy them first)", comms::g_followers[device].load(), device);
        streams.swap(comms::g_free_streams[device]);
        events.swap(p.free_events);
        for (hipStream_t s : streams) {
            auto& all = comms::g_pool_streams[device];
            for (size_t i = 0; i < all.size(); ++i)
                if (all[i] == s) {
                    all[i] = all.back();
                    all.pop_back();
                    break;
                }
        }
    }
    const comms_status_t st = comms_buf_pool_trim(device);
    {
        // the cached blocks' use-events came back into the pool's free list while the buffer cache was emptied: they
        // were last recorded on streams that are about to go, so they go too
        std::lock_guard<std::mutex> lp(p.m);
        events.insert(events.end(), p.free_events.begin(), p.free_events.end());
        p.free_events.clear();
    }
    if (streams.empty() && events.empty()) return st;
    COMMS_TRY(use_device(device));
    COMMS_HIP_TRY(hipDeviceSynchronize());
    for (hipEvent_t e : events) (void)hipEventDestroy(e);
    for (hipStream_t s : streams) (void)hipStreamDestroy(s);
    return st;
}
comms_status_t comms_buf_upload(comms_buf_t* b, size_t offset, const void* host, size_t bytes) {
    COMMS_ARG(b && (host || !bytes), "NULL argument");
    COMMS_ARG(offset <= b->bytes && bytes <= b->bytes - offset, "upload of %zu at %zu exceeds %zu",
              bytes, offset, b->bytes);
    COMMS_TRY(use_device(b->device));
    COMMS_TRY(comms_buf_sync(b));  // a pending launch that fills the buffer comes first
    if (bytes)
        COMMS_HIP_TRY(hipMemcpy(static_cast<char*>(b->ptr) + offset, host, bytes,
                                hipMemcpyHostToDevice));
    return COMMS_OK;
}
comms_status_t comms_buf_download(const comms_buf_t* b, size_t offset, void* host, size_t bytes) {
    COMMS_ARG(b && (host || !bytes), "NULL argument");
    COMMS_ARG(offset <= b->bytes && bytes <= b->bytes - offset,
              "download of %zu at %zu exceeds %zu", bytes, offset, b->bytes);
    COMMS_TRY(use_device(b->device));
    COMMS_TRY(comms_buf_sync(const_cast<comms_buf_t*>(b)));  // the producer's launch (on its own stream) comes first
    if (bytes)
        COMMS_HIP_TRY(hipMemcpy(host, static_cast<const char*>(b->ptr) + offset, bytes,
                                hipMemcpyDeviceToHost));
    return COMMS_OK;
}
void* comms_buf_ptr(const comms_buf_t* b) { return b ? b->ptr : nullptr; }
size_t comms_buf_size(const comms_buf_t* b) { return b ? b->bytes : 0; }
int32_t comms_buf_device(const comms_buf_t* b) { return b ? b->device : -1; }

comms_status_t comms_timer_create(size_t n_pairs, int32_t device, comms_timer_t** out) {
    COMMS_ARG(out != nullptr, "out is NULL");
    *out = nullptr;
    COMMS_ARG(n_pairs >= 1 && n_pairs <= (1u << 20), "n_pairs out of range");
    COMMS_TRY(use_device(device));
    comms_timer* t = new (std::nothrow) comms_timer;
    COMMS_ARG(t != nullptr, "out of host memory");
    t->device = device;
    t->n = n_pairs;
    t->start = new (std::nothrow) hipEvent_t[n_pairs]();
    t->stop = new (std::nothrow) hipEvent_t[n_pairs]();
    hipError_t e = (t->start && t->stop) ? hipSuccess : hipErrorOutOfMemory;
    for (size_t i = 0; i < n_pairs && e == hipSuccess; ++i) {
        e = hipEventCreate(&t->start[i]);
        if (e == hipSuccess) e = hipEventCreate(&t->stop[i]);
    }
    if (e != hipSuccess) {  // undo what exists: no events, arrays or timer may leak
        comms_timer_destroy(t);
        return fail(COMMS_ERR_DEVICE, "timer of %zu event pairs: %s", n_pairs, hipGetErrorString(e));
    }
    *out = t;
    return COMMS_OK;
}
static comms_status_t timer_init_stamps(comms_timer* t) {
    const size_t bytes = t->ns * kStampSlots * sizeof(unsigned long long);
    COMMS_HIP_TRY(hipMemsetAsync(t->d_begin, 0xFF, bytes, nullptr));
    COMMS_HIP_TRY(hipMemsetAsync(t->d_end, 0, bytes, nullptr));
    COMMS_HIP_TRY(hipStreamSynchronize(nullptr));  // the handles' streams do not wait for the legacy stream
    return COMMS_OK;
}
comms_status_t comms_timer_add_stamps(comms_timer_t* t, size_t n_launches) {
    COMMS_ARG(t != nullptr, "timer is NULL");
    COMMS_ARG(n_launches >= 1 && n_launches <= (1u << 20), "n_launches out of range");
    COMMS_ARG(t->d_begin == nullptr, "the timer has stamp slots already");
    COMMS_TRY(use_device(t->device));
    const size_t bytes = n_launches * kStampSlots * sizeof(unsigned long long);
    hipError_t e = hipMalloc(&t->d_begin, bytes);
    if (e == hipSuccess) e = hipMalloc(&t->d_end, bytes);
    if (e != hipSuccess) {
        if (t->d_begin) (void)hipFree(t->d_begin);
        t->d_begin = t->d_end = nullptr;
        return fail(COMMS_ERR_DEVICE, "stamp slots for %zu launches: %s", n_launches, hipGetErrorString(e));
    }
    t->ns = n_launches;
    t->snext = 0;
    return timer_init_stamps(t);
}
comms_status_t comms_timer_create_stamps(size_t n_launches, int32_t device, comms_timer_t** out) {
    COMMS_ARG(out != nullptr, "out is NULL");
    *out = nullptr;
    COMMS_TRY(use_device(device));
    comms_timer* t = new (std::nothrow) comms_timer;
    COMMS_ARG(t != nullptr, "out of host memory");
    t->device = device;
    const comms_status_t st = comms_timer_add_stamps(t, n_launches);
    if (st != COMMS_OK) {
        comms_timer_destroy(t);
        return st;
    }
    *out = t;
    return COMMS_OK;
}
comms_status_t comms_timer_set_stride(comms_timer_t* t, size_t stride) {
    COMMS_ARG(t != nullptr, "timer is NULL");
    COMMS_ARG(stride >= 1, "stride must be at least 1");
    t->stride = stride;
    return COMMS_OK;
}
comms_status_t comms_timer_reset(comms_timer_t* t) {
    COMMS_ARG(t != nullptr, "timer is NULL");
    if (t->d_begin) {  // launches that still stamp the old slots come first
        COMMS_TRY(use_device(t->device));
        COMMS_HIP_TRY(hipDeviceSynchronize());
        COMMS_TRY(timer_init_stamps(t));
    }
    t->next = t->seq = t->snext = 0;
    return COMMS_OK;
}
comms_status_t comms_timer_read_stamps(comms_timer_t* t, float* ms, size_t cap, size_t* out_count) {
    COMMS_ARG(t && out_count && (ms || !cap), "NULL argument");
    COMMS_TRY(use_device(t->device));
    const size_t have = t->snext < cap ? t->snext : cap;
    *out_count = 0;
    if (!have) return COMMS_OK;
    COMMS_HIP_TRY(hipDeviceSynchronize());
    std::vector<unsigned long long> b(have * kStampSlots), e(have * kStampSlots);
    COMMS_HIP_TRY(hipMemcpy(b.data(), t->d_begin, b.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    COMMS_HIP_TRY(hipMemcpy(e.data(), t->d_end, e.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < have; ++i) {
        unsigned long long lo = ~0ull, hi = 0;
        for (int k = 0; k < kStampSlots; ++k) {
            lo = b[i * kStampSlots + k] < lo ? b[i * kStampSlots + k] : lo;
            hi = e[i * kStampSlots + k] > hi ? e[i * kStampSlots + k] : hi;
        }
        // s_memrealtime ticks at 100 MHz; a launch whose kernel takes no KStamp left its slots untouched
        ms[i] = hi > lo ? static_cast<float>(static_cast<double>(hi - lo) * 1e-5) : 0.0f;
    }
    *out_count = have;
    return COMMS_OK;
}
comms_status_t comms_timer_read(comms_timer_t* t, float* ms, size_t cap, size_t* out_count) {
    COMMS_ARG(t && out_count && (ms || !cap), "NULL argument");
    COMMS_TRY(use_device(t->device));
    if (!t->n) return comms_timer_read_stamps(t, ms, cap, out_count);  // a stamps-only timer
    const size_t have = t->next < t->n ? t->next : t->n;
    const size_t first = t->next - have;  // oldest launch still held
    size_t w = 0;
    for (size_t i = 0; i < have && w < cap; ++i, ++w) {
        const size_t slot = (first + i) % t->n;
        COMMS_HIP_TRY(hipEventSynchronize(t->stop[slot]));
        COMMS_HIP_TRY(hipEventElapsedTime(&ms[w], t->start[slot], t->stop[slot]));
    }
    *out_count = w;
    return COMMS_OK;
}
comms_status_t comms_timer_destroy(comms_timer_t* t) {
    if (!t) return COMMS_OK;
    (void)use_device(t->device);
    for (size_t i = 0; i < t->n; ++i) {
        if (t->start && t->start[i]) (void)hipEventDestroy(t->start[i]);
        if (t->stop && t->stop[i]) (void)hipEventDestroy(t->stop[i]);
    }
    delete[] t->start;
    delete[] t->stop;
    if (t->d_begin) (void)hipFree(t->d_begin);
    if (t->d_end) (void)hipFree(t->d_end);
    delete t;
    return COMMS_OK;
}

comms_status_t comms_synth_iq_dev(comms_c32* d_out, size_t n, uint64_t first_index,
                                  uint64_t seed, int32_t device, void* stream) {
    COMMS_ARG(d_out || !n, "d_out is NULL");
    COMMS_TRY(use_device(device));
    if (!n) return COMMS_OK;
    size_t blocks = (n + 255) / 256;
    if (blocks > 8 * kNumCU) blocks = 8 * kNumCU;
    synth_iq_kernel<<<dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                      reinterpret_cast<hipStream_t>(stream)>>>(reinterpret_cast<float2*>(d_out),
                                                               n, first_index, seed);
    return launch_ok("synth_iq_kernel");
}
void comms_synth_iq_host(comms_c32* out, size_t n, uint64_t first_index, uint64_t seed) {
    for (size_t i = 0; i < n; ++i) {
        uint64_t g = first_index + i;
        out[i].re = unit_pm1(splitmix(seed, 2 * g));
        out[i].im = unit_pm1(splitmix(seed, 2 * g + 1));
    }
}

}  // extern "C"
