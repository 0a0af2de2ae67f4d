// host_path.hip -- what the host-pointer (Vec<Complex<f32>>) path can reach on this box (round 5):
//   (1) hipMemcpy from / to PAGEABLE memory (what comms_*_run did until round 4), one direction at a time;
//   (2) hipMemcpyAsync from / to PINNED staging in chunks, one direction and both directions at once (two streams);
//   (3) CPU memcpy pageable -> pinned with 1 ... 16 threads (the copy a staging pipeline adds).
// build: hipcc -O2 --offload-arch=gfx950 -o comms_rs_amd/lib/host_path scripts/probes/host_path.hip -lpthread
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_));                  \
            return 1;                                                              \
        }                                                                          \
    } while (0)

static double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void par_copy(char* dst, const char* src, size_t bytes, int threads) {
    if (threads <= 1) {
        memcpy(dst, src, bytes);
        return;
    }
    std::vector<std::thread> th;
    const size_t per = (bytes / threads + 4095) & ~size_t(4095);
    for (int t = 0; t < threads; ++t) {
        const size_t a = per * t, b = a + per < bytes ? a + per : bytes;
        if (a >= bytes) break;
        th.emplace_back([=] { memcpy(dst + a, src + a, b - a); });
    }
    for (auto& t : th) t.join();
}

int main(int argc, char** argv) {
    const size_t total = size_t(argc > 1 ? atoi(argv[1]) : 128) << 20;
    char* page_in = static_cast<char*>(malloc(total));
    char* page_out = static_cast<char*>(malloc(total));
    memset(page_in, 1, total);
    memset(page_out, 2, total);
    char *d_a, *d_b;
    CK(hipMalloc(&d_a, total));
    CK(hipMalloc(&d_b, total));
    printf("hardware threads %u, buffer %zu MiB\n", std::thread::hardware_concurrency(), total >> 20);
    // (1) pageable, synchronous
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now();
        CK(hipMemcpy(d_a, page_in, total, hipMemcpyHostToDevice));
        double t1 = now();
        CK(hipMemcpy(page_out, d_b, total, hipMemcpyDeviceToHost));
        double t2 = now();
        printf("pageable hipMemcpy: H2D %.1f GB/s, D2H %.1f GB/s\n", total / (t1 - t0) / 1e9, total / (t2 - t1) / 1e9);
    }
    // (2) pinned staging, chunked
    hipStream_t s_in, s_out;
    CK(hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking));
    for (size_t chunk_mb : {2, 4, 8, 16, 32}) {
        const size_t chunk = chunk_mb << 20, nch = total / chunk;
        char *pin_a, *pin_b;
        CK(hipHostMalloc(&pin_a, total, hipHostMallocDefault));
        CK(hipHostMalloc(&pin_b, total, hipHostMallocDefault));
        memset(pin_a, 3, total);
        memset(pin_b, 4, total);
        double t0 = now();
        for (size_t k = 0; k < nch; ++k) CK(hipMemcpyAsync(d_a + k * chunk, pin_a + k * chunk, chunk, hipMemcpyHostToDevice, s_in));
        CK(hipStreamSynchronize(s_in));
        double t1 = now();
        for (size_t k = 0; k < nch; ++k) CK(hipMemcpyAsync(pin_b + k * chunk, d_b + k * chunk, chunk, hipMemcpyDeviceToHost, s_out));
        CK(hipStreamSynchronize(s_out));
        double t2 = now();
        for (size_t k = 0; k < nch; ++k) {
            CK(hipMemcpyAsync(d_a + k * chunk, pin_a + k * chunk, chunk, hipMemcpyHostToDevice, s_in));
            CK(hipMemcpyAsync(pin_b + k * chunk, d_b + k * chunk, chunk, hipMemcpyDeviceToHost, s_out));
        }
        CK(hipStreamSynchronize(s_in));
        CK(hipStreamSynchronize(s_out));
        double t3 = now();
        printf("pinned, chunks of %2zu MiB: H2D %.1f GB/s, D2H %.1f GB/s, both at once %.1f GB/s each way\n", chunk_mb,
               total / (t1 - t0) / 1e9, total / (t2 - t1) / 1e9, total / (t3 - t2) / 1e9);
        CK(hipHostFree(pin_a));
        CK(hipHostFree(pin_b));
    }
    // (3) CPU copies pageable <-> pinned
    {
        char* pin;
        CK(hipHostMalloc(&pin, total, hipHostMallocDefault));
        memset(pin, 5, total);
        for (int th : {1, 2, 4, 8, 12, 16}) {
            double best_in = 1e9, best_out = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                double t0 = now();
                par_copy(pin, page_in, total, th);
                double t1 = now();
                par_copy(page_out, pin, total, th);
                double t2 = now();
                best_in = t1 - t0 < best_in ? t1 - t0 : best_in;
                best_out = t2 - t1 < best_out ? t2 - t1 : best_out;
            }
            printf("CPU memcpy, %2d threads (incl. thread start): pageable -> pinned %.1f GB/s, pinned -> pageable %.1f GB/s\n", th,
                   total / best_in / 1e9, total / best_out / 1e9);
        }
        // chunks of 8 MiB, threads started per chunk (what a pipeline stage sees)
        for (int th : {1, 4, 8}) {
            const size_t chunk = 8u << 20;
            double t0 = now();
            for (size_t off = 0; off < total; off += chunk) par_copy(pin + off, page_in + off, chunk, th);
            double t1 = now();
            printf("CPU memcpy in 8-MiB pieces, %d threads started per piece: %.1f GB/s\n", th, total / (t1 - t0) / 1e9);
        }
        CK(hipHostFree(pin));
    }
    // (4) hipHostRegister of the caller's pageable buffer (pin in place), then async copy
    {
        double t0 = now();
        hipError_t e = hipHostRegister(page_in, total, hipHostRegisterDefault);
        double t1 = now();
        if (e == hipSuccess) {
            CK(hipMemcpyAsync(d_a, page_in, total, hipMemcpyHostToDevice, s_in));
            CK(hipStreamSynchronize(s_in));
            double t2 = now();
            CK(hipHostUnregister(page_in));
            double t3 = now();
            printf("hipHostRegister %zu MiB: %.2f ms, copy %.1f GB/s, unregister %.2f ms\n", total >> 20, (t1 - t0) * 1e3, total / (t2 - t1) / 1e9, (t3 - t2) * 1e3);
        } else
            printf("hipHostRegister failed: %s\n", hipGetErrorString(e));
    }
    return 0;
}
