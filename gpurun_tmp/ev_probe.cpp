#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
int main() {
    int bad = 0;
    for (int rep = 0; rep < 300; ++rep) {
        hipStream_t s;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return 2;
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return 3;
        void* p = nullptr;
        (void)hipMalloc(&p, 1 << 16);
        (void)hipMemsetAsync(p, 0, 1 << 16, s);
        (void)hipEventRecord(e, s);
        (void)hipStreamSynchronize(s);
        (void)hipStreamDestroy(s);
        // churn the heap a little so that the stream object's memory is reused
        std::vector<hipStream_t> t(4);
        for (auto& x : t) (void)hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
        for (auto& x : t) (void)hipStreamDestroy(x);
        std::vector<char*> junk;
        for (int i = 0; i < 400; ++i) {
            const size_t sz = 256u << (i % 8);  // 256 B .. 32 KiB: whatever size class the runtime's stream object came from
            junk.push_back(new char[sz]);
            int* w = reinterpret_cast<int*>(junk.back());
            for (size_t k = 0; k < sz / 4; ++k) w[k] = 1;  // "capture status active" if read as an enum
        }
        hipError_t r = hipEventSynchronize(e);
        if (r != hipSuccess) { if (bad < 5) std::printf("rep %d: hipEventSynchronize after stream destroy: %s\n", rep, hipGetErrorString(r)); ++bad; }
        hipStream_t s2;
        (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
        r = hipStreamWaitEvent(s2, e, 0);
        if (r != hipSuccess) { if (bad < 5) std::printf("rep %d: hipStreamWaitEvent: %s\n", rep, hipGetErrorString(r)); ++bad; }
        (void)hipStreamSynchronize(s2);
        (void)hipStreamDestroy(s2);
        for (auto q : junk) delete[] q;
        (void)hipEventDestroy(e);
        (void)hipFree(p);
    }
    std::printf("bad = %d of 300\n", bad);
    return 0;
}
