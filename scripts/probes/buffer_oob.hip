// Probe: are raw-buffer range checks on gfx950 applied to voffset + soffset, or to voffset alone?
// build: hipcc --offload-arch=gfx950 -O2 scripts/probes/buffer_oob.hip -o comms_rs_amd/lib/probe_buffer_oob
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__global__ void k(float* buf, unsigned nrec, unsigned voff, unsigned soff, float* res) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf, 0, (int)nrec, 0x00020000);
    unsigned x = __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0);
    res[0] = __uint_as_float(x);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(-5.0f), r, voff, soff, 0);
}
int main() {
    float *d, *res;
    hipMalloc(&d, 4096); hipMalloc(&res, 16);
    float h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = 1.0f + i;
    struct { unsigned nrec, voff, soff; const char* what; } cases[] = {
        {64, 0, 0, "in range"}, {64, 64, 0, "voffset == num_records"}, {64, 0, 64, "soffset == num_records, voffset 0"},
        {64, 32, 32, "voffset + soffset == num_records"}, {64, 60, 0, "last dword"}, {64, 0, 60, "last dword via soffset"},
        {64, 0xFFFFFFF0u, 0, "voffset wrapped negative"}, {64, 0xFFFFFFF0u, 32, "negative voffset + soffset back in range (index 4)"}};
    for (auto& c : cases) {
        hipMemcpy(d, h, 4096, hipMemcpyHostToDevice);
        k<<<1, 1>>>(d, c.nrec, c.voff, c.soff, res);
        float r; hipMemcpy(&r, res, 4, hipMemcpyDeviceToHost);
        hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost);
        int stored = -1;
        for (int i = 0; i < 1024; ++i) if (h[i] == -5.0f) stored = i;
        for (int i = 0; i < 1024; ++i) h[i] = 1.0f + i;
        printf("%-52s nrec=%u voff=%u soff=%u -> load %.1f, store landed at dword %d\n", c.what, c.nrec, c.voff, c.soff, r, stored);
    }
    return 0;
}
