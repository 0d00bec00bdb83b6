#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
// pattern A: lane (hh, col) writes 16 B at row (tile*32 + col), byte offset 16 hh + 32 e   (32 stores per lane per tile)
// pattern B: lane l writes 16 B at row (tile*32 + 4 k + l / 16), byte offset 16 (l % 16) + 256 qtr  (full 256 B per 16 lanes)
// pattern C: fully contiguous 1 KiB per instruction
template <int PAT>
__global__ __launch_bounds__(256) void st_kernel(float *out, int tiles_per_wave) {
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    f4 v = {1.f * lane, 2.f, 3.f, 4.f};
    for (int t = 0; t < tiles_per_wave; ++t) {
        char *base = reinterpret_cast<char *>(out) + ((size_t)wave * tiles_per_wave + t) * 32768;
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            size_t off;
            if (PAT == 0) off = (size_t)(lane & 31) * 1024 + 16 * (lane >> 5) + 32 * e;
            else if (PAT == 1) off = (size_t)(4 * (e & 7) + (lane >> 4)) * 1024 + 16 * (lane & 15) + 256 * (e >> 3);
            else off = (size_t)e * 1024 + 16 * lane;
            *reinterpret_cast<f4 *>(base + off) = v;
            v.x += 1.f;
        }
    }
}
int main() {
    const size_t bytes = 1ull << 30;
    float *d; hipMalloc(&d, bytes);
    const int waves = 1024, tpw = (int)(bytes / 32768 / waves);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pat = 0; pat < 3; ++pat) for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) {
            if (pat == 0) st_kernel<0><<<waves / 4, 256>>>(d, tpw);
            else if (pat == 1) st_kernel<1><<<waves / 4, 256>>>(d, tpw);
            else st_kernel<2><<<waves / 4, 256>>>(d, tpw);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("pattern %d: %.3f ms per GiB = %.2f TB/s\n", pat, ms / 5, bytes / (ms / 5) / 1e9);
    }
    return 0;
}
