// Sustained fp32 MFMA rate of the box (v_mfma_f32_16x16x4_f32, the instruction every fused kernel is built on):
// 8 independent accumulators per wave, 8 waves per CU, ~1 ms per launch.  Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters, float a0) {
    f4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x * 1e-6f, b = 1.0f + threadIdx.x * 1e-7f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (s == 12345.678f) out[0] = s;
}
int main() {
    float *d; hipMalloc(&d, 4);
    const int blocks = 256 * 2, iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        mfma_loop<<<blocks, 256>>>(d, iters, 0.5f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int k = 0; k < 20; ++k) mfma_loop<<<blocks, 256>>>(d, iters, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = 20.0 * blocks * 4 * (double)iters * 8 * 2048.0;
        printf("rep %d: %.3f ms per launch, %.1f TFLOP/s sustained over %.0f ms\n", rep, ms / 20, flops / (ms * 1e-3) / 1e12, ms);
    }
    return 0;
}
