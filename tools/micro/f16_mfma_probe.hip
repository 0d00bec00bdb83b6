// f16_mfma_probe.hip -- what v_mfma_f32_32x32x16_f16 / v_mfma_f32_16x16x32_f16 and v_cvt_pk_f16_f32 do with fp16 SUBNORMALS on
// this hardware (the two-piece fp16 split of split_f16.h leans on the low piece keeping its absolute precision below 2^-14).
//   hipcc -O2 --offload-arch=gfx950 tools/micro/f16_mfma_probe.hip -o tools/micro/f16_mfma_probe.bin && tools/micro/f16_mfma_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void probe(float *out, float a_val, float b_val) {
    const int lane = threadIdx.x;
    h8 a, b;
    for (int t = 0; t < 8; ++t) { a[t] = (_Float16)0.f; b[t] = (_Float16)0.f; }
    // A[i][k]: lane (hh, i) holds k = 8 hh + t;  B[k][j]: lane (hh, j).  One non-zero product: A[0][0] * B[0][0]
    if (lane == 0) { a[0] = (_Float16)a_val; b[0] = (_Float16)b_val; }
    f16v c;
    for (int e = 0; e < 16; ++e) c[e] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    f4 c4 = {0.f, 0.f, 0.f, 0.f};
    c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c4, 0, 0, 0);
    unsigned pk;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(a_val), "v"(b_val));
    if (lane == 0) { out[0] = c[0]; out[1] = c4[0]; out[2] = __uint_as_float(pk); out[3] = (float)a[0]; }
}

int main() {
    float *d, h[4];
    hipMalloc(&d, 16);
    const float cases[][2] = {{ldexpf(1.f, -20), 1024.f}, {ldexpf(1.f, -24), 4096.f}, {ldexpf(1.5f, -16), ldexpf(1.25f, -15)},
                              {ldexpf(1.f, -14), 1.f}, {1000.f, ldexpf(1.f, -22)}};
    int kept = 0;
    for (auto &c : cases) {
        probe<<<1, 64>>>(d, c[0], c[1]);
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        unsigned pk; __builtin_memcpy(&pk, &h[2], 4);
        const double exact = (double)c[0] * (double)c[1];
        printf("a = %.6e  b = %.6e  exact a.b = %.6e | mfma 32x32x16_f16: %.6e  16x16x32_f16: %.6e | cvt_pk_f16_f32 -> 0x%04x 0x%04x | (float)(_Float16)a = %.6e\n",
               c[0], c[1], exact, h[0], h[1], pk & 0xffff, pk >> 16, h[3]);
        kept += h[0] == (float)exact && h[1] == (float)exact;
    }
    printf(kept == 5 ? "RESULT: fp16 subnormal inputs are KEPT by the f16 MFMAs\n" : "RESULT: fp16 subnormal inputs are FLUSHED (or rounded) by the f16 MFMAs: %d of 5 exact\n", kept);
    return 0;
}
