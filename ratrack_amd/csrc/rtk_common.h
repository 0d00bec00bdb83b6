// Shared host/device helpers for librtk_hip.so (gfx950 only; wavefront = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "rtk_pointnet2.h"
#include "rtk_train.h"

// Batch-statistic replicas (RTK_STAT_SLOTS, rtk_train.h): `gc2` = groups * channels * 2 doubles per replica.
__device__ __forceinline__ double *rtk_stat_slot(double *sums, size_t gc2, unsigned key) { return sums + (size_t)(key % RTK_STAT_SLOTS) * gc2; }
__device__ __forceinline__ double rtk_stat_read(const double *sums, size_t gc2, size_t idx) {
    double a = 0.0;
#pragma unroll
    for (int s = 0; s < RTK_STAT_SLOTS; ++s) a += sums[(size_t)s * gc2 + idx];
    return a;
}

#define RTK_WAVE 64

void rtk_set_error(const char *fmt, ...);

#define RTK_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            rtk_set_error(__VA_ARGS__);   \
            return RTK_ERR_INVALID;       \
        }                                 \
    } while (0)

// Launch failures are reported to the caller, never exit() (the reference's launchers do
// fprintf+exit(-1), e.g. sampling_gpu.cu:39-43).
#define RTK_CHECK_LAUNCH(name)                                                     \
    do {                                                                           \
        hipError_t e_ = hipGetLastError();                                         \
        if (e_ != hipSuccess) {                                                    \
            rtk_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));   \
            return RTK_ERR_LAUNCH;                                                 \
        }                                                                          \
    } while (0)

static inline int rtk_divup(long a, long b) { return (int)((a + b - 1) / b); }

#ifdef __HIPCC__
// The one squared-distance formula of the arithmetic contract (include/rtk_pointnet2.h).
__device__ __forceinline__ float rtk_sqdist(float ax, float ay, float az, float bx, float by, float bz) {
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
}
#endif
