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

// ---- BatchNorm finalisation inside a consumer kernel (rtk_bn_fin_t, rtk_train.h) -------------------------------------------------
// Every workgroup derives the constants of its own (group, channel) from the
// replica sums; bn_fin_publish -- called by ONE thread per channel in the whole grid -- also stores all groups' constants into par
// and updates the running statistics (group 0, then group 1, ...: the reference calls the module once per frame).
// element count of group g: `count` for every group, or -- padded batches of clouds of different sizes (n_valid) -- the device
// array group_counts[groups] (rtk_train_point_weights): the step stays one captured graph whatever the clouds' sizes
__device__ __forceinline__ double rtk_group_count(double count, const double *group_counts, int g) { return group_counts ? group_counts[g] : count; }

__device__ __forceinline__ void bn_fin_constants(const rtk_bn_fin_t &F, int channels, int groups, int g, int c, float &mean, float &rstd,
                                                 float &sc, float &sh) {
    const size_t GC = (size_t)groups * channels, o = ((size_t)g * channels + c) * 2;
    const double s = rtk_stat_read(F.sums, GC * 2, o), ss = rtk_stat_read(F.sums, GC * 2, o + 1);
    const double cnt = rtk_group_count(F.count, F.group_counts, g);
    const double m = s / cnt;
    double var = ss / cnt - m * m;
    var = var > 0.0 ? var : 0.0;
    rstd = (float)(1.0 / sqrt(var + (double)F.eps));
    sc = F.gamma[c] * rstd;
    mean = (float)m;
    sh = F.beta[c] - mean * sc;
}

__device__ __forceinline__ void bn_fin_publish(const rtk_bn_fin_t &F, int channels, int groups, int c, float *par) {
    const size_t GC = (size_t)groups * channels;
    if (c == 0 && F.num_batches_tracked) *F.num_batches_tracked += groups;
    float rm = F.running_mean ? F.running_mean[c] : 0.f, rv = F.running_var ? F.running_var[c] : 0.f;
    for (int g = 0; g < groups; ++g) {
        const size_t o2 = ((size_t)g * channels + c) * 2;
        const double s = rtk_stat_read(F.sums, GC * 2, o2), ss = rtk_stat_read(F.sums, GC * 2, o2 + 1);
        const double cnt = rtk_group_count(F.count, F.group_counts, g);
        const double mean = s / cnt;
        double var = ss / cnt - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)F.eps));
        const float sc = F.gamma[c] * rstd;
        const size_t o = (size_t)g * channels + c;
        par[o] = (float)mean;
        par[GC + o] = rstd;
        par[2 * GC + o] = sc;
        par[3 * GC + o] = F.beta[c] - (float)mean * sc;
        rm = (1.f - F.momentum) * rm + F.momentum * (float)mean;
        rv = (1.f - F.momentum) * rv + F.momentum * (float)(var * (cnt / (cnt - 1.0)));
    }
    if (F.running_mean) F.running_mean[c] = rm;
    if (F.running_var) F.running_var[c] = rv;
}

