// Shared host/device helpers for librtk_hip.so (gfx950 only; wavefront = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "rtk_pointnet2.h"
#include "rtk_train.h"

// ---- batch-statistic accumulators (rtk_train.h) ------------------------------------------------------------------------------
// A statistics buffer is (slots, groups, C, 2) float64 words, zero-initialised, `gc2` = groups * C * 2 words per slot; slots =
// RTK_STAT_SLOTS (tag 0) or RTK_STAT_SLOTS_ORDERED (tag 1).
// Two ways of adding to it, chosen by the caller per buffer through BIT 0 OF THE POINTER it hands over (buffers of
// float64 are 8-byte aligned; producers and consumers of one buffer must be given the same tag):
//
// tag 0 -- float64 atomics into one of RTK_STAT_ATOMIC_REPLICAS replicas (slot = replica, picked by the producer's sample index: the
//   epilogues of 64..1000 workgroups do not all queue on one address).  Every addition rounds, so the sum depends on the order of
//   arrival in its last bits: a step's gradients differ from run to run at 1e-6 of their largest element.
//
// tag 1 -- ORDER-INDEPENDENT: a value is the sum of its addends in 90-bit fixed point (unit 2^-U), held as three 30-bit limbs that
//   are each kept in a float64 of their own AS INTEGERS: an addend's limbs (integers below 2^30, with the addend's sign) are added
//   with float64 atomics, and as long as a limb's sum stays below 2^53 -- 2^23 addends -- every one of those additions is EXACT, so
//   their order cannot matter.  Slots: limb l of replica r = slot 3 r + l (RTK_STAT_REPLICAS replicas), the last slot counts addends
//   that were not finite or out of range -- the sum then reads as NaN.  What an addend loses is below one unit:
//     KIND 0 (forward sums: sum w z, sum w z^2): unit 2^-36 (1.5e-11), addends below 2^53 (9e15);
//     KIND 1 (backward sums: sum dy, sum dy xhat): unit 2^-66 (1.4e-20), addends below 2^23 (8.4e6).
//   Cost at B = 64: two or three atomics per addend instead of one, on five replicas instead of eight, sixteen words per value to
//   read instead of eight: +0.4 ... 1 us per producer and per consumer launch.
#define RTK_STAT_REPLICAS ((RTK_STAT_SLOTS_ORDERED - 1) / 3)
#define RTK_STAT_FLAGS (RTK_STAT_SLOTS_ORDERED - 1)
#define RTK_STAT_ATOMIC_REPLICAS RTK_STAT_SLOTS
#define RTK_STAT_FORWARD 0
#define RTK_STAT_BACKWARD 1
static_assert(RTK_STAT_REPLICAS >= 1, "RTK_STAT_SLOTS_ORDERED");
__device__ __forceinline__ bool rtk_stat_tag(const double *sums) { return (reinterpret_cast<uintptr_t>(sums) & 1u) != 0; }
__device__ __forceinline__ double *rtk_stat_base(const double *sums) {
    return reinterpret_cast<double *>(reinterpret_cast<uintptr_t>(sums) & ~(uintptr_t)1);
}
template <int KIND>
__device__ __forceinline__ void rtk_stat_add(double *tagged, size_t gc2, unsigned key, size_t idx, double x) {
    double *sums = rtk_stat_base(tagged);
    if (!rtk_stat_tag(tagged)) {
        atomicAdd(sums + (size_t)(key % RTK_STAT_ATOMIC_REPLICAS) * gc2 + idx, x);
        return;
    }
    // m = |x| / (unit 2^60): limb 2 = floor(m), limb 1 = the next 30 bits, limb 0 the 30 after those -- scalings by powers of two,
    // floor and differences of neighbours: all exact in float64
    const double m = fabs(x) * (KIND == RTK_STAT_FORWARD ? 0x1p-24 : 0x1p6);
    if (!(m < 0x1p29)) {                    // NaN, infinity or too large for the limbs
        atomicAdd(sums + (size_t)RTK_STAT_FLAGS * gc2 + idx, 1.0);
        return;
    }
    const double l2 = floor(m), r1 = (m - l2) * 0x1p30, l1 = floor(r1), l0 = floor((r1 - l1) * 0x1p30);
    const double q[3] = {l0, l1, l2};
    double *dst = sums + (size_t)(3 * (key % RTK_STAT_REPLICAS)) * gc2 + idx;
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        if (q[l] != 0.0) atomicAdd(dst + (size_t)l * gc2, x < 0 ? -q[l] : q[l]);      // (result unused: no return trip)
    }
}
// a statistic's two values (idx, idx + 1: sum and sum of squares / sum d and sum d xhat) with ONE test of the tag, so that all the
// loads of both travel together (two calls of a one-value read would be two round trips in a row)
template <int KIND>
__device__ __forceinline__ void rtk_stat_read2(const double *tagged, size_t gc2, size_t idx, double &v0, double &v1) {
    const double *sums = rtk_stat_base(tagged);
    if (!rtk_stat_tag(tagged)) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int s = 0; s < RTK_STAT_ATOMIC_REPLICAS; ++s) { a += sums[(size_t)s * gc2 + idx]; b += sums[(size_t)s * gc2 + idx + 1]; }
        v0 = a; v1 = b;
        return;
    }
    const double f0 = sums[(size_t)RTK_STAT_FLAGS * gc2 + idx], f1 = sums[(size_t)RTK_STAT_FLAGS * gc2 + idx + 1];
    double limb[2][3];
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int r = 0; r < RTK_STAT_REPLICAS; ++r) {      // integers: exact
            a += sums[(size_t)(3 * r + l) * gc2 + idx];
            b += sums[(size_t)(3 * r + l) * gc2 + idx + 1];
        }
        limb[0][l] = a; limb[1][l] = b;
    }
    const double unit = KIND == RTK_STAT_FORWARD ? 0x1p-36 : 0x1p-66, nan = __longlong_as_double(0x7ff8000000000000ll);
    v0 = f0 != 0.0 ? nan : fma(fma(limb[0][2], 0x1p30, limb[0][1]), 0x1p30, limb[0][0]) * unit;
    v1 = f1 != 0.0 ? nan : fma(fma(limb[1][2], 0x1p30, limb[1][1]), 0x1p30, limb[1][0]) * unit;
}
template <int KIND>
__device__ __forceinline__ double rtk_stat_read(const double *tagged, size_t gc2, size_t idx) {
    double v0, v1;
    rtk_stat_read2<KIND>(tagged, gc2, idx & ~(size_t)1, v0, v1);
    return (idx & 1) ? v1 : v0;
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
    double s, ss;
    rtk_stat_read2<RTK_STAT_FORWARD>(F.sums, GC * 2, o, s, ss);
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
        double s, ss;
        rtk_stat_read2<RTK_STAT_FORWARD>(F.sums, GC * 2, o2, s, ss);
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

