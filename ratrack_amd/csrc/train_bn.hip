// Training-mode BatchNorm2d + ReLU (+ max-pool over nsample) on de-duplicated, row-weighted activations
// (include/rtk_train.h).  Reference: lib/pytorch_utils.py:20-32,104-123 (Conv2d -> BatchNorm2d -> ReLU) and
// lib/pointnet2_modules.py:44-47 (F.max_pool2d over the neighbourhood axis).
//
// All kernels are HBM-bound streaming passes over z (samples, C, rows, ns): one workgroup per (channel, sample)
// plane (rows*ns contiguous floats), 16-byte loads, per-thread float64 partial sums, one float64 atomic pair per
// workgroup.  In the pooled variants the ns/4 lanes that hold one row's float4 chunks are adjacent, so the max /
// arg-max over the neighbourhood is a 0..3-step xor-shuffle inside the wave -- no second pass, no index tensor.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "rtk_common.h"
#include "rtk_train.h"

namespace {

constexpr int BN_T = 256;

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// block-wide sum of two doubles; result valid in thread 0
__device__ __forceinline__ void block_sum2(double &a, double &b) {
    __shared__ double s_red[2][BN_T / 64];
    a = wave_sum_f64(a);
    b = wave_sum_f64(b);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_red[0][w] = a; s_red[1][w] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = s_red[0][0]; b = s_red[1][0];
#pragma unroll
        for (int i = 1; i < BN_T / 64; ++i) { a += s_red[0][i]; b += s_red[1][i]; }
    }
}

struct Plane {
    size_t base;   // element offset of the (b, c) plane
    int g;         // statistics group of sample b
};

__device__ __forceinline__ Plane plane_of(int samples, int channels, int rows, int ns, int groups) {
    const int c = blockIdx.x, b = blockIdx.y;
    Plane p;
    p.base = ((size_t)b * channels + c) * (size_t)rows * ns;
    p.g = b / (samples / groups);
    return p;
}

// ---- forward: weighted sums --------------------------------------------------------------------------------------
__global__ __launch_bounds__(BN_T) void bn_stats_kernel(int samples, int channels, int rows, int lg_ns, int groups,
                                                        const float *__restrict__ z, const float *__restrict__ rw,
                                                        double *__restrict__ sums) {
    const Plane p = plane_of(samples, channels, rows, 1 << lg_ns, groups);
    const int E = rows << lg_ns;
    const float *zp = z + p.base;
    const float *w = rw ? rw + (size_t)blockIdx.y * rows : nullptr;
    double s = 0.0, ss = 0.0;
    if ((E & 3) == 0) {
        for (int e4 = threadIdx.x; e4 < (E >> 2); e4 += BN_T) {
            const float4 v = *reinterpret_cast<const float4 *>(zp + 4 * e4);
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const double wi = w ? (double)w[(4 * e4 + i) >> lg_ns] : 1.0;
                const double x = vv[i];
                s += wi * x;
                ss += wi * x * x;
            }
        }
    } else {
        for (int e = threadIdx.x; e < E; e += BN_T) {
            const double wi = w ? (double)w[e >> lg_ns] : 1.0;
            const double x = zp[e];
            s += wi * x;
            ss += wi * x * x;
        }
    }
    block_sum2(s, ss);
    if (threadIdx.x == 0) {
        const size_t o2 = ((size_t)p.g * channels + blockIdx.x) * 2;
        rtk_stat_add<RTK_STAT_FORWARD>(sums, (size_t)groups * channels * 2, blockIdx.y, o2, s);
        rtk_stat_add<RTK_STAT_FORWARD>(sums, (size_t)groups * channels * 2, blockIdx.y, o2 + 1, ss);
    }
}

// First layer of a set-abstraction SharedMLP in training mode, straight from the per-point projection:
//   z1[b][c][row][k] = proj[b][c][idx[b][row][k]] + Wx[c] . dxyz[b][:, row, k]
// (conv([d_xyz || feats[idx]]) = Wx.d_xyz + (Wf.feats)[idx], lib/pointnet2_utils.py:269-292 + lib/pytorch_utils.py:20-32)
// written once, with the layer's weighted batch sums accumulated on the way -- instead of a gather kernel, a 3-channel
// convolution, an addition and a statistics pass.  One workgroup per (channel, sample) plane; the plane's projection row
// sits in LDS, so the gather never leaves the CU.
template <int CPB>      // channels per workgroup: idx and d_xyz are read once per CPB output planes
__global__ __launch_bounds__(BN_T) void sa_first_layer_kernel(int samples, int channels, int rows, int lg_ns, int groups, int n_src,
                                                              const float *__restrict__ proj, const int *__restrict__ idx,
                                                              const float *__restrict__ dxyz, const float *__restrict__ wx, int wx_pitch,
                                                              const float *__restrict__ rw, float *__restrict__ z,
                                                              double *__restrict__ sums) {
    extern __shared__ float s_proj[];                              // [CPB][n_src]
    const int c0 = blockIdx.x * CPB, b = blockIdx.y;
    const int grp = b / (samples / groups);
    const int E = rows << lg_ns;                                   // multiple of 4 (ns >= 4)
    for (int i = threadIdx.x; i < CPB * n_src; i += BN_T) s_proj[i] = proj[((size_t)b * channels + c0) * n_src + i];
    __syncthreads();
    float w0[CPB], w1[CPB], w2[CPB];
#pragma unroll
    for (int q = 0; q < CPB; ++q) { const float *w = wx + (size_t)(c0 + q) * wx_pitch; w0[q] = w[0]; w1[q] = w[1]; w2[q] = w[2]; }
    const int *ib = idx + (size_t)b * E;
    const float *dx = dxyz + (size_t)b * 3 * E, *dy = dx + E, *dz = dy + E;
    const float *w = rw ? rw + (size_t)b * rows : nullptr;
    float *zp = z + ((size_t)b * channels + c0) * E;
    double s[CPB], ss[CPB];
#pragma unroll
    for (int q = 0; q < CPB; ++q) { s[q] = 0.0; ss[q] = 0.0; }
    for (int e4 = threadIdx.x; e4 < (E >> 2); e4 += BN_T) {
        const int4 t = *reinterpret_cast<const int4 *>(ib + 4 * e4);
        const float4 x = *reinterpret_cast<const float4 *>(dx + 4 * e4);
        const float4 y = *reinterpret_cast<const float4 *>(dy + 4 * e4);
        const float4 u = *reinterpret_cast<const float4 *>(dz + 4 * e4);
        const double wi = w ? (double)w[(4 * e4) >> lg_ns] : 1.0;     // ns >= 4: the four positions share a row
#pragma unroll
        for (int q = 0; q < CPB; ++q) {
            const float *pr = s_proj + q * n_src;
            float4 v;
            v.x = pr[t.x] + __fmaf_rn(w2[q], u.x, __fmaf_rn(w1[q], y.x, w0[q] * x.x));
            v.y = pr[t.y] + __fmaf_rn(w2[q], u.y, __fmaf_rn(w1[q], y.y, w0[q] * x.y));
            v.z = pr[t.z] + __fmaf_rn(w2[q], u.z, __fmaf_rn(w1[q], y.z, w0[q] * x.z));
            v.w = pr[t.w] + __fmaf_rn(w2[q], u.w, __fmaf_rn(w1[q], y.w, w0[q] * x.w));
            *reinterpret_cast<float4 *>(zp + (size_t)q * E + 4 * e4) = v;
            const double a = v.x, bb = v.y, cc = v.z, d = v.w;
            s[q] += wi * ((a + bb) + (cc + d));
            ss[q] += wi * ((a * a + bb * bb) + (cc * cc + d * d));
        }
    }
#pragma unroll
    for (int q = 0; q < CPB; ++q) {
        double a = s[q], c = ss[q];
        block_sum2(a, c);
        if (threadIdx.x == 0) {
            const size_t o2 = ((size_t)grp * channels + c0 + q) * 2;
            rtk_stat_add<RTK_STAT_FORWARD>(sums, (size_t)groups * channels * 2, b, o2, a);
            rtk_stat_add<RTK_STAT_FORWARD>(sums, (size_t)groups * channels * 2, b, o2 + 1, c);
        }
        __syncthreads();      // block_sum2's LDS scratch is reused by the next channel
    }
}

// ---- forward: normalise + ReLU --------------------------------------------------------------------------------------
// (scale, shift) of this workgroup's (group, channel): read from par, or -- fin.sums given -- finalised here from the batch sums, in
// which case the sample-0 workgroup of every channel also publishes par and the running statistics (bn_fin_publish)
__device__ __forceinline__ void fwd_constants(const rtk_bn_fin_t &fin, float *par, int channels, int groups, int g, float &sc, float &sh) {
    const size_t GC = (size_t)groups * channels, o = (size_t)g * channels + blockIdx.x;
    if (!fin.sums) {
        sc = par[2 * GC + o];
        sh = par[3 * GC + o];
        return;
    }
    __shared__ float s_c[2];
    if (threadIdx.x == 0) {
        float mean, rstd;
        bn_fin_constants(fin, channels, groups, g, blockIdx.x, mean, rstd, s_c[0], s_c[1]);
        if (blockIdx.y == 0) bn_fin_publish(fin, channels, groups, blockIdx.x, par);
    }
    __syncthreads();
    sc = s_c[0];
    sh = s_c[1];
}

__global__ __launch_bounds__(BN_T) void bn_relu_fwd_kernel(int samples, int channels, int rows, int ns, int groups,
                                                           const float *__restrict__ z, float *par, const rtk_bn_fin_t fin,
                                                           float *__restrict__ y) {
    const Plane p = plane_of(samples, channels, rows, ns, groups);
    float sc, sh;
    fwd_constants(fin, par, channels, groups, p.g, sc, sh);
    const int E = rows * ns;
    const float *zp = z + p.base;
    float *yp = y + p.base;
    if ((E & 3) == 0) {
        for (int e4 = threadIdx.x; e4 < (E >> 2); e4 += BN_T) {
            float4 v = *reinterpret_cast<const float4 *>(zp + 4 * e4);
            v.x = fmaxf(__fmaf_rn(v.x, sc, sh), 0.f);
            v.y = fmaxf(__fmaf_rn(v.y, sc, sh), 0.f);
            v.z = fmaxf(__fmaf_rn(v.z, sc, sh), 0.f);
            v.w = fmaxf(__fmaf_rn(v.w, sc, sh), 0.f);
            *reinterpret_cast<float4 *>(yp + 4 * e4) = v;
        }
    } else {
        for (int e = threadIdx.x; e < E; e += BN_T) yp[e] = fmaxf(__fmaf_rn(zp[e], sc, sh), 0.f);
    }
}

// One row = G = ns/4 adjacent lanes, each holding a float4 chunk.  Returns (in every lane of the group) the row's maximum
// of y = relu(z sc + sh), and the z value and element index (within the row) of its FIRST arg-max.
template <int G>
__device__ __forceinline__ void row_argmax(const float4 v, float sc, float sh, int sub, float &ymax, float &zarg, int &karg) {
    const float zz[4] = {v.x, v.y, v.z, v.w};
    ymax = -1.f;
    zarg = 0.f;
    karg = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float yi = fmaxf(__fmaf_rn(zz[i], sc, sh), 0.f);
        if (yi > ymax) { ymax = yi; zarg = zz[i]; karg = 4 * sub + i; }
    }
#pragma unroll
    for (int o = 1; o < G; o <<= 1) {
        const float y2 = __shfl_xor(ymax, o, 64), z2 = __shfl_xor(zarg, o, 64);
        const int k2 = __shfl_xor(karg, o, 64);
        const bool take = y2 > ymax || (y2 == ymax && k2 < karg);
        ymax = take ? y2 : ymax;
        zarg = take ? z2 : zarg;
        karg = take ? k2 : karg;
    }
}

template <int G>
__global__ __launch_bounds__(BN_T) void bn_relu_pool_fwd_kernel(int samples, int channels, int rows, int groups,
                                                                const float *__restrict__ z, float *par, const rtk_bn_fin_t fin,
                                                                float *__restrict__ y, float *__restrict__ zarg_out,
                                                                unsigned char *__restrict__ karg_out) {
    constexpr int NS = 4 * G;
    const Plane p = plane_of(samples, channels, rows, NS, groups);
    float sc, sh;
    fwd_constants(fin, par, channels, groups, p.g, sc, sh);
    const float *zp = z + p.base;
    const size_t rbase = ((size_t)blockIdx.y * channels + blockIdx.x) * rows;
    float *yp = y + rbase;
    const int E4 = rows * G;
    for (int base = 0; base < E4; base += BN_T) {
        const int e4 = base + threadIdx.x;
        const bool ok = e4 < E4;
        const float4 v = ok ? *reinterpret_cast<const float4 *>(zp + 4 * e4) : make_float4(0.f, 0.f, 0.f, 0.f);
        float ymax, zarg;
        int karg;
        row_argmax<G>(v, sc, sh, e4 % G, ymax, zarg, karg);
        if (ok && (e4 % G) == 0) {
            yp[e4 / G] = ymax;
            if (zarg_out) {      // for the backward: the element the row's gradient goes to (none while the whole row is clipped: 255)
                zarg_out[rbase + e4 / G] = zarg;
                karg_out[rbase + e4 / G] = ymax > 0.f ? (unsigned char)karg : (unsigned char)255;
            }
        }
    }
}

// Backward statistics of a pooled BatchNorm + ReLU layer from the saved arg-max (zarg, karg): only one element per row carries a
// gradient, so sums2 = (sum d, sum d xhat(zarg)) over the rows -- no pass over z.  One wave per (channel, sample) plane of `rows`.
__global__ __launch_bounds__(BN_T) void pool_bwd_stats_arg_kernel(int samples, int channels, int rows, int groups,
                                                                  const float *__restrict__ dy, const float *__restrict__ zarg,
                                                                  const unsigned char *__restrict__ karg, const float *__restrict__ par,
                                                                  double *__restrict__ sums2) {
    const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
    if (c >= channels) return;
    const int g = b / (samples / groups);
    const size_t GC = (size_t)groups * channels, o = (size_t)g * channels + c;
    const float mean = par[o], rstd = par[GC + o];
    const size_t base = ((size_t)b * channels + c) * rows;
    double s = 0.0, sx = 0.0;
    for (int r = lane; r < rows; r += 64) {
        const float d = karg[base + r] != 255 ? dy[base + r] : 0.f;
        s += (double)d;
        sx += (double)d * (double)((zarg[base + r] - mean) * rstd);
    }
    s = wave_sum_f64(s);
    sx = wave_sum_f64(sx);
    if (lane == 0) {
        rtk_stat_add<RTK_STAT_BACKWARD>(sums2, GC * 2, b, o * 2, s);
        rtk_stat_add<RTK_STAT_BACKWARD>(sums2, GC * 2, b, o * 2 + 1, sx);
    }
}

// ---- backward, pass 1 -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BN_T) void bn_relu_bwd_stats_kernel(int samples, int channels, int rows, int ns, int groups,
                                                                 const float *__restrict__ z, const float *__restrict__ dy,
                                                                 const float *__restrict__ par, double *__restrict__ sums2) {
    const Plane p = plane_of(samples, channels, rows, ns, groups);
    const size_t GC = (size_t)groups * channels, o = (size_t)p.g * channels + blockIdx.x;
    const float mean = par[o], rstd = par[GC + o], sc = par[2 * GC + o], sh = par[3 * GC + o];
    const int E = rows * ns;
    const float *zp = z + p.base, *dp = dy + p.base;
    double s = 0.0, sx = 0.0;
    auto acc = [&](float zi, float di) {
        const bool on = __fmaf_rn(zi, sc, sh) > 0.f;
        const float d = on ? di : 0.f;
        s += (double)d;
        sx += (double)d * (double)((zi - mean) * rstd);
    };
    if ((E & 3) == 0) {
        for (int e4 = threadIdx.x; e4 < (E >> 2); e4 += BN_T) {
            const float4 v = *reinterpret_cast<const float4 *>(zp + 4 * e4);
            const float4 d = *reinterpret_cast<const float4 *>(dp + 4 * e4);
            acc(v.x, d.x); acc(v.y, d.y); acc(v.z, d.z); acc(v.w, d.w);
        }
    } else {
        for (int e = threadIdx.x; e < E; e += BN_T) acc(zp[e], dp[e]);
    }
    block_sum2(s, sx);
    if (threadIdx.x == 0) {
        rtk_stat_add<RTK_STAT_BACKWARD>(sums2, GC * 2, blockIdx.y, o * 2, s);
        rtk_stat_add<RTK_STAT_BACKWARD>(sums2, GC * 2, blockIdx.y, o * 2 + 1, sx);
    }
}

template <int G>
__global__ __launch_bounds__(BN_T) void bn_relu_pool_bwd_stats_kernel(int samples, int channels, int rows, int groups,
                                                                      const float *__restrict__ z, const float *__restrict__ dy,
                                                                      const float *__restrict__ par, double *__restrict__ sums2) {
    constexpr int NS = 4 * G;
    const Plane p = plane_of(samples, channels, rows, NS, groups);
    const size_t GC = (size_t)groups * channels, o = (size_t)p.g * channels + blockIdx.x;
    const float mean = par[o], rstd = par[GC + o], sc = par[2 * GC + o], sh = par[3 * GC + o];
    const float *zp = z + p.base;
    const float *dp = dy + ((size_t)blockIdx.y * channels + blockIdx.x) * rows;
    const int E4 = rows * G;
    double s = 0.0, sx = 0.0;
    for (int base = 0; base < E4; base += BN_T) {
        const int e4 = base + threadIdx.x;
        const bool ok = e4 < E4;
        const float4 v = ok ? *reinterpret_cast<const float4 *>(zp + 4 * e4) : make_float4(0.f, 0.f, 0.f, 0.f);
        float ymax, zarg;
        int karg;
        row_argmax<G>(v, sc, sh, e4 % G, ymax, zarg, karg);
        if (ok && (e4 % G) == 0 && ymax > 0.f) {
            const float d = dp[e4 / G];
            s += (double)d;
            sx += (double)d * (double)((zarg - mean) * rstd);
        }
    }
    block_sum2(s, sx);
    if (threadIdx.x == 0) {
        rtk_stat_add<RTK_STAT_BACKWARD>(sums2, GC * 2, blockIdx.y, o * 2, s);
        rtk_stat_add<RTK_STAT_BACKWARD>(sums2, GC * 2, blockIdx.y, o * 2 + 1, sx);
    }
}

// ---- small planes (per-point layers: ns = 1, a plane is a few hundred elements) ---------------------------------------------------
// One WAVE per (channel, sample) plane, four planes per workgroup: with one workgroup per plane these passes were 16 384 workgroups
// of one element per thread -- launch rounds, not bandwidth.  Same arithmetic as the kernels above; planes of E <= 1024, E % 4 == 0.
__device__ __forceinline__ float wave_bcast(float v) { return __shfl(v, 0, 64); }

__global__ __launch_bounds__(BN_T) void bn_relu_fwd_small_kernel(int samples, int channels, int E, int groups, const float *__restrict__ z,
                                                                 float *par, const rtk_bn_fin_t fin, float *__restrict__ y) {
    const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
    if (c >= channels) return;
    const int g = b / (samples / groups);
    const size_t GC = (size_t)groups * channels, o = (size_t)g * channels + c;
    float sc = 0.f, sh = 0.f;
    if (lane == 0) {
        if (fin.sums) {
            float mean, rstd;
            bn_fin_constants(fin, channels, groups, g, c, mean, rstd, sc, sh);
            if (b == 0) bn_fin_publish(fin, channels, groups, c, par);
        } else {
            sc = par[2 * GC + o];
            sh = par[3 * GC + o];
        }
    }
    sc = wave_bcast(sc);
    sh = wave_bcast(sh);
    const size_t base = ((size_t)b * channels + c) * E;
    for (int e4 = lane; e4 < (E >> 2); e4 += 64) {
        float4 v = *reinterpret_cast<const float4 *>(z + base + 4 * e4);
        v.x = fmaxf(__fmaf_rn(v.x, sc, sh), 0.f);
        v.y = fmaxf(__fmaf_rn(v.y, sc, sh), 0.f);
        v.z = fmaxf(__fmaf_rn(v.z, sc, sh), 0.f);
        v.w = fmaxf(__fmaf_rn(v.w, sc, sh), 0.f);
        *reinterpret_cast<float4 *>(y + base + 4 * e4) = v;
    }
}

__global__ __launch_bounds__(BN_T) void bn_relu_bwd_stats_small_kernel(int samples, int channels, int E, int groups, const float *__restrict__ z,
                                                                       const float *__restrict__ dy, const float *__restrict__ par,
                                                                       double *__restrict__ sums2) {
    const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
    if (c >= channels) return;
    const int g = b / (samples / groups);
    const size_t GC = (size_t)groups * channels, o = (size_t)g * channels + c;
    const float mean = par[o], rstd = par[GC + o], sc = par[2 * GC + o], sh = par[3 * GC + o];
    const size_t base = ((size_t)b * channels + c) * E;
    double s = 0.0, sx = 0.0;
    auto acc = [&](float zi, float di) {
        const bool on = __fmaf_rn(zi, sc, sh) > 0.f;
        const float d = on ? di : 0.f;
        s += (double)d;
        sx += (double)d * (double)((zi - mean) * rstd);
    };
    for (int e4 = lane; e4 < (E >> 2); e4 += 64) {
        const float4 v = *reinterpret_cast<const float4 *>(z + base + 4 * e4);
        const float4 d = *reinterpret_cast<const float4 *>(dy + base + 4 * e4);
        acc(v.x, d.x); acc(v.y, d.y); acc(v.z, d.z); acc(v.w, d.w);
    }
    s = wave_sum_f64(s);
    sx = wave_sum_f64(sx);
    if (lane == 0) {
        rtk_stat_add<RTK_STAT_BACKWARD>(sums2, GC * 2, b, o * 2, s);
        rtk_stat_add<RTK_STAT_BACKWARD>(sums2, GC * 2, b, o * 2 + 1, sx);
    }
}

__global__ __launch_bounds__(BN_T) void bn_relu_bwd_apply_small_kernel(int samples, int channels, int E, int groups, const float *__restrict__ z,
                                                                       const float *__restrict__ dy, const float *__restrict__ par,
                                                                       const float *__restrict__ rw, const double *__restrict__ sums2,
                                                                       double count0, const double *__restrict__ gcnt, float *__restrict__ dz, float *__restrict__ dgb) {
    const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
    if (c >= channels) return;
    const int g = b / (samples / groups);
    const double count = rtk_group_count(count0, gcnt, g);
    const size_t GC = (size_t)groups * channels, o = (size_t)g * channels + c;
    const float mean = par[o], rstd = par[GC + o], sc = par[2 * GC + o], sh = par[3 * GC + o];
    float c1 = 0.f, c2 = 0.f;
    if (lane == 0) {
        {
            double v0_, v1_;
            rtk_stat_read2<RTK_STAT_BACKWARD>(sums2, GC * 2, o * 2, v0_, v1_);
            c1 = (float)(v0_ / count);
            c2 = (float)(v1_ / count);
        }
        if (dgb && b == 0) {                                       // parameter gradients: sum over the groups
            double db = 0.0, dg = 0.0;
            for (int gg = 0; gg < groups; ++gg) {
                {
                    double v0_, v1_;
                    rtk_stat_read2<RTK_STAT_BACKWARD>(sums2, GC * 2, ((size_t)gg * channels + c) * 2, v0_, v1_);
                    db += v0_;
                    dg += v1_;
                }
            }
            dgb[c] = (float)dg;
            dgb[channels + c] = (float)db;
        }
    }
    c1 = wave_bcast(c1);
    c2 = wave_bcast(c2);
    const size_t base = ((size_t)b * channels + c) * E;
    const float *w = rw ? rw + (size_t)b * E : nullptr;            // ns = 1: one weight per element
    auto one = [&](float zi, float di, int e) -> float {
        const bool on = __fmaf_rn(zi, sc, sh) > 0.f;
        const float wi = w ? w[e] : 1.f;
        const float xh = (zi - mean) * rstd;
        return sc * ((on ? di : 0.f) - wi * (c1 + xh * c2));
    };
    for (int e4 = lane; e4 < (E >> 2); e4 += 64) {
        const float4 v = *reinterpret_cast<const float4 *>(z + base + 4 * e4);
        const float4 d = *reinterpret_cast<const float4 *>(dy + base + 4 * e4);
        float4 r;
        r.x = one(v.x, d.x, 4 * e4); r.y = one(v.y, d.y, 4 * e4 + 1);
        r.z = one(v.z, d.z, 4 * e4 + 2); r.w = one(v.w, d.w, 4 * e4 + 3);
        *reinterpret_cast<float4 *>(dz + base + 4 * e4) = r;
    }
}

// ---- per-point layers: BatchNorm + ReLU backward in ONE kernel -------------------------------------------------------------------
// The two passes need the batch sums between them -- a grid-wide dependency, hence two launches (statistics, apply) in general.  A
// per-point layer's channel, however, is small (samples x E <= 64 Ki elements): ONE workgroup takes a whole channel -- all samples of
// group 0, then group 1, ... --, sums it (first pass), and applies (second pass; the re-read comes out of L2).  Same arithmetic as
// bn_relu_bwd_stats_small_kernel + bn_relu_bwd_apply_small_kernel except for the order of the float64 sums.
constexpr int BF_T = 1024;
__global__ __launch_bounds__(BF_T) void bn_relu_bwd_fused_small_kernel(int samples, int channels, int E, int groups, const float *__restrict__ z,
                                                                       const float *__restrict__ dy, const float *__restrict__ par,
                                                                       const float *__restrict__ rw, double count0, const double *__restrict__ gcnt,
                                                                       float *__restrict__ dz, float *__restrict__ dgb) {
    __shared__ double s_red[2][BF_T / 64];
    __shared__ float s_c[2];
    const int c = blockIdx.x, t = threadIdx.x, per_group = samples / groups, E4 = E >> 2;
    const size_t GC = (size_t)groups * channels;
    double dg_all = 0.0, db_all = 0.0;
    // gridDim.y == groups: a workgroup per (channel, group) -- twice the workgroups for the two-frame batches, whose channels alone
    // fill only half the chip; dgamma / dbeta are then ADDED to a zeroed buffer (two commutative float adds per address: deterministic).
    // (Keeping the group's elements in registers between the two phases -- all loads issued at once, nothing read twice -- was slower:
    // 11 -> 13-15 us.)
    const int g_lo = gridDim.y > 1 ? blockIdx.y : 0, g_hi = gridDim.y > 1 ? blockIdx.y + 1 : groups;
    for (int g = g_lo; g < g_hi; ++g) {
        const size_t o = (size_t)g * channels + c;
        const float mean = par[o], rstd = par[GC + o], sc = par[2 * GC + o], sh = par[3 * GC + o];
        const int n4 = per_group * E4;                              // float4 elements of this group's part of the channel
        double s = 0.0, sx = 0.0;
        for (int i = t; i < n4; i += BF_T) {
            const int b = g * per_group + i / E4, e4 = i - (i / E4) * E4;
            const size_t base = ((size_t)b * channels + c) * E + 4 * e4;
            const float4 v = *reinterpret_cast<const float4 *>(z + base), d = *reinterpret_cast<const float4 *>(dy + base);
            const float zz[4] = {v.x, v.y, v.z, v.w}, dd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float di = __fmaf_rn(zz[k], sc, sh) > 0.f ? dd[k] : 0.f;
                s += (double)di;
                sx += (double)di * (double)((zz[k] - mean) * rstd);
            }
        }
        s = wave_sum_f64(s);
        sx = wave_sum_f64(sx);
        __syncthreads();                                            // (the previous group's constants have been read)
        if ((t & 63) == 0) { s_red[0][t >> 6] = s; s_red[1][t >> 6] = sx; }
        __syncthreads();
        if (t == 0) {
            double a = 0.0, b2 = 0.0;
#pragma unroll
            for (int i = 0; i < BF_T / 64; ++i) { a += s_red[0][i]; b2 += s_red[1][i]; }
            const double count = rtk_group_count(count0, gcnt, g);
            s_c[0] = (float)(a / count);
            s_c[1] = (float)(b2 / count);
            db_all += a;
            dg_all += b2;
        }
        __syncthreads();
        const float c1 = s_c[0], c2 = s_c[1];
        for (int i = t; i < n4; i += BF_T) {
            const int b = g * per_group + i / E4, e4 = i - (i / E4) * E4;
            const size_t base = ((size_t)b * channels + c) * E + 4 * e4;
            const float4 v = *reinterpret_cast<const float4 *>(z + base), d = *reinterpret_cast<const float4 *>(dy + base);
            float4 w4 = make_float4(1.f, 1.f, 1.f, 1.f);
            if (rw) w4 = *reinterpret_cast<const float4 *>(rw + (size_t)b * E + 4 * e4);      // ns = 1: one weight per element
            auto one = [&](float zi, float di, float wi) -> float {
                const bool on = __fmaf_rn(zi, sc, sh) > 0.f;
                const float xh = (zi - mean) * rstd;
                return sc * ((on ? di : 0.f) - wi * (c1 + xh * c2));
            };
            float4 r;
            r.x = one(v.x, d.x, w4.x); r.y = one(v.y, d.y, w4.y); r.z = one(v.z, d.z, w4.z); r.w = one(v.w, d.w, w4.w);
            *reinterpret_cast<float4 *>(dz + base) = r;
        }
    }
    if (t == 0 && dgb) {                                            // parameter gradients: sums over the groups
        if (gridDim.y > 1) {
            atomicAdd(dgb + c, (float)dg_all);
            atomicAdd(dgb + channels + c, (float)db_all);
        } else {
            dgb[c] = (float)dg_all;
            dgb[channels + c] = (float)db_all;
        }
    }
}

// ---- backward, pass 2 -------------------------------------------------------------------------------------------------
struct BwdCoef {
    float mean, rstd, sc, sh, c1, c2;
};

__device__ __forceinline__ BwdCoef bwd_coef(int channels, int groups, int g, const float *par, const double *sums2, double count,
                                            float *dgb) {
    const int c = blockIdx.x;
    const size_t GC = (size_t)groups * channels, o = (size_t)g * channels + c;
    BwdCoef k;
    k.mean = par[o]; k.rstd = par[GC + o]; k.sc = par[2 * GC + o]; k.sh = par[3 * GC + o];
    {
        double v0_, v1_;
        rtk_stat_read2<RTK_STAT_BACKWARD>(sums2, GC * 2, o * 2, v0_, v1_);
        k.c1 = (float)(v0_ / count);
        k.c2 = (float)(v1_ / count);
    }
    if (dgb && blockIdx.y == 0 && threadIdx.x == 0) {     // parameter gradients: sum over the groups
        double db = 0.0, dg = 0.0;
        for (int gg = 0; gg < groups; ++gg) {
            {
                double v0_, v1_;
                rtk_stat_read2<RTK_STAT_BACKWARD>(sums2, GC * 2, ((size_t)gg * channels + c) * 2, v0_, v1_);
                db += v0_;
                dg += v1_;
            }
        }
        dgb[c] = (float)dg;
        dgb[channels + c] = (float)db;
    }
    return k;
}

__global__ __launch_bounds__(BN_T) void bn_relu_bwd_apply_kernel(int samples, int channels, int rows, int lg_ns, int groups,
                                                                 const float *__restrict__ z, const float *__restrict__ dy,
                                                                 const float *__restrict__ par, const float *__restrict__ rw,
                                                                 const double *__restrict__ sums2, double count, const double *__restrict__ gcnt,
                                                                 float *__restrict__ dz, float *__restrict__ dgb) {
    const Plane p = plane_of(samples, channels, rows, 1 << lg_ns, groups);
    const BwdCoef k = bwd_coef(channels, groups, p.g, par, sums2, rtk_group_count(count, gcnt, p.g), dgb);
    const int E = rows << lg_ns;
    const float *zp = z + p.base, *dp = dy + p.base;
    float *op = dz + p.base;
    const float *w = rw ? rw + (size_t)blockIdx.y * rows : nullptr;
    auto one = [&](float zi, float di, int e) -> float {
        const bool on = __fmaf_rn(zi, k.sc, k.sh) > 0.f;
        const float wi = w ? w[e >> lg_ns] : 1.f;
        const float xh = (zi - k.mean) * k.rstd;
        return k.sc * ((on ? di : 0.f) - wi * (k.c1 + xh * k.c2));
    };
    if ((E & 3) == 0) {
        for (int e4 = threadIdx.x; e4 < (E >> 2); e4 += BN_T) {
            const float4 v = *reinterpret_cast<const float4 *>(zp + 4 * e4);
            const float4 d = *reinterpret_cast<const float4 *>(dp + 4 * e4);
            float4 r;
            r.x = one(v.x, d.x, 4 * e4); r.y = one(v.y, d.y, 4 * e4 + 1);
            r.z = one(v.z, d.z, 4 * e4 + 2); r.w = one(v.w, d.w, 4 * e4 + 3);
            *reinterpret_cast<float4 *>(op + 4 * e4) = r;
        }
    } else {
        for (int e = threadIdx.x; e < E; e += BN_T) op[e] = one(zp[e], dp[e], e);
    }
}

template <int G>
__global__ __launch_bounds__(BN_T) void bn_relu_pool_bwd_apply_kernel(int samples, int channels, int rows, int groups,
                                                                      const float *__restrict__ z, const float *__restrict__ dy,
                                                                      const float *__restrict__ par, const float *__restrict__ rw,
                                                                      const double *__restrict__ sums2, double count, const double *__restrict__ gcnt,
                                                                      float *__restrict__ dz, float *__restrict__ dgb) {
    constexpr int NS = 4 * G;
    const Plane p = plane_of(samples, channels, rows, NS, groups);
    const BwdCoef k = bwd_coef(channels, groups, p.g, par, sums2, rtk_group_count(count, gcnt, p.g), dgb);
    const float *zp = z + p.base;
    const float *dp = dy + ((size_t)blockIdx.y * channels + blockIdx.x) * rows;
    float *op = dz + p.base;
    const float *w = rw ? rw + (size_t)blockIdx.y * rows : nullptr;
    const int E4 = rows * G;
    for (int base = 0; base < E4; base += BN_T) {
        const int e4 = base + threadIdx.x;
        const bool ok = e4 < E4;
        const float4 v = ok ? *reinterpret_cast<const float4 *>(zp + 4 * e4) : make_float4(0.f, 0.f, 0.f, 0.f);
        float ymax, zarg;
        int karg;
        const int sub = e4 % G;
        row_argmax<G>(v, k.sc, k.sh, sub, ymax, zarg, karg);
        if (ok) {
            const int row = e4 / G;
            const float d = ymax > 0.f ? dp[row] : 0.f;
            const float wi = w ? w[row] : 1.f;
            const float zz[4] = {v.x, v.y, v.z, v.w};
            float r[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float xh = (zz[i] - k.mean) * k.rstd;
                r[i] = k.sc * ((karg == 4 * sub + i ? d : 0.f) - wi * (k.c1 + xh * k.c2));
            }
            *reinterpret_cast<float4 *>(op + 4 * e4) = make_float4(r[0], r[1], r[2], r[3]);
        }
    }
}

int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

#define RTK_BN_COMMON_CHECKS(name)                                                                                    \
    RTK_REQUIRE(samples > 0 && channels > 0 && rows > 0 && ns > 0, name ": empty tensor");                          \
    RTK_REQUIRE(groups > 0 && samples % groups == 0, name ": samples (%d) not divisible by groups (%d)", samples, groups); \
    RTK_REQUIRE(ilog2_exact(ns) >= 0, name ": ns (%d) must be a power of two", ns);                                 \
    RTK_REQUIRE(samples <= 65535, name ": too many samples (%d)", samples)

// per-point layers: one wave per plane (bn_relu_*_small_kernel)
#define BN_SMALL_PLANE (ns == 1 && rows <= 1024 && (rows & 3) == 0)

#define RTK_BN_POOL_DISPATCH(KERNEL, ...)                                                   \
    switch (ns) {                                                                           \
    case 4: KERNEL<1><<<grid, BN_T, 0, s>>>(__VA_ARGS__); break;                            \
    case 8: KERNEL<2><<<grid, BN_T, 0, s>>>(__VA_ARGS__); break;                            \
    case 16: KERNEL<4><<<grid, BN_T, 0, s>>>(__VA_ARGS__); break;                           \
    case 32: KERNEL<8><<<grid, BN_T, 0, s>>>(__VA_ARGS__); break;                           \
    default: rtk_set_error("pooled BatchNorm: ns (%d) must be 4, 8, 16 or 32", ns); return RTK_ERR_INVALID; \
    }

}  // namespace

extern "C" int rtk_bn_train_stats(int samples, int channels, int rows, int ns, int groups, const float *z, const float *row_weight,
                                  double *sums, rtk_stream_t stream) {
    RTK_BN_COMMON_CHECKS("rtk_bn_train_stats");
    hipStream_t s = (hipStream_t)stream;
    bn_stats_kernel<<<dim3(channels, samples), BN_T, 0, s>>>(samples, channels, rows, ilog2_exact(ns), groups, z, row_weight, sums);
    RTK_CHECK_LAUNCH("rtk_bn_train_stats");
    return RTK_OK;
}

static int bn_relu_fwd_launch(int samples, int channels, int rows, int ns, int groups, const float *z, float *par,
                              const rtk_bn_fin_t &fin, int pool, float *y, rtk_stream_t stream, float *zarg_out = nullptr,
                              unsigned char *karg_out = nullptr) {
    RTK_BN_COMMON_CHECKS("rtk_bn_relu_fwd");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(channels, samples);
    if (pool) {
        RTK_BN_POOL_DISPATCH(bn_relu_pool_fwd_kernel, samples, channels, rows, groups, z, par, fin, y, zarg_out, karg_out)
    } else if (BN_SMALL_PLANE) {
        bn_relu_fwd_small_kernel<<<dim3(rtk_divup(channels, 4), samples), BN_T, 0, s>>>(samples, channels, rows, groups, z, par, fin, y);
    } else {
        bn_relu_fwd_kernel<<<grid, BN_T, 0, s>>>(samples, channels, rows, ns, groups, z, par, fin, y);
    }
    RTK_CHECK_LAUNCH("rtk_bn_relu_fwd");
    return RTK_OK;
}

extern "C" int rtk_bn_relu_fwd(int samples, int channels, int rows, int ns, int groups, const float *z, const float *par, int pool,
                               float *y, rtk_stream_t stream) {
    rtk_bn_fin_t none = {};
    return bn_relu_fwd_launch(samples, channels, rows, ns, groups, z, const_cast<float *>(par), none, pool, y, stream);
}

extern "C" int rtk_bn_relu_fwd_fin(int samples, int channels, int rows, int ns, int groups, const float *z, const rtk_bn_fin_t *fin,
                                   float *par_out, int pool, float *y, rtk_stream_t stream) {
    RTK_REQUIRE(fin && fin->sums && fin->gamma && fin->beta && fin->count > 1.0 && par_out, "rtk_bn_relu_fwd_fin: bad finalisation arguments");
    return bn_relu_fwd_launch(samples, channels, rows, ns, groups, z, par_out, *fin, pool, y, stream);
}

extern "C" int rtk_bn_relu_pool_fwd_fin_arg(int samples, int channels, int rows, int ns, int groups, const float *z, const rtk_bn_fin_t *fin,
                                            float *par_out, float *y, float *zarg_out, unsigned char *karg_out, rtk_stream_t stream) {
    RTK_REQUIRE(fin && fin->sums && fin->gamma && fin->beta && fin->count > 1.0 && par_out && zarg_out && karg_out,
                "rtk_bn_relu_pool_fwd_fin_arg: bad arguments");
    return bn_relu_fwd_launch(samples, channels, rows, ns, groups, z, par_out, *fin, 1, y, stream, zarg_out, karg_out);
}

extern "C" int rtk_pool_bwd_stats_arg(int samples, int channels, int rows, int groups, const float *dy, const float *zarg,
                                      const unsigned char *karg, const float *par, double *sums2, rtk_stream_t stream) {
    RTK_REQUIRE(samples > 0 && samples <= 65535 && channels > 0 && rows > 0 && groups > 0 && samples % groups == 0 && dy && zarg && karg && par &&
                sums2, "rtk_pool_bwd_stats_arg: bad arguments");
    pool_bwd_stats_arg_kernel<<<dim3(rtk_divup(channels, 4), samples), BN_T, 0, (hipStream_t)stream>>>(samples, channels, rows, groups, dy, zarg,
                                                                                                 karg, par, sums2);
    RTK_CHECK_LAUNCH("rtk_pool_bwd_stats_arg");
    return RTK_OK;
}

extern "C" int rtk_bn_relu_bwd_stats(int samples, int channels, int rows, int ns, int groups, const float *z, const float *dy,
                                     const float *par, int pool, double *sums2, rtk_stream_t stream) {
    RTK_BN_COMMON_CHECKS("rtk_bn_relu_bwd_stats");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(channels, samples);
    if (pool) {
        RTK_BN_POOL_DISPATCH(bn_relu_pool_bwd_stats_kernel, samples, channels, rows, groups, z, dy, par, sums2)
    } else if (BN_SMALL_PLANE) {
        bn_relu_bwd_stats_small_kernel<<<dim3(rtk_divup(channels, 4), samples), BN_T, 0, s>>>(samples, channels, rows, groups, z, dy, par, sums2);
    } else {
        bn_relu_bwd_stats_kernel<<<grid, BN_T, 0, s>>>(samples, channels, rows, ns, groups, z, dy, par, sums2);
    }
    RTK_CHECK_LAUNCH("rtk_bn_relu_bwd_stats");
    return RTK_OK;
}

extern "C" int rtk_bn_relu_bwd_apply(int samples, int channels, int rows, int ns, int groups, const float *z, const float *dy,
                                     const float *par, const float *row_weight, const double *sums2, double count, const double *group_counts, int pool,
                                     float *dz, float *dgamma_dbeta, rtk_stream_t stream) {
    RTK_BN_COMMON_CHECKS("rtk_bn_relu_bwd_apply");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(channels, samples);
    if (pool) {
        RTK_BN_POOL_DISPATCH(bn_relu_pool_bwd_apply_kernel, samples, channels, rows, groups, z, dy, par, row_weight, sums2, count, group_counts, dz,
                             dgamma_dbeta)
    } else if (BN_SMALL_PLANE) {
        bn_relu_bwd_apply_small_kernel<<<dim3(rtk_divup(channels, 4), samples), BN_T, 0, s>>>(samples, channels, rows, groups, z, dy, par, row_weight,
                                                                                           sums2, count, group_counts, dz, dgamma_dbeta);
    } else {
        bn_relu_bwd_apply_kernel<<<grid, BN_T, 0, s>>>(samples, channels, rows, ilog2_exact(ns), groups, z, dy, par, row_weight, sums2,
                                                       count, group_counts, dz, dgamma_dbeta);
    }
    RTK_CHECK_LAUNCH("rtk_bn_relu_bwd_apply");
    return RTK_OK;
}

extern "C" int rtk_bn_relu_bwd_small(int samples, int channels, int positions, int groups, const float *z, const float *dy, const float *par,
                                     const float *row_weight, double count, const double *group_counts, float *dz, float *dgamma_dbeta,
                                     int split_groups, rtk_stream_t stream) {
    RTK_REQUIRE(samples > 0 && channels > 0 && positions > 0 && (positions & 3) == 0 && groups > 0 && samples % groups == 0 && z && dy && par && dz,
                "rtk_bn_relu_bwd_small: bad arguments (positions %% 4 == 0)");
    RTK_REQUIRE((long)samples * positions <= 65536, "rtk_bn_relu_bwd_small: %d x %d elements per channel (at most 65536: use the two-pass kernels)",
                samples, positions);
    RTK_REQUIRE(((size_t)z & 15) == 0 && ((size_t)dy & 15) == 0 && ((size_t)dz & 15) == 0 && (!row_weight || ((size_t)row_weight & 15) == 0),
                "rtk_bn_relu_bwd_small: tensors must be 16-byte aligned");
    // two groups: a workgroup per (channel, group); dgamma_dbeta must then arrive ZEROED (the two groups' sums are added to it)
    const dim3 grid(channels, (groups == 2 && split_groups) ? 2 : 1);
    bn_relu_bwd_fused_small_kernel<<<grid, BF_T, 0, (hipStream_t)stream>>>(samples, channels, positions, groups, z, dy, par, row_weight, count,
                                                                          group_counts, dz, dgamma_dbeta);
    RTK_CHECK_LAUNCH("rtk_bn_relu_bwd_small");
    return RTK_OK;
}

extern "C" int rtk_sa_first_layer(int samples, int channels, int rows, int ns, int groups, int n_src, const float *proj, const int *idx,
                                  const float *dxyz, const float *wx, int wx_pitch, const float *row_weight, float *z, double *sums,
                                  rtk_stream_t stream) {
    RTK_BN_COMMON_CHECKS("rtk_sa_first_layer");
    RTK_REQUIRE(ns >= 4 && n_src > 0 && n_src <= 16384, "rtk_sa_first_layer: ns (%d) must be >= 4, n_src (%d) <= 16384", ns, n_src);
    RTK_REQUIRE(proj && idx && dxyz && wx && wx_pitch >= 3 && z && sums, "rtk_sa_first_layer: null argument");
    hipStream_t s = (hipStream_t)stream;
    if (channels % 8 == 0 && (size_t)n_src * 32 <= 64 * 1024 && (long)(channels / 8) * samples >= 1024)      // (enough workgroups left)
        sa_first_layer_kernel<8><<<dim3(channels / 8, samples), BN_T, (size_t)n_src * 32, s>>>(samples, channels, rows, ilog2_exact(ns), groups,
                                                                                            n_src, proj, idx, dxyz, wx, wx_pitch, row_weight, z, sums);
    else if (channels % 4 == 0 && (size_t)n_src * 16 <= 64 * 1024)
        sa_first_layer_kernel<4><<<dim3(channels / 4, samples), BN_T, (size_t)n_src * 16, s>>>(samples, channels, rows, ilog2_exact(ns), groups,
                                                                                            n_src, proj, idx, dxyz, wx, wx_pitch, row_weight, z, sums);
    else
        sa_first_layer_kernel<1><<<dim3(channels, samples), BN_T, (size_t)n_src * 4, s>>>(samples, channels, rows, ilog2_exact(ns), groups, n_src,
                                                                                       proj, idx, dxyz, wx, wx_pitch, row_weight, z, sums);
    RTK_CHECK_LAUNCH("rtk_sa_first_layer");
    return RTK_OK;
}

// ---- de-duplicated geometry tables of the training path (ratrack_amd/train_path.py) ----------------------------------------
// One launch each instead of a dozen framework kernels (slice copies, comparisons, where, gather, subtract, sqrt, divide).
namespace {

__global__ void train_group_geometry_kernel(int n_src_rows, int npoint, int rows, int ns, const float *__restrict__ src_xyz,
                                            const float *__restrict__ dst_xyz, const int *__restrict__ ball,
                                            int src_live_rows, const int *__restrict__ dst_nuniq, int *__restrict__ idx_out,
                                            float *__restrict__ dxyz) {
    const int b = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;      // (row, k)
    if (e >= rows * ns) return;
    const int row = e / ns, k = e % ns;
    // centroid rows >= nuniq are copies of centroid 0; the ball query skipped them (their lists are unwritten zeros): they take
    // centroid 0's list, which makes the level's rows [nuniq, rows) true copies of row 0 -- values a later level may gather
    const int brow = (dst_nuniq && row >= dst_nuniq[b]) ? 0 : row;
    int t = ball[((size_t)b * npoint + brow) * ns + k];
    // Source rows the de-duplicated level tensor does not hold (>= its row count) are copies of row 0.  Rows in [nuniq, rows) DO
    // exist -- computed copies of row 0 with statistics weight 0 -- and are used as they are: redirecting them to row 0 as well
    // piled thousands of references on one row (real radar frames padded to a common size: 142 copies), and the inverse tables of
    // the gather-form backward rank every list in O(length^2).  Same values forward; the gradient of a copy reaches the weights
    // through an identical chain.
    if (t >= src_live_rows) t = 0;
    idx_out[(size_t)b * rows * ns + e] = t;
    const float *p = src_xyz + ((size_t)b * n_src_rows + t) * 3, *c = dst_xyz + ((size_t)b * npoint + row) * 3;
    float *o = dxyz + (size_t)b * 3 * rows * ns + e;
    o[0] = __fsub_rn(p[0], c[0]);
    o[(size_t)rows * ns] = __fsub_rn(p[1], c[1]);
    o[2 * (size_t)rows * ns] = __fsub_rn(p[2], c[2]);
}

__global__ void train_interp_kernel(int rows_total, int rows, const float *__restrict__ d2, const int *__restrict__ idx,
                                    const int *__restrict__ known_nuniq, int *__restrict__ idx_out, float *__restrict__ w_out) {
    const int b = blockIdx.y, r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float *d = d2 + ((size_t)b * rows_total + r) * 3;
    const int *id = idx + ((size_t)b * rows_total + r) * 3;
    const int nu = known_nuniq ? known_nuniq[b] : 0x7fffffff;
    // lib/pointnet2_modules.py:143-146: dist_recip = 1 / (dist + 1e-8), weight = dist_recip / sum(dist_recip); dist = sqrt(d2)
    const float r0 = __fdiv_rn(1.0f, __fadd_rn(__fsqrt_rn(d[0]), 1e-8f));
    const float r1 = __fdiv_rn(1.0f, __fadd_rn(__fsqrt_rn(d[1]), 1e-8f));
    const float r2 = __fdiv_rn(1.0f, __fadd_rn(__fsqrt_rn(d[2]), 1e-8f));
    const float norm = __fadd_rn(__fadd_rn(r0, r1), r2);
    float *w = w_out + ((size_t)b * rows + r) * 3;
    int *io = idx_out + ((size_t)b * rows + r) * 3;
    w[0] = __fdiv_rn(r0, norm); w[1] = __fdiv_rn(r1, norm); w[2] = __fdiv_rn(r2, norm);
    io[0] = id[0] >= nu ? 0 : id[0]; io[1] = id[1] >= nu ? 0 : id[1]; io[2] = id[2] >= nu ? 0 : id[2];
}

// level-0 statistics weights of a padded batch: w[s][r] = r < n_valid[s]; counts[g] = sum of n_valid over group g's samples (float64,
// what the BatchNorm kernels read as the group's element count)
__global__ void train_point_weights_kernel(int samples, int rows, int groups, const int *__restrict__ n_valid, float *__restrict__ w,
                                           double *__restrict__ counts) {
    const int b = blockIdx.y, r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rows) w[(size_t)b * rows + r] = r < n_valid[b] ? 1.f : 0.f;
    if (blockIdx.x == 0 && threadIdx.x == 0 && b < groups) {
        const int per = samples / groups;
        long tot = 0;
        for (int i = 0; i < per; ++i) tot += n_valid[b * per + i];
        counts[b] = (double)tot;
    }
}

__global__ void train_row_weights_kernel(int rows, int npoint, const int *__restrict__ nuniq, float *__restrict__ w) {
    const int b = blockIdx.y, r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const int nu = nuniq[b];
    w[(size_t)b * rows + r] = (r < nu ? 1.f : 0.f) + (r == 0 ? (float)(npoint - nu) : 0.f);
}

}  // namespace

extern "C" int rtk_train_group_geometry(int samples, int n_src_rows, int npoint, int rows, int ns, const float *src_xyz,
                                        const float *dst_xyz, const int *ball_idx, int src_live_rows, const int *dst_nuniq, int *idx_out,
                                        float *dxyz, rtk_stream_t stream) {
    RTK_REQUIRE(samples > 0 && samples <= 65535 && n_src_rows > 0 && npoint > 0 && rows > 0 && rows <= npoint && ns > 0 && src_xyz &&
                dst_xyz && ball_idx && idx_out && dxyz && src_live_rows > 0 && src_live_rows <= n_src_rows,
                "rtk_train_group_geometry: bad arguments");
    train_group_geometry_kernel<<<dim3(rtk_divup((long)rows * ns, 256), samples), 256, 0, (hipStream_t)stream>>>(
        n_src_rows, npoint, rows, ns, src_xyz, dst_xyz, ball_idx, src_live_rows, dst_nuniq, idx_out, dxyz);
    RTK_CHECK_LAUNCH("rtk_train_group_geometry");
    return RTK_OK;
}

extern "C" int rtk_train_interp_weights(int samples, int rows_total, int rows, const float *dist2, const int *idx, const int *known_nuniq,
                                        int *idx_out, float *weight_out, rtk_stream_t stream) {
    RTK_REQUIRE(samples > 0 && samples <= 65535 && rows > 0 && rows <= rows_total && dist2 && idx && idx_out && weight_out,
                "rtk_train_interp_weights: bad arguments");
    train_interp_kernel<<<dim3(rtk_divup(rows, 256), samples), 256, 0, (hipStream_t)stream>>>(rows_total, rows, dist2, idx, known_nuniq,
                                                                                        idx_out, weight_out);
    RTK_CHECK_LAUNCH("rtk_train_interp_weights");
    return RTK_OK;
}

extern "C" int rtk_train_point_weights(int samples, int rows, int groups, const int *n_valid, float *weights, double *group_counts,
                                       rtk_stream_t stream) {
    RTK_REQUIRE(samples > 0 && samples <= 65535 && rows > 0 && groups > 0 && groups <= samples && samples % groups == 0 && n_valid && weights &&
                group_counts, "rtk_train_point_weights: bad arguments");
    train_point_weights_kernel<<<dim3(rtk_divup(rows, 256), samples), 256, 0, (hipStream_t)stream>>>(samples, rows, groups, n_valid, weights,
                                                                                                group_counts);
    RTK_CHECK_LAUNCH("rtk_train_point_weights");
    return RTK_OK;
}

extern "C" int rtk_train_row_weights(int samples, int rows, int npoint, const int *nuniq, float *weights, rtk_stream_t stream) {
    RTK_REQUIRE(samples > 0 && samples <= 65535 && rows > 0 && npoint >= rows && nuniq && weights, "rtk_train_row_weights: bad arguments");
    train_row_weights_kernel<<<dim3(rtk_divup(rows, 256), samples), 256, 0, (hipStream_t)stream>>>(rows, npoint, nuniq, weights);
    RTK_CHECK_LAUNCH("rtk_train_row_weights");
    return RTK_OK;
}
