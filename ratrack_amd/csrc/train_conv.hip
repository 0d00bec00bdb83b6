// Training-mode 1x1 convolution fused with the BatchNorm work around it (include/rtk_train.h), for the set-abstraction
// SharedMLPs (lib/pytorch_utils.py:20-32: Conv2d(1x1, no bias) -> BatchNorm2d(batch statistics) -> ReLU).
//
//   forward  (rtk_conv_bn_fwd):  a = relu(scale_prev x + shift_prev)   [the previous layer's BatchNorm + ReLU, on load]
//                                z = W a                               [fp32 MFMA, activation-stationary]
//                                sums += (sum w z, sum w z^2)           [this layer's batch statistics, in the epilogue]
//                                optionally stores a (the weight gradient of this layer needs it)
//   backward (rtk_conv_bn_bwd):  dy = W^T dz ;  m = [scale_prev zprev + shift_prev > 0]
//        pass 0:  sums2 += (sum dy m, sum dy m xhat)                    [BatchNorm backward statistics of the previous layer]
//        pass 1:  dzprev = scale_prev (dy m - w (c1 + xhat c2))          [recomputes dy: 2 reads + 1 write instead of 4 + 2]
//
// Tensors are NCHW planes (samples, C, P = rows*ns).  A wave owns 64 consecutive positions of one sample as FOUR
// interleaved MFMA tiles (tile t = positions 4j + t, j = lane & 15): every global access is a 16-byte load/store of 4
// consecutive positions of one channel, 256 contiguous bytes per 16 lanes, and the D layout of the MFMA
// (channel 16v + 4g + r, position j) re-assembles into the same float4s without any shuffle.  Weights (<= 16 KiB packed)
// sit in LDS for the whole kernel.  The kernels are HBM-bound by design: 2 reads + 2 writes (forward), 2 + 0 and 2 + 1
// (backward) of the activation planes, against conv + 2 BatchNorm passes + ReLU-free apply of the unfused path.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdlib.h>

#include "fused_common.h"
#include "rtk_common.h"
#include "rtk_train.h"

namespace {

constexpr int TC_T = 256;            // 4 waves
constexpr int TC_CHUNK = 64;         // positions per wave iteration

struct TcParams {
    int samples, rows, lg_ns, groups, P;
    const float *in;        // (S, 16U, P): x (forward) or dz (backward)
    const float *w_packed;  // the layer's weight W, row-major (cout, cin) of the FORWARD convolution (fragments are built on load)
    const float *pre;       // (4, groups, Cpre) mean | rstd | scale | shift of the previous layer's BatchNorm; NULL = identity (forward only)
    const float *zprev;     // backward: (S, 16V, P)
    const float *rw;        // (S, rows) row weights or NULL
    float *out;             // forward: z (S, 16V, P); backward pass 1: dzprev
    float *act_out;         // forward: a (S, 16U, P) or NULL
    double *sums;           // (groups, 16V, 2)
    double count;
    float *dgb;             // backward pass 1: (2, 16V) dgamma | dbeta
    float *partial;         // weight gradient: one (16V, 16U) partial block per workgroup
    rtk_bn_fin_t fin;       // forward: fin.sums != NULL: the previous layer's BatchNorm is finalised here (and published into pre_out)
    float *pre_out;
    // backward with a POOLED source: `in` is not dz but z of the pooled (last) layer; its dz is formed on load,
    //     dz = scale ((k == karg ? d : 0) - w (c1 + xhat c2)),   d = dout[row], (c1, c2) = pool_sums / count, xhat = (z - mean) rstd
    const float *pool_dout;         // (S, 16U, rows)
    const unsigned char *pool_karg; // (S, 16U, rows): index of the row's arg-max element, 255 = no gradient
    const float *pool_par;          // (4, groups, 16U) of the pooled layer's BatchNorm
    const double *pool_sums;        // (RTK_STAT_SLOTS, groups, 16U, 2): rtk_pool_bwd_stats_arg
    float *pool_dgb;                // (2, 16U) dgamma | dbeta of the pooled layer's BatchNorm (written by the apply pass)
    const float *gamma_prev, *beta_prev;   // wgrad+stats: the previous layer's BatchNorm affine parameters (16 cin each)
    double *stats_out;              // wgrad+stats: (RTK_STAT_SLOTS, groups, cin, 2) float64, zero-initialised
};

// dz of a pooled layer from z (four consecutive positions of one row: ns >= 4), see TcParams
struct PoolCoef {
    float mean, rstd, sc, c1, c2;
};
__device__ __forceinline__ f4 pool_dz(const f4 z, const PoolCoef k, float d, int karg, int k0, float wl) {
    f4 o;
    o.x = k.sc * ((karg == k0 ? d : 0.f) - wl * (k.c1 + (z.x - k.mean) * k.rstd * k.c2));
    o.y = k.sc * ((karg == k0 + 1 ? d : 0.f) - wl * (k.c1 + (z.y - k.mean) * k.rstd * k.c2));
    o.z = k.sc * ((karg == k0 + 2 ? d : 0.f) - wl * (k.c1 + (z.z - k.mean) * k.rstd * k.c2));
    o.w = k.sc * ((karg == k0 + 3 ? d : 0.f) - wl * (k.c1 + (z.w - k.mean) * k.rstd * k.c2));
    return o;
}

// MODE 0: forward.  MODE 1: backward statistics.  MODE 2: backward apply.  POOL (backward): pooled source, see TcParams.
template <int U, int V, int MODE, bool POOL = false>
__global__ __launch_bounds__(TC_T, 2) void conv_bn_kernel(const TcParams Q) {
    __shared__ __attribute__((aligned(16))) f4 s_w[U * V * 64];
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int grp = b / (Q.samples / Q.groups);
    // A-operand fragments straight from the row-major weight (no host-side packing): fragment (u,v), lane (g,i) holds
    // Wm[16v+i][16u+4g .. +3] where Wm = W (forward, (16V,16U)) or W^T (backward, W is (16U,16V))
    for (int e = threadIdx.x; e < U * V * 64; e += TC_T) {
        const int f = e >> 6, l = e & 63, fu = f / V, fv = f % V, fg = l >> 4, fi = l & 15;
        f4 w;
        if (MODE == 0) {
            w = *reinterpret_cast<const f4 *>(Q.w_packed + (size_t)(16 * fv + fi) * (16 * U) + 16 * fu + 4 * fg);
        } else {
            const float *src = Q.w_packed + (size_t)(16 * fu + 4 * fg) * (16 * V) + 16 * fv + fi;
            w = (f4){src[0], src[16 * V], src[2 * 16 * V], src[3 * 16 * V]};
        }
        s_w[e] = w;
    }
    __syncthreads();
    const int P = Q.P;
    // BatchNorm constants of the channels on the affine side (inputs in the forward, outputs in the backward), in LDS:
    // lanes that share g read the same word (broadcast), and 6 x 16V registers per lane would spill
    constexpr int CA = 16 * (MODE == 0 ? U : V);
    __shared__ float s_sc[CA], s_sh[CA], s_mu[CA], s_rs[CA], s_c1[CA], s_c2[CA];
    const bool fin_pre = MODE == 0 && Q.fin.sums != nullptr;
    const bool has_pre = Q.pre != nullptr || fin_pre;
    {
        const size_t GC = (size_t)Q.groups * CA, o = (size_t)grp * CA;
        for (int c = threadIdx.x; c < CA; c += TC_T) {
            if (fin_pre) {      // finalise the previous BatchNorm from its batch sums; the first workgroup publishes par + running statistics
                bn_fin_constants(Q.fin, CA, Q.groups, grp, c, s_mu[c], s_rs[c], s_sc[c], s_sh[c]);
                if (b == 0 && blockIdx.x == 0) bn_fin_publish(Q.fin, CA, Q.groups, c, Q.pre_out);
                continue;
            }
            s_mu[c] = has_pre ? Q.pre[o + c] : 0.f;
            s_rs[c] = has_pre ? Q.pre[GC + o + c] : 1.f;
            s_sc[c] = has_pre ? Q.pre[2 * GC + o + c] : 1.f;
            s_sh[c] = has_pre ? Q.pre[3 * GC + o + c] : 0.f;
            if (MODE == 2) {
                {
                    double v0_, v1_;
                    rtk_stat_read2<RTK_STAT_BACKWARD>(Q.sums, GC * 2, (o + c) * 2, v0_, v1_);
                    s_c1[c] = (float)(v0_ / Q.count);
                    s_c2[c] = (float)(v1_ / Q.count);
                }
            }
        }
    }
    // pooled source: constants of the dz-side channels (16U)
    __shared__ float s_pm[POOL ? 16 * U : 1], s_pr[POOL ? 16 * U : 1], s_ps[POOL ? 16 * U : 1], s_p1[POOL ? 16 * U : 1], s_p2[POOL ? 16 * U : 1];
    if constexpr (POOL) {
        const size_t GC = (size_t)Q.groups * 16 * U, o = (size_t)grp * 16 * U;
        for (int c = threadIdx.x; c < 16 * U; c += TC_T) {
            s_pm[c] = Q.pool_par[o + c];
            s_pr[c] = Q.pool_par[GC + o + c];
            s_ps[c] = Q.pool_par[2 * GC + o + c];
            {
                double v0_, v1_;
                rtk_stat_read2<RTK_STAT_BACKWARD>(Q.pool_sums, GC * 2, (o + c) * 2, v0_, v1_);
                s_p1[c] = (float)(v0_ / Q.count);
                s_p2[c] = (float)(v1_ / Q.count);
            }
        }
        if (Q.pool_dgb && b == 0 && blockIdx.x == 0 && threadIdx.x < 16 * U) {
            const int c = threadIdx.x;
            double db = 0.0, dg = 0.0;
            for (int gg = 0; gg < Q.groups; ++gg) {
                {
                    double v0_, v1_;
                    rtk_stat_read2<RTK_STAT_BACKWARD>(Q.pool_sums, GC * 2, ((size_t)gg * 16 * U + c) * 2, v0_, v1_);
                    db += v0_;
                    dg += v1_;
                }
            }
            Q.pool_dgb[c] = (float)dg;
            Q.pool_dgb[16 * U + c] = (float)db;
        }
    }
    __syncthreads();
    if (MODE == 2 && Q.dgb && b == 0 && blockIdx.x == 0 && threadIdx.x < 16 * V) {      // parameter gradients: sum over the groups
        const int c = threadIdx.x;
        double db = 0.0, dg = 0.0;
        for (int gg = 0; gg < Q.groups; ++gg) {
            {
                double v0_, v1_;
                rtk_stat_read2<RTK_STAT_BACKWARD>(Q.sums, (size_t)Q.groups * 16 * V * 2, ((size_t)gg * 16 * V + c) * 2, v0_, v1_);
                db += v0_;
                dg += v1_;
            }
        }
        Q.dgb[c] = (float)dg;
        Q.dgb[16 * V + c] = (float)db;
    }
    f4 st0[V], st1[V];      // per-lane partial statistics (float within the wave's chunks, float64 across waves)
#pragma unroll
    for (int v = 0; v < V; ++v) { st0[v] = f4_zero(); st1[v] = f4_zero(); }

    // uniform (SGPR) plane bases + 32-bit per-lane byte offsets: 64-bit per-channel addresses would cost 2 VGPRs each
    const char *inb = reinterpret_cast<const char *>(Q.in + (size_t)b * 16 * U * P);
    const char *zpb = MODE != 0 ? reinterpret_cast<const char *>(Q.zprev + (size_t)b * 16 * V * P) : nullptr;
    char *outb = Q.out ? reinterpret_cast<char *>(Q.out + (size_t)b * 16 * V * P) : nullptr;
    char *actb = (MODE == 0 && Q.act_out) ? reinterpret_cast<char *>(Q.act_out + (size_t)b * 16 * U * P) : nullptr;
    const unsigned pitch = 4u * (unsigned)P;                      // bytes per channel plane
    const int nchunks = (P + TC_CHUNK - 1) / TC_CHUNK;
    const int stride = gridDim.x * (TC_T / 64);
    // Register plan (U = V = 4): input tiles 64 + zprev of half the output blocks 32 (backward) + accumulators of HALF the
    // output blocks 32 + statistics 32.  The backward requests a pass's zprev rows before that pass's MFMAs and every mode
    // requests the next chunk's input at the end of the current one -- nothing is waited for right after issue.
    // output blocks per MFMA pass; the statistics pass (32 accumulator registers more) takes quarter passes and stays at 3 waves/SIMD
    constexpr int VH = V >= 4 ? ((MODE == 1 || (POOL && U >= 4)) ? V / 4 : V / 2) : V;      // (quarter passes where the prefetched pooled values need the registers)
    f4 xin[U][4];
    // ... and with it what else the chunk needs from memory: the rows' weight and, with a pooled source, every channel's (dout, karg) of
    // the row.  Loaded where they are used (the top of the chunk), each chunk waited a round trip for them before its first MFMA.
    float wl_in = 1.f, pd_in[POOL ? U : 1][4];
    int pk_in[POOL ? U : 1][4];
    auto load_in = [&](int chunk, f4 (&dst)[U][4]) {
        int pp = chunk * TC_CHUNK + 4 * j;
        pp = pp < P ? pp : P - 4;         // tail lanes re-read the last valid float4 (unconditional loads: no branch per row);
                                          // their results are never stored and enter no sum
        wl_in = Q.rw ? Q.rw[(size_t)b * Q.rows + (pp >> Q.lg_ns)] : 1.f;
        if constexpr (POOL) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const size_t ro = ((size_t)b * 16 * U + 16 * u + 4 * g + r) * Q.rows + (pp >> Q.lg_ns);
                    pd_in[u][r] = Q.pool_dout[ro];
                    pk_in[u][r] = (int)Q.pool_karg[ro];
                }
        }
        // ONE per-lane byte offset (lane's channel group + position), opaque to the optimiser, added to uniform row pointers:
        // otherwise every one of the 16U + 32V row addresses is hoisted as a loop-invariant 64-bit VGPR pair and spills
        unsigned lo = (unsigned)(4 * g) * pitch + 4u * (unsigned)pp;
        asm volatile("" : "+v"(lo));
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                dst[u][r] = *reinterpret_cast<const f4 *>(inb + (size_t)((unsigned)(16 * u + r) * pitch) + lo);
    };
    int ch = blockIdx.x * (TC_T / 64) + wave;
    if (ch < nchunks) load_in(ch, xin);
    for (; ch < nchunks; ch += stride) {
        const int p = ch * TC_CHUNK + 4 * j;                       // this lane's 4 positions
        const bool ok = p < P;                                     // P % 4 == 0: all four or none
        unsigned lane_off = (unsigned)(4 * g) * pitch + 4u * (unsigned)(ok ? p : P - 4);      // see load_in
        asm volatile("" : "+v"(lane_off));
        const float wl = ok ? wl_in : 0.f;
        asm volatile("" ::: "memory");      // re-read the BatchNorm constants from LDS every chunk: hoisted, they pin 4 x 16V registers

        // ---- tiles: h[t][u][r] = in[channel 16u + 4g + r][position p + t] (previous BatchNorm + ReLU applied in the forward) ----
        f4 h[4][U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f4 x = xin[u][r];
                if constexpr (POOL) {      // x holds z of the pooled layer: form its dz (the four positions share a row: ns >= 4)
                    const int cc = 16 * u + 4 * g + r;
                    const int pq = ok ? p : P - 4;
                    const PoolCoef kc = {s_pm[cc], s_pr[cc], s_ps[cc], s_p1[cc], s_p2[cc]};
                    x = pool_dz(x, kc, pd_in[u][r], pk_in[u][r], pq & ((1 << Q.lg_ns) - 1), wl);
                }
                if (MODE == 0) {
                    if (has_pre) {
                        const float sc = s_sc[16 * u + 4 * g + r], sh = s_sh[16 * u + 4 * g + r];
                        x.x = fmaxf(__fmaf_rn(x.x, sc, sh), 0.f);
                        x.y = fmaxf(__fmaf_rn(x.y, sc, sh), 0.f);
                        x.z = fmaxf(__fmaf_rn(x.z, sc, sh), 0.f);
                        x.w = fmaxf(__fmaf_rn(x.w, sc, sh), 0.f);
                    }
                    if (actb && ok) *reinterpret_cast<f4 *>(actb + (size_t)((unsigned)(16 * u + r) * pitch) + lane_off) = x;
                }
                h[0][u][r] = x.x; h[1][u][r] = x.y; h[2][u][r] = x.z; h[3][u][r] = x.w;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int v0 = 0; v0 < V; v0 += VH) {
            // backward: this pass's zprev rows are requested now and consumed after the pass's MFMAs
            f4 zp[MODE == 0 ? 1 : VH][4];
            if constexpr (MODE != 0) {
#pragma unroll
                for (int q = 0; q < VH; ++q)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        zp[q][r] = *reinterpret_cast<const f4 *>(zpb + (size_t)((unsigned)(16 * (v0 + q) + r) * pitch) + lane_off);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- 4 tiles x (16 VH x 16U) MFMA --------------------------------------------------------------------------------
            f4 acc[4][VH];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int q = 0; q < VH; ++q) acc[t][q] = f4_zero();
#pragma unroll
            for (int u = 0; u < U; ++u) {
                f4 wf[VH];
#pragma unroll
                for (int q = 0; q < VH; ++q) wf[q] = s_w[(u * V + v0 + q) * 64 + lane];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int q = 0; q < VH; ++q) {
                        acc[t][q] = mfma4(wf[q].x, h[t][u].x, acc[t][q]);
                        acc[t][q] = mfma4(wf[q].y, h[t][u].y, acc[t][q]);
                        acc[t][q] = mfma4(wf[q].z, h[t][u].z, acc[t][q]);
                        acc[t][q] = mfma4(wf[q].w, h[t][u].w, acc[t][q]);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- epilogue: out[channel 16v + 4g + r][p .. p+3] = (acc[0..3][v][r]) -------------------------------------------
#pragma unroll
            for (int q = 0; q < VH; ++q) {
                const int v = v0 + q;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const f4 y = {acc[0][q][r], acc[1][q][r], acc[2][q][r], acc[3][q][r]};
                    char *orow = outb + (size_t)((unsigned)(16 * v + r) * pitch);
                    if constexpr (MODE == 0) {
                        if (ok) *reinterpret_cast<f4 *>(orow + lane_off) = y;
                        st0[v][r] += wl * ((y.x + y.y) + (y.z + y.w));
                        st1[v][r] += wl * ((y.x * y.x + y.y * y.y) + (y.z * y.z + y.w * y.w));
                    } else {
                        const f4 zq = zp[q][r];
                        const int cc = 16 * v + 4 * g + r;
                        const float sc = s_sc[cc], sh = s_sh[cc], mu = s_mu[cc], rs = s_rs[cc];
                        f4 d, xh;
                        d.x = (ok && __fmaf_rn(zq.x, sc, sh) > 0.f) ? y.x : 0.f;      // tail lanes contribute nothing
                        d.y = (ok && __fmaf_rn(zq.y, sc, sh) > 0.f) ? y.y : 0.f;
                        d.z = (ok && __fmaf_rn(zq.z, sc, sh) > 0.f) ? y.z : 0.f;
                        d.w = (ok && __fmaf_rn(zq.w, sc, sh) > 0.f) ? y.w : 0.f;
                        xh.x = (zq.x - mu) * rs; xh.y = (zq.y - mu) * rs; xh.z = (zq.z - mu) * rs; xh.w = (zq.w - mu) * rs;
                        if (MODE == 1) {
                            st0[v][r] += (d.x + d.y) + (d.z + d.w);
                            st1[v][r] += (d.x * xh.x + d.y * xh.y) + (d.z * xh.z + d.w * xh.w);
                        } else if (ok) {
                            const float k1 = s_c1[cc], k2 = s_c2[cc];
                            f4 o;
                            o.x = sc * (d.x - wl * (k1 + xh.x * k2));
                            o.y = sc * (d.y - wl * (k1 + xh.y * k2));
                            o.z = sc * (d.z - wl * (k1 + xh.z * k2));
                            o.w = sc * (d.w - wl * (k1 + xh.w * k2));
                            *reinterpret_cast<f4 *>(orow + lane_off) = o;
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ch + stride < nchunks) load_in(ch + stride, xin);      // the next chunk's input is in flight across the loop back-edge
    }
    if (MODE == 2) return;
    // ---- statistics: lanes -> wave (xor shuffles over j) -> workgroup (LDS) -> one float64 atomic pair per channel ------
    __shared__ double s_red[TC_T / 64][V * 16][2];
#pragma unroll
    for (int v = 0; v < V; ++v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double a0 = (double)st0[v][r], a1 = (double)st1[v][r];
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) { a0 += __shfl_xor(a0, o, 64); a1 += __shfl_xor(a1, o, 64); }
            if (j == 0) { s_red[wave][16 * v + 4 * g + r][0] = a0; s_red[wave][16 * v + 4 * g + r][1] = a1; }
        }
    }
    __syncthreads();
    if (threadIdx.x < 16 * V) {
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int w = 0; w < TC_T / 64; ++w) { a0 += s_red[w][threadIdx.x][0]; a1 += s_red[w][threadIdx.x][1]; }
        const size_t o2 = ((size_t)grp * 16 * V + threadIdx.x) * 2;
        rtk_stat_add<MODE == 0 ? RTK_STAT_FORWARD : RTK_STAT_BACKWARD>(Q.sums, (size_t)Q.groups * 16 * V * 2, b, o2, a0);
        rtk_stat_add<MODE == 0 ? RTK_STAT_FORWARD : RTK_STAT_BACKWARD>(Q.sums, (size_t)Q.groups * 16 * V * 2, b, o2 + 1, a1);
    }
}

// Weight gradient of z = W relu(BatchNorm(zprev)):  dW[co][ci] += sum_{b,p} dz[b][co][p] * a[b][ci][p],  a recomputed from
// zprev on load (the forward does not have to store it).  Both operands are contiguous along the contraction axis p, so
// a lane's float4 along p IS four MFMA k-steps:  k-step s of lane group g contracts position p0 + 4g + s, for the A operand
// (lane (g,i) <- dz row 16v+i) and the B operand (lane (g,j) <- a row 16u+j) alike.  The 16V x 16U accumulator block stays
// in registers over the wave's whole position range; waves are reduced through LDS, workgroups with float atomics.
template <int U, int V>
__global__ __launch_bounds__(TC_T, 2) void conv_wgrad_kernel(const TcParams Q) {
    __shared__ float s_red[TC_T / 64][16 * V][16 * U + 1];      // one image per wave: written in parallel, added on the way out
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int grp = b / (Q.samples / Q.groups);
    const int P = Q.P;
    const unsigned pitch = 4u * (unsigned)P;
    const char *dzb = reinterpret_cast<const char *>(Q.in + (size_t)b * 16 * V * P);        // dz (S, 16V, P)
    const char *zpb = reinterpret_cast<const char *>(Q.zprev + (size_t)b * 16 * U * P);     // zprev (S, 16U, P)
    float sc[U], sh[U];
    {
        const size_t GC = (size_t)Q.groups * 16 * U, o = (size_t)grp * 16 * U;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            sc[u] = Q.pre[2 * GC + o + 16 * u + j];
            sh[u] = Q.pre[3 * GC + o + 16 * u + j];
        }
    }
    f4 acc[V][U];
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
        for (int u = 0; u < U; ++u) acc[v][u] = f4_zero();
    const int ntiles = (P + 15) / 16;
    // unconditional loads (tail lanes re-read the last float4 of the plane and are zeroed afterwards: a load under a per-lane
    // condition is a basic block of its own, waited for before the next one is issued); the next tile's operands are requested
    // before this tile's MFMAs
    auto fetch = [&](int t, f4 (&a)[V], f4 (&x)[U]) {
        const int p = 16 * t + 4 * g;
        const unsigned po = 4u * (unsigned)(p < P ? p : P - 4);
#pragma unroll
        for (int v = 0; v < V; ++v) a[v] = *reinterpret_cast<const f4 *>(dzb + (unsigned)(16 * v + j) * pitch + po);
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = *reinterpret_cast<const f4 *>(zpb + (unsigned)(16 * u + j) * pitch + po);
    };
    const int tstep = gridDim.x * (TC_T / 64);
    int t = blockIdx.x * (TC_T / 64) + wave;
    f4 an[V], xn[U];
    if (t < ntiles) fetch(t, an, xn);
    for (; t < ntiles; t += tstep) {
        const bool ok = 16 * t + 4 * g < P;
        f4 a[V], bq[U];
#pragma unroll
        for (int v = 0; v < V; ++v) a[v] = ok ? an[v] : f4_zero();
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f4 x = xn[u];
            x.x = fmaxf(__fmaf_rn(x.x, sc[u], sh[u]), 0.f);
            x.y = fmaxf(__fmaf_rn(x.y, sc[u], sh[u]), 0.f);
            x.z = fmaxf(__fmaf_rn(x.z, sc[u], sh[u]), 0.f);
            x.w = fmaxf(__fmaf_rn(x.w, sc[u], sh[u]), 0.f);
            bq[u] = x;
        }
        if (t + tstep < ntiles) fetch(t + tstep, an, xn);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int v = 0; v < V; ++v)
#pragma unroll
                for (int u = 0; u < U; ++u) acc[v][u] = mfma4(a[v][q], bq[u][q], acc[v][u]);
    }
    // D layout: acc[v][u][r] = dW[16v + 4g + r][16u + j]; every wave stores its own LDS image (one round, one barrier), the images
    // are added on the way out
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) s_red[wave][16 * v + 4 * g + r][16 * u + j] = acc[v][u][r];
    __syncthreads();
    static_assert(TC_T == 256, "four waves");
    // this workgroup's partial dW: plain stores, added up in a fixed order by conv_wgrad_reduce_kernel (as float atomics, 256 U V per
    // workgroup and up to 512 workgroups deep on every address, they were a large part of the kernel)
    float *part = Q.partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (256 * U * V);
    for (int e = threadIdx.x; e < 16 * V * 16 * U; e += TC_T) {
        const int co = e / (16 * U), ci = e % (16 * U);
        part[e] = (s_red[0][co][ci] + s_red[1][co][ci]) + (s_red[2][co][ci] + s_red[3][co][ci]);
    }
}

// dw[e] += sum over the workgroups' partials: 16 outputs x 16 lanes over the workgroup axis, eight independent loads per lane and round
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(int n, int wgs, const float *__restrict__ partial, float *__restrict__ dw) {
    __shared__ float s_part[16][17];
    const int el = threadIdx.x & 15, zl = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;
    const float *src = partial + (e < n ? e : n - 1);
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = 0.f;
    for (int z0 = zl; z0 < wgs; z0 += 128) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int z = z0 + 16 * k;
            const float v = src[(size_t)(z < wgs ? z : 0) * n];
            a[k] += z < wgs ? v : 0.f;
        }
    }
    s_part[zl][el] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    __syncthreads();
    if (zl || e >= n) return;
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) sum += s_part[q][el];
    dw[e] += sum;
}

// ---- weight gradient + BatchNorm-backward statistics in ONE pass over (dz, zprev) -------------------------------------------------
// For z = W a, a = relu(BatchNorm(zprev)) = m (gamma xhat + beta) with m the ReLU mask and xhat the normalised zprev, define
//     G[co][ci] = sum_p dz[co][p] m[ci][p],      H[co][ci] = sum_p dz[co][p] m[ci][p] xhat[ci][p]        (per statistics group).
// Then   dW[co][ci]  = sum_p dz a = gamma[ci] H + beta[ci] G                                              (summed over the groups)
// and the statistics the backward of the previous BatchNorm needs, with dy = W^T dz:
//     sum_p dy[ci] m      = sum_co W[co][ci] G[co][ci],        sum_p dy[ci] m xhat = sum_co W[co][ci] H[co][ci].
// Two position-contractions on MFMA (the operands of the old weight-gradient kernel, B = m and B = m xhat instead of B = a) replace
// the weight-gradient kernel AND the statistics pass of rtk_conv_bn_bwd, which read the same two tensors a second time.  One wave
// per SIMD (two 16V x 16U accumulator blocks = 128 registers at 64 x 64), the next tile's operands in flight across the MFMAs.
#ifndef TS_OCC
#define TS_OCC 1      // 2 waves per SIMD: 256 registers, 139 spilled, the train step 8.94 -> 9.99 ms
#endif
template <int U, int V, bool POOL>
__global__ __launch_bounds__(TC_T, TS_OCC) void conv_wgrad_stats_kernel(const TcParams Q) {
    __shared__ float s_red[TC_T / 64][16 * V][16 * U + 1];
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int grp = b / (Q.samples / Q.groups);
    const int P = Q.P;
    const unsigned pitch = 4u * (unsigned)P;
    const char *dzb = reinterpret_cast<const char *>(Q.in + (size_t)b * 16 * V * P);        // dz, or z of the pooled layer (S, 16V, P)
    const char *zpb = reinterpret_cast<const char *>(Q.zprev + (size_t)b * 16 * U * P);     // zprev (S, 16U, P)
    float sc[U], sh[U], mu[U], rs[U];
    {
        const size_t GC = (size_t)Q.groups * 16 * U, o = (size_t)grp * 16 * U;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            mu[u] = Q.pre[o + 16 * u + j];
            rs[u] = Q.pre[GC + o + 16 * u + j];
            sc[u] = Q.pre[2 * GC + o + 16 * u + j];
            sh[u] = Q.pre[3 * GC + o + 16 * u + j];
        }
    }
    PoolCoef pk[POOL ? V : 1];
    if constexpr (POOL) {
        const size_t GC = (size_t)Q.groups * 16 * V, o = (size_t)grp * 16 * V;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int c = 16 * v + j;
            pk[v].mean = Q.pool_par[o + c];
            pk[v].rstd = Q.pool_par[GC + o + c];
            pk[v].sc = Q.pool_par[2 * GC + o + c];
            {
                double v0_, v1_;
                rtk_stat_read2<RTK_STAT_BACKWARD>(Q.pool_sums, GC * 2, (o + c) * 2, v0_, v1_);
                pk[v].c1 = (float)(v0_ / Q.count);
                pk[v].c2 = (float)(v1_ / Q.count);
            }
        }
    }
    f4 accG[V][U], accH[V][U];
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
        for (int u = 0; u < U; ++u) { accG[v][u] = f4_zero(); accH[v][u] = f4_zero(); }
    // The two contractions run on the bf16 matrix pipe (v_mfma_f32_16x16x32_bf16, the C/D layout of the fp32-input 16x16x4): a tile is
    // 32 positions, lane (g, j) holds EIGHT consecutive positions p0 + 8g .. + 7 of row j of every operand block (two float4 loads,
    // 128 contiguous bytes per row and tile).  dz and m xhat are taken as three exact bf16 pieces each (split_mfma.h), the mask m is
    // 0 / 1 -- exact in ONE piece: G = dz m^T is three MFMAs per block, H = dz (m xhat)^T six, against 2 x 8 fp32-input MFMAs of twice
    // the issue time for the same 32 positions (512 -> 144 cycles per block pair; the splitting is ~45 VALU per operand block).
    const int ntiles = (P + 31) / 32;
    struct Tile {
        f4 a[V][2], x[U][2];
        float d[POOL ? V : 1][2];
        int k[POOL ? V : 1][2];
        float wl[2];      // the rows' weights, requested with the tile (loaded where they are used, each tile waited a round trip for them)
    };
    auto fetch = [&](int t, Tile &T) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int p = 32 * t + 8 * g + 4 * h;
            const int pq = p < P ? p : P - 4;
            const unsigned po = 4u * (unsigned)pq;
#pragma unroll
            for (int v = 0; v < V; ++v) T.a[v][h] = *reinterpret_cast<const f4 *>(dzb + (unsigned)(16 * v + j) * pitch + po);
#pragma unroll
            for (int u = 0; u < U; ++u) T.x[u][h] = *reinterpret_cast<const f4 *>(zpb + (unsigned)(16 * u + j) * pitch + po);
            if constexpr (POOL) {
                T.wl[h] = Q.rw ? Q.rw[(size_t)b * Q.rows + (pq >> Q.lg_ns)] : 1.f;
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const size_t ro = ((size_t)b * 16 * V + 16 * v + j) * Q.rows + (pq >> Q.lg_ns);
                    T.d[v][h] = Q.pool_dout[ro];
                    T.k[v][h] = (int)Q.pool_karg[ro];
                }
            }
        }
    };
    const int tstep = gridDim.x * (TC_T / 64);
    int t = blockIdx.x * (TC_T / 64) + wave;
    Tile nxt;
    if (t < ntiles) fetch(t, nxt);
    for (; t < ntiles; t += tstep) {
        u4v ap[V][3], bm[U], bx[U][3];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            f4 q[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int p = 32 * t + 8 * g + 4 * h;
                const bool ok = p < P;
                q[h] = nxt.a[v][h];
                if constexpr (POOL) q[h] = pool_dz(q[h], pk[v], nxt.d[v][h], nxt.k[v][h], (ok ? p : P - 4) & ((1 << Q.lg_ns) - 1), nxt.wl[h]);
                q[h] = ok ? q[h] : f4_zero();
            }
            split3(q[0], q[1], ap[v]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f4 xh[2];
            unsigned mb[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f4 x = nxt.x[u][h];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool on = __fmaf_rn(x[e], sc[u], sh[u]) > 0.f;
                    mb[4 * h + e] = on ? 0x3F80u : 0u;                       // bf16 1.0 / 0
                    xh[h][e] = on ? (x[e] - mu[u]) * rs[u] : 0.f;
                }
            }
#pragma unroll
            for (int w = 0; w < 4; ++w) bm[u][w] = mb[2 * w] | (mb[2 * w + 1] << 16);
            split3(xh[0], xh[1], bx[u]);
        }
        if (t + tstep < ntiles) fetch(t + tstep, nxt);
#pragma unroll
        for (int v = 0; v < V; ++v)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                accG[v][u] = mfma16_bf(ap[v][2], bm[u], accG[v][u]);
                accG[v][u] = mfma16_bf(ap[v][1], bm[u], accG[v][u]);
                accG[v][u] = mfma16_bf(ap[v][0], bm[u], accG[v][u]);
                accH[v][u] = mfma16_bf(ap[v][2], bx[u][0], accH[v][u]);      // small terms first
                accH[v][u] = mfma16_bf(ap[v][1], bx[u][1], accH[v][u]);
                accH[v][u] = mfma16_bf(ap[v][0], bx[u][2], accH[v][u]);
                accH[v][u] = mfma16_bf(ap[v][1], bx[u][0], accH[v][u]);
                accH[v][u] = mfma16_bf(ap[v][0], bx[u][1], accH[v][u]);
                accH[v][u] = mfma16_bf(ap[v][0], bx[u][0], accH[v][u]);
            }
    }
    // D layout: acc[v][u][r] = out[16v + 4g + r][16u + j]; per-wave LDS images added on the way out, G then H
    float *part = Q.partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (2 * 256 * U * V);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();
#pragma unroll
        for (int v = 0; v < V; ++v)
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) s_red[wave][16 * v + 4 * g + r][16 * u + j] = half ? accH[v][u][r] : accG[v][u][r];
        __syncthreads();
        for (int e = threadIdx.x; e < 16 * V * 16 * U; e += TC_T) {
            const int co = e / (16 * U), ci = e % (16 * U);
            part[half * (256 * U * V) + e] = (s_red[0][co][ci] + s_red[1][co][ci]) + (s_red[2][co][ci] + s_red[3][co][ci]);
        }
    }
}

// partials (wgs, 2, n) -> per group G_g, H_g (fixed summation order) -> dw[e] += sum_g gamma[ci] H_g + beta[ci] G_g, and the
// BatchNorm-backward statistics stats[g][ci] += (W[co][ci] G_g, W[co][ci] H_g) (rtk_stat_add: 16V contributions per value)
__global__ __launch_bounds__(256) void conv_wgrad_stats_reduce_kernel(int n, int cin, int wgs, int groups, const float *__restrict__ partial,
                                                                      const float *__restrict__ w, const float *__restrict__ gamma,
                                                                      const float *__restrict__ beta, float *__restrict__ dw,
                                                                      double *__restrict__ stats) {
    __shared__ float s_part[2][16][17];
    const int el = threadIdx.x & 15, zl = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;
    const float *src = partial + (e < n ? e : n - 1);
    const int per = wgs / groups;
    float dwe = 0.f;
    for (int grp = 0; grp < groups; ++grp) {
        float a[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int k = 0; k < 4; ++k) a[h][k] = 0.f;
        for (int z0 = zl; z0 < per; z0 += 64) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int z = z0 + 16 * k;
                const size_t row = (size_t)(grp * per + (z < per ? z : 0)) * 2 * n;
                const float g0 = src[row], h0 = src[row + n];
                a[0][k] += z < per ? g0 : 0.f;
                a[1][k] += z < per ? h0 : 0.f;
            }
        }
        __syncthreads();
        s_part[0][zl][el] = (a[0][0] + a[0][1]) + (a[0][2] + a[0][3]);
        s_part[1][zl][el] = (a[1][0] + a[1][1]) + (a[1][2] + a[1][3]);
        __syncthreads();
        if (zl == 0 && e < n) {
            float G = 0.f, H = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) { G += s_part[0][q][el]; H += s_part[1][q][el]; }
            const int ci = e % cin;
            dwe += gamma[ci] * H + beta[ci] * G;
            const size_t o2 = ((size_t)grp * cin + ci) * 2;             // (RTK_STAT_SLOTS, groups, cin, 2)
            rtk_stat_add<RTK_STAT_BACKWARD>(stats, (size_t)groups * cin * 2, blockIdx.x, o2, (double)w[e] * (double)G);
            rtk_stat_add<RTK_STAT_BACKWARD>(stats, (size_t)groups * cin * 2, blockIdx.x, o2 + 1, (double)w[e] * (double)H);
        }
    }
    if (zl == 0 && e < n) dw[e] += dwe;
}

int ilog2x(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

template <int MODE, bool POOL = false>
int launch(const TcParams &Q, int cin, int cout, hipStream_t s) {
    const int U = cin / 16, V = cout / 16;
    const int nchunks = (Q.P + TC_CHUNK - 1) / TC_CHUNK;
    int gx = (nchunks + 3) / 4;                       // one chunk per wave per pass ...
    // ... but few, fat workgroups (the weight tile and the BatchNorm constants are staged per workgroup); measured per mode at the
    // largest set-abstraction shape (tools/experiments/exp_convbn.py): the forward and the statistics pass like them fatter than the apply pass
    const int max_wgs = MODE == 0 ? 1024 : MODE == 1 ? 512 : 4096;
    while ((long)gx * Q.samples > max_wgs && gx > 1) gx = (gx + 1) / 2;
    const dim3 grid(gx, Q.samples);
#define TC_CASE(u, v)                                                    \
    if (U == u && V == v) {                                              \
        conv_bn_kernel<u, v, MODE, POOL><<<grid, TC_T, 0, s>>>(Q);       \
        return 0;                                                        \
    }
    TC_CASE(1, 1) TC_CASE(1, 2) TC_CASE(1, 4) TC_CASE(2, 1) TC_CASE(2, 2) TC_CASE(2, 4) TC_CASE(4, 1) TC_CASE(4, 2) TC_CASE(4, 4)
#undef TC_CASE
    return 1;
}

int check(const char *who, int samples, int cin, int cout, int rows, int ns, int groups) {
    if ((double)rows * ns * 64.0 * 4.0 >= 4294967296.0) {
        rtk_set_error("%s: sample planes larger than 4 GiB (32-bit offsets)", who);
        return RTK_ERR_INVALID;
    }
    if (samples <= 0 || rows <= 0 || groups <= 0 || samples % groups || samples > 65535) {
        rtk_set_error("%s: bad batch (%d samples, %d groups)", who, samples, groups);
        return RTK_ERR_INVALID;
    }
    if (ilog2x(ns) < 2) {
        rtk_set_error("%s: ns (%d) must be a power of two >= 4", who, ns);
        return RTK_ERR_INVALID;
    }
    auto okc = [](int c) { return c == 16 || c == 32 || c == 64; };
    if (!okc(cin) || !okc(cout)) {
        rtk_set_error("%s: channel counts (%d -> %d) must be 16, 32 or 64", who, cin, cout);
        return RTK_ERR_UNSUPPORTED;
    }
    return RTK_OK;
}

}  // namespace

extern "C" int rtk_conv_bn_fwd(int samples, int cin, int cout, int rows, int ns, int groups, const float *x, const float *pre_par,
                               const float *w, float *z, float *act_out, const float *row_weight, double *sums,
                               rtk_stream_t stream) {
    if (int rc = check("rtk_conv_bn_fwd", samples, cin, cout, rows, ns, groups)) return rc;
    RTK_REQUIRE(x && w && z && sums, "rtk_conv_bn_fwd: null argument");
    TcParams Q = {};
    Q.samples = samples; Q.rows = rows; Q.lg_ns = ilog2x(ns); Q.groups = groups; Q.P = rows * ns;
    Q.in = x; Q.w_packed = w; Q.pre = pre_par; Q.rw = row_weight; Q.out = z; Q.act_out = act_out; Q.sums = sums;
    launch<0>(Q, cin, cout, (hipStream_t)stream);
    RTK_CHECK_LAUNCH("rtk_conv_bn_fwd");
    return RTK_OK;
}

extern "C" int rtk_conv_bn_fwd_fin(int samples, int cin, int cout, int rows, int ns, int groups, const float *x, const rtk_bn_fin_t *pre_fin,
                                   float *pre_par_out, const float *w, float *z, float *act_out, const float *row_weight, double *sums,
                                   rtk_stream_t stream) {
    if (int rc = check("rtk_conv_bn_fwd_fin", samples, cin, cout, rows, ns, groups)) return rc;
    RTK_REQUIRE(x && w && z && sums && pre_fin && pre_fin->sums && pre_fin->gamma && pre_fin->beta && pre_fin->count > 1.0 && pre_par_out,
                "rtk_conv_bn_fwd_fin: null argument");
    TcParams Q = {};
    Q.samples = samples; Q.rows = rows; Q.lg_ns = ilog2x(ns); Q.groups = groups; Q.P = rows * ns;
    Q.in = x; Q.w_packed = w; Q.pre = nullptr; Q.rw = row_weight; Q.out = z; Q.act_out = act_out; Q.sums = sums;
    Q.fin = *pre_fin; Q.pre_out = pre_par_out;
    launch<0>(Q, cin, cout, (hipStream_t)stream);
    RTK_CHECK_LAUNCH("rtk_conv_bn_fwd_fin");
    return RTK_OK;
}

extern "C" int rtk_conv_bn_bwd(int samples, int cprev, int cout, int rows, int ns, int groups, const float *dz, const float *w,
                               const float *zprev, const float *pre_par, const float *row_weight, double *sums2, double count, int apply,
                               float *dzprev, float *dgamma_dbeta, rtk_stream_t stream) {
    if (int rc = check("rtk_conv_bn_bwd", samples, cout, cprev, rows, ns, groups)) return rc;
    RTK_REQUIRE(dz && w && zprev && pre_par && sums2 && (!apply || dzprev), "rtk_conv_bn_bwd: null argument");
    TcParams Q = {};
    Q.samples = samples; Q.rows = rows; Q.lg_ns = ilog2x(ns); Q.groups = groups; Q.P = rows * ns;
    Q.in = dz; Q.w_packed = w; Q.pre = pre_par; Q.zprev = zprev; Q.rw = row_weight; Q.out = dzprev; Q.sums = sums2;
    Q.count = count; Q.dgb = dgamma_dbeta;
    if (apply) launch<2>(Q, cout, cprev, (hipStream_t)stream);
    else launch<1>(Q, cout, cprev, (hipStream_t)stream);
    RTK_CHECK_LAUNCH("rtk_conv_bn_bwd");
    return RTK_OK;
}

static void set_pool(TcParams &Q, const rtk_pool_src_t *pool) {
    Q.pool_dout = pool->dout; Q.pool_karg = pool->karg; Q.pool_par = pool->par; Q.pool_sums = pool->sums2; Q.pool_dgb = pool->dgamma_dbeta;
}

extern "C" int rtk_conv_bn_bwd_apply(int samples, int cprev, int cout, int rows, int ns, int groups, const float *dz_or_z, const rtk_pool_src_t *pool,
                                     const float *w, const float *zprev, const float *pre_par, const float *row_weight, const double *sums2,
                                     double count, float *dzprev, float *dgamma_dbeta, rtk_stream_t stream) {
    if (int rc = check("rtk_conv_bn_bwd_apply", samples, cout, cprev, rows, ns, groups)) return rc;
    RTK_REQUIRE(dz_or_z && w && zprev && pre_par && sums2 && dzprev, "rtk_conv_bn_bwd_apply: null argument");
    RTK_REQUIRE(!pool || (pool->dout && pool->karg && pool->par && pool->sums2), "rtk_conv_bn_bwd_apply: incomplete pooled source");
    TcParams Q = {};
    Q.samples = samples; Q.rows = rows; Q.lg_ns = ilog2x(ns); Q.groups = groups; Q.P = rows * ns;
    Q.in = dz_or_z; Q.w_packed = w; Q.pre = pre_par; Q.zprev = zprev; Q.rw = row_weight; Q.out = dzprev; Q.sums = const_cast<double *>(sums2);
    Q.count = count; Q.dgb = dgamma_dbeta;
    if (pool) {
        set_pool(Q, pool);
        launch<2, true>(Q, cout, cprev, (hipStream_t)stream);
    } else {
        launch<2, false>(Q, cout, cprev, (hipStream_t)stream);
    }
    RTK_CHECK_LAUNCH("rtk_conv_bn_bwd_apply");
    return RTK_OK;
}

extern "C" int rtk_conv_wgrad_stats(int samples, int cprev, int cout, int rows, int ns, int groups, const float *dz_or_z, const rtk_pool_src_t *pool,
                                    const float *zprev, const float *pre_par, const float *row_weight, double count, const float *w,
                                    const float *gamma_prev, const float *beta_prev, float *dw, double *sums2_out, float *workspace,
                                    long workspace_floats, rtk_stream_t stream) {
    if (int rc = check("rtk_conv_wgrad_stats", samples, cprev, cout, rows, ns, groups)) return rc;
    RTK_REQUIRE(dz_or_z && zprev && pre_par && w && gamma_prev && beta_prev && dw && sums2_out && workspace, "rtk_conv_wgrad_stats: null argument");
    RTK_REQUIRE(!pool || (pool->dout && pool->karg && pool->par && pool->sums2), "rtk_conv_wgrad_stats: incomplete pooled source");
    TcParams Q = {};
    Q.samples = samples; Q.rows = rows; Q.lg_ns = ilog2x(ns); Q.groups = groups; Q.P = rows * ns;
    Q.in = dz_or_z; Q.zprev = zprev; Q.pre = pre_par; Q.rw = row_weight; Q.count = count; Q.partial = workspace;
    if (pool) set_pool(Q, pool);
    const int U = cprev / 16, V = cout / 16;
    const int ntiles = (Q.P + 31) / 32;                                  // 32-position tiles (conv_wgrad_stats_kernel)
    RTK_REQUIRE(workspace_floats >= 2L * samples * cprev * cout, "rtk_conv_wgrad_stats: workspace of %ld floats < %ld", workspace_floats,
                2L * samples * cprev * cout);
    int gx = (ntiles + 3) / 4;                                           // at least one tile per wave ...
    while (((long)gx * samples > 512 * TS_OCC || 2L * gx * samples * cprev * cout > workspace_floats) && gx > 1) gx = (gx + 1) / 2;
    const dim3 grid(gx, samples);                                        // ... and about two one-wave-per-SIMD workgroups per CU
    hipStream_t s = (hipStream_t)stream;
#define TS_CASE(u, v)                                                                  \
    if (U == u && V == v) {                                                            \
        if (pool) conv_wgrad_stats_kernel<u, v, true><<<grid, TC_T, 0, s>>>(Q);        \
        else conv_wgrad_stats_kernel<u, v, false><<<grid, TC_T, 0, s>>>(Q);            \
    }
    TS_CASE(1, 1) TS_CASE(1, 2) TS_CASE(1, 4) TS_CASE(2, 1) TS_CASE(2, 2) TS_CASE(2, 4) TS_CASE(4, 1) TS_CASE(4, 2) TS_CASE(4, 4)
#undef TS_CASE
    RTK_CHECK_LAUNCH("rtk_conv_wgrad_stats");
    conv_wgrad_stats_reduce_kernel<<<(cprev * cout + 15) / 16, 256, 0, s>>>(cprev * cout, cprev, gx * samples, groups, workspace, w, gamma_prev,
                                                                          beta_prev, dw, sums2_out);
    RTK_CHECK_LAUNCH("rtk_conv_wgrad_stats");
    return RTK_OK;
}

extern "C" int rtk_conv_wgrad(int samples, int cprev, int cout, int rows, int ns, int groups, const float *dz, const float *zprev,
                              const float *pre_par, float *dw, float *workspace, long workspace_floats, rtk_stream_t stream) {
    if (int rc = check("rtk_conv_wgrad", samples, cprev, cout, rows, ns, groups)) return rc;
    RTK_REQUIRE(dz && zprev && pre_par && dw && workspace, "rtk_conv_wgrad: null argument");
    TcParams Q = {};
    Q.samples = samples; Q.rows = rows; Q.lg_ns = ilog2x(ns); Q.groups = groups; Q.P = rows * ns;
    Q.in = dz; Q.zprev = zprev; Q.pre = pre_par; Q.out = dw; Q.partial = workspace;
    const int U = cprev / 16, V = cout / 16;
    const int ntiles = (Q.P + 15) / 16;
    RTK_REQUIRE(workspace_floats >= (long)samples * cprev * cout, "rtk_conv_wgrad: workspace of %ld floats < %ld", workspace_floats,
                (long)samples * cprev * cout);
    int gx = (ntiles + 3) / 4;                                           // at least one tile per wave ...
    while (((long)gx * samples > 1024 || (long)gx * samples * cprev * cout > workspace_floats) && gx > 1) gx = (gx + 1) / 2;
    const dim3 grid(gx, samples);                                        // ... and about four workgroups (one partial dW each) per CU
    hipStream_t s = (hipStream_t)stream;
#define TW_CASE(u, v)                                        \
    if (U == u && V == v) conv_wgrad_kernel<u, v><<<grid, TC_T, 0, s>>>(Q);
    TW_CASE(1, 1) TW_CASE(1, 2) TW_CASE(1, 4) TW_CASE(2, 1) TW_CASE(2, 2) TW_CASE(2, 4) TW_CASE(4, 1) TW_CASE(4, 2) TW_CASE(4, 4)
#undef TW_CASE
    RTK_CHECK_LAUNCH("rtk_conv_wgrad");
    conv_wgrad_reduce_kernel<<<(cprev * cout + 15) / 16, 256, 0, s>>>(cprev * cout, gx * samples, workspace, dw);
    RTK_CHECK_LAUNCH("rtk_conv_wgrad");
    return RTK_OK;
}
