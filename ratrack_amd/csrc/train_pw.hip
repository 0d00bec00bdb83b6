// Training-mode PER-POINT layers (include/rtk_train.h): the 1x1 convolutions that act on one row per point / centroid --
// feature-propagation MLPs (lib/pointnet2_modules.py:140-158), the nn.Linear bottlenecks of PNHead and the layer-1 feature
// projections (utils/model_utils/model_utils.py:393-424), the predictor heads (:308-357).  The reference runs each of them as
// cat -> Conv2d -> (BatchNorm2d -> ReLU) with the framework's backward (convolution_backward + reductions); here
//
//   rtk_pw_conv   z = W . [src_0 ; src_1 ; ...] (+ bias), optionally with the weighted batch sums of z (BatchNorm statistics)
//                 in the epilogue.  The concatenation is VIRTUAL: every operand is its own tensor, channel-major planes
//                 (sample, channel, position) or point-major rows (sample, position, channel), any strides -- no cat, no
//                 transposes, no .contiguous() copies.  With transpose_w the same kernel is the INPUT GRADIENT
//                 [dsrc_0 ; dsrc_1 ; ...] = W^T dz, written straight into each source's gradient tensor.
//   rtk_pw_wgrad  dW[o][k] += sum over samples and positions of dz[o] x [src_0 ; src_1 ; ... ; 1][k]  (the optional ones row
//                 yields the bias gradient) -- MFMA over the position axis, workgroup partials added with float atomics.
//
// Register tiling (fp32 v_mfma_f32_16x16x4_f32, same as train_conv.hip): a wave owns 64 positions as FOUR interleaved
// 16-position tiles (tile t = positions 4j + t).  Lane (g, j) holds, per 16-channel block, x[t][r] = X[channel 4g + r][position
// 4j + t]: in a channel-major operand that is one float4 along the positions per r, in a point-major operand one float4 along
// the channels per t -- both layouts load and store 16 bytes per lane.  Weights are staged per (64 out, 64 in) tile in LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "fused_common.h"
#include "rtk_common.h"
#include "rtk_train.h"

namespace {

constexpr int PW_T = 256;
constexpr int PW_MAXOP = 4;

// 16-byte accesses that are only 4-byte aligned (odd position counts, channel offsets like the 3 xyz columns in front of a
// layer's feature columns): gfx950 global memory takes them as one dwordx4
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ f4 pw_ld4(const float *p) { const f4u v = *reinterpret_cast<const f4u *>(p); return (f4){v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ void pw_st4(float *p, f4 v) { *reinterpret_cast<f4u *>(p) = (f4u){v.x, v.y, v.z, v.w}; }

struct PwOp {             // one operand of a virtual concatenation
    const float *ptr;     // element (s, c, p):  channel-major ptr[s*sample_stride + c*pitch + p], point-major ptr[s*sample_stride + p*pitch + c]
    long sample_stride;
    int pitch;
    int channels;
    int layout;           // 0 channel-major, 1 point-major, 2 constant one (wgrad: the bias row)
    int col0;             // first column (input side) / row (output side) of this operand in W
};

struct PwParams {
    int samples, P, groups;
    int nsrc, ndst;
    PwOp src[PW_MAXOP], dst[PW_MAXOP];
    const float *W;
    int w_pitch, transpose_w;       // element (o, k) = transpose_w ? W[k*w_pitch + o] : W[o*w_pitch + k]
    const float *bias;              // indexed by the W row of the output channel, or NULL
    const float *rw;                // (samples, P) statistics weights or NULL
    double *sums;                   // (groups, stat_channels, 2) or NULL; indexed by the W row of the output channel
    int stat_channels;
    int nchunks;                    // output chunks of <= 64 channels: chunk -> (dst, first channel)
    unsigned char chunk_dst[96];
    short chunk_c0[96];
    int accumulate;                 // stores add to the destination (dgrad into a tensor that already holds a partial gradient)
};

__device__ __forceinline__ float pw_load1(const PwOp &op, const float *base, int c, int p, int P) {
    const int cc = min(c, op.channels - 1), pp = min(p, P - 1);          // unconditional load + select (see pw_load_block)
    const float v = op.layout ? base[(size_t)pp * op.pitch + cc] : base[(size_t)cc * op.pitch + pp];
    return (c < op.channels && p < P) ? v : 0.f;
}

// x[t][r] = X[channel c0 + 4g + r][position p + t] of one 16-channel block.  `fastp` (wave-uniform): the wave's 64 positions are in
// range and the operand is 16-byte aligned, so every access is a vector load guarded only by the lane's channel range.
__device__ __forceinline__ void pw_load_block(const PwOp &op, const float *base, int c0, int g, int p, int P, bool fastp, f4 (&x)[4]) {
    const int c = c0 + 4 * g;
    // The vector paths load unconditionally from a clamped (always valid) address and select afterwards: a conditional load is a
    // basic block of its own, and the compiler then waits for each load before issuing the next (measured: 2x on the whole kernel).
    if (fastp && op.layout && (op.channels & 3) == 0) {
        const int cc = min(c, op.channels - 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) x[t] = pw_ld4(base + (size_t)(p + t) * op.pitch + cc);
        if (c >= op.channels) {
#pragma unroll
            for (int t = 0; t < 4; ++t) x[t] = f4_zero();
        }
    } else if (fastp && !op.layout) {
        f4 q[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) q[r] = pw_ld4(base + (size_t)min(c + r, op.channels - 1) * op.pitch + p);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (c + r >= op.channels) q[r] = f4_zero();
#pragma unroll
        for (int t = 0; t < 4; ++t) x[t] = (f4){q[0][t], q[1][t], q[2][t], q[3][t]};
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) x[t][r] = pw_load1(op, base, c + r, p + t, P);
    }
}


// NW waves per workgroup (they share a staged weight tile), VB 16-channel output blocks per workgroup.  <4,4> for large batches;
// <1,1> for small ones: 16 times the workgroups, each a sixteenth of the serial work (at B = 1 the layers are pure latency)
template <int NW, int VB>
__global__ __launch_bounds__(64 * NW) void pw_conv_kernel(const PwParams Q) {
    constexpr int NT = 64 * NW;
    constexpr int OC = 16 * VB;      // output channels per workgroup
    __shared__ __attribute__((aligned(16))) f4 s_w[2 * 4 * VB * 64];    // double-buffered (16 VB out, 64 in) weight tile, fragment (u, v)
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15, wave = threadIdx.x >> 6;
    const int b = blockIdx.z, P = Q.P;
    const PwOp &D = Q.dst[Q.chunk_dst[blockIdx.y]];
    const int oc0 = Q.chunk_c0[blockIdx.y];                         // first channel of this chunk inside its destination
    const int orow0 = D.col0 + oc0;                                 // ... and its row of W
    const int nout = min(OC, D.channels - oc0);
    const int p0 = (blockIdx.x * NW + wave) * 64;
    const int p = p0 + 4 * j;
    const bool pfull = p0 + 64 <= P;
    f4 acc[4][VB];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int v = 0; v < VB; ++v) acc[t][v] = f4_zero();
    // K loop over the 64-channel tiles of all sources, software-pipelined: the weight tile and the input blocks of tile i + 1 are
    // requested (into registers) before the MFMAs of tile i are issued, so a tile costs max(MFMA time, memory latency) instead of
    // their sum -- with one wave per SIMD (a 256-position x 64-channel workgroup per CU is all these layers offer) nothing else
    // hides a round trip.  One barrier per tile (double-buffered LDS tile: a wave can only be one tile ahead).
    constexpr int NWL = 4 * VB * 64 / NT;                            // weight float4s staged per thread and tile
    int nk = 0;
    for (int s = 0; s < Q.nsrc; ++s) nk += (Q.src[s].channels + 63) >> 6;
    f4 wreg[NWL], xn[4][4];
    auto request = [&](int s, int k0) {
        const PwOp &S = Q.src[s];
        // unconditional loads from clamped addresses, zeros selected afterwards (see pw_load_block); the vector form when the whole
        // tile is inside the matrix (workgroup-uniform)
        if (!Q.transpose_w) {                // fragment (u, v), lane (fg, fi) = W[o = 16v + fi][k = 16u + 4fg .. +3]
            const bool kfull = k0 + 64 <= S.channels;
#pragma unroll
            for (int i = 0; i < NWL; ++i) {
                const int e = threadIdx.x + i * NT;
                const int f = e >> 6, l = e & 63, u = f / VB, v = f % VB, fg = l >> 4, fi = l & 15;
                const int o = 16 * v + fi, k = k0 + 16 * u + 4 * fg;
                const float *row = Q.W + (size_t)(orow0 + min(o, nout - 1)) * Q.w_pitch + S.col0;
                f4 w;
                if (kfull) w = pw_ld4(row + k);
                else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float t = row[min(k + q, S.channels - 1)];
                        w[q] = k + q < S.channels ? t : 0.f;
                    }
                }
                wreg[i] = o < nout ? w : f4_zero();
            }
        } else {                             // element (o, k) = W[k][o]: rows of W along o (contiguous)
            const bool ofull = nout == OC;
#pragma unroll
            for (int i = 0; i < NWL; ++i) {
                const int e = threadIdx.x + i * NT;
                const int kk = e / (4 * VB), o4 = (e % (4 * VB)) * 4, k = k0 + kk;
                const float *row = Q.W + (size_t)(S.col0 + min(k, S.channels - 1)) * Q.w_pitch + orow0;
                f4 w;
                if (ofull) w = pw_ld4(row + o4);
                else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float t = row[min(o4 + q, nout - 1)];
                        w[q] = o4 + q < nout ? t : 0.f;
                    }
                }
                wreg[i] = k < S.channels ? w : f4_zero();
            }
        }
        const float *sb = S.ptr + (size_t)b * S.sample_stride;
        const int nu = min(4, (S.channels - k0 + 15) >> 4);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (u < nu) pw_load_block(S, sb, k0 + 16 * u, g, p, P, pfull, xn[u]);
    };
    int buf = 0, s_cur = 0, k_cur = 0;
    request(0, 0);
    for (int it = 0; it < nk; ++it) {
        f4 *sw = s_w + buf * (4 * VB * 64);
        if (!Q.transpose_w) {
#pragma unroll
            for (int i = 0; i < NWL; ++i) sw[threadIdx.x + i * NT] = wreg[i];
        } else {                             // scatter the four values of each read into their fragments
            float *swf = reinterpret_cast<float *>(sw);
#pragma unroll
            for (int i = 0; i < NWL; ++i) {
                const int e = threadIdx.x + i * NT;
                const int kk = e / (4 * VB), o4 = (e % (4 * VB)) * 4;
                const int u = kk >> 4, fg = (kk >> 2) & 3, kq = kk & 3;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int o = o4 + q;
                    swf[(((u * VB + (o >> 4)) * 64) + fg * 16 + (o & 15)) * 4 + kq] = wreg[i][q];
                }
            }
        }
        f4 x[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) x[u][t] = xn[u][t];
        const int nu = min(4, (Q.src[s_cur].channels - k_cur + 15) >> 4);
        k_cur += 64;
        if (k_cur >= Q.src[s_cur].channels) { ++s_cur; k_cur = 0; }
        if (it + 1 < nk) request(s_cur, k_cur);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (u >= nu) break;
#pragma unroll
            for (int v = 0; v < VB; ++v) {
                if (16 * v >= nout) continue;                     // workgroup-uniform: narrow layers skip the empty blocks
                const f4 wf = sw[(u * VB + v) * 64 + lane];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t][v] = mfma4(wf[q], x[u][t][q], acc[t][v]);
            }
        }
        buf ^= 1;
    }
    // ---- epilogue: acc[t][v][r] = Z[channel oc0 + 16v + 4g + r][position p + t] -------------------------------------------------
    float wl[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) wl[t] = (p + t < P) ? (Q.rw ? Q.rw[(size_t)b * P + p + t] : 1.f) : 0.f;
    float *db = const_cast<float *>(D.ptr) + (size_t)b * D.sample_stride;
    const bool dfast = pfull;
    __shared__ double s_red[NW][OC][2];
    const bool stats = Q.sums != nullptr;
#pragma unroll
    for (int v = 0; v < VB; ++v) {
        if (16 * v >= nout) break;
        const bool cfull = 16 * v + 16 <= nout;
        f4 y[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) y[t] = acc[t][v];
        if (Q.bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * v + 4 * g + r;
                const float bv = c < nout ? Q.bias[orow0 + c] : 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) y[t][r] += bv;
            }
        }
        if (dfast && cfull && !Q.accumulate) {
            if (D.layout) {
#pragma unroll
                for (int t = 0; t < 4; ++t) pw_st4(db + (size_t)(p + t) * D.pitch + oc0 + 16 * v + 4 * g, y[t]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    pw_st4(db + (size_t)(oc0 + 16 * v + 4 * g + r) * D.pitch + p, (f4){y[0][r], y[1][r], y[2][r], y[3][r]});
            }
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 16 * v + 4 * g + r;
                    if (c < nout && p + t < P) {
                        float *o = D.layout ? db + (size_t)(p + t) * D.pitch + oc0 + c : db + (size_t)(oc0 + c) * D.pitch + p + t;
                        *o = Q.accumulate ? *o + y[t][r] : y[t][r];
                    }
                }
        }
        if (stats) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double a0 = 0.0, a1 = 0.0;
#pragma unroll
                for (int t = 0; t < 4; ++t) { a0 += (double)(wl[t] * y[t][r]); a1 += (double)(wl[t] * y[t][r] * y[t][r]); }
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) { a0 += __shfl_xor(a0, o, 64); a1 += __shfl_xor(a1, o, 64); }
                if (j == 0) { s_red[wave][16 * v + 4 * g + r][0] = a0; s_red[wave][16 * v + 4 * g + r][1] = a1; }
            }
        }
    }
    if (stats) {
        __syncthreads();
        if (threadIdx.x < nout) {
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) { a0 += s_red[w][threadIdx.x][0]; a1 += s_red[w][threadIdx.x][1]; }
            const int grp = b / (Q.samples / Q.groups);
            const size_t o2 = ((size_t)grp * Q.stat_channels + orow0 + threadIdx.x) * 2;
            rtk_stat_add<RTK_STAT_FORWARD>(Q.sums, (size_t)Q.groups * Q.stat_channels * 2, b, o2, a0);
            rtk_stat_add<RTK_STAT_FORWARD>(Q.sums, (size_t)Q.groups * Q.stat_channels * 2, b, o2 + 1, a1);
        }
    }
}

// ---- weight gradient ----------------------------------------------------------------------------------------------------
// Operand fragments of one 64-channel group over a 16-position tile: val[blk][s], k-step s of lane group g = position pt + 4g + s.
//   channel-major: blk = 16-channel block, lane row i <-> channel 16 blk + i   (one float4 along the positions per block)
//   point-major:   blk = channel residue,  lane row i <-> channel 4 i + blk     (one float4 along the channels per k-step)
__device__ __forceinline__ int pw_frag_channel(int layout, int blk, int i) { return layout == 1 ? 4 * i + blk : 16 * blk + i; }

__device__ __forceinline__ void pw_load_frags(const PwOp &op, const float *base, int c0, int i, int pq, int P, bool fastp, f4 (&val)[4]) {
    if (op.layout == 2) {           // the constant-one row: channel 0 only
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
            for (int s = 0; s < 4; ++s) val[blk][s] = (c0 + pw_frag_channel(0, blk, i) == 0 && pq + s < P) ? 1.f : 0.f;
        return;
    }
    // vector paths: unconditional loads from clamped addresses, zeros selected afterwards (see pw_load_block)
    if (fastp && op.layout == 1 && (op.channels & 3) == 0) {
        const int cc = min(c0 + 4 * i, op.channels - 4);
        f4 q[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) q[s] = pw_ld4(base + (size_t)(pq + s) * op.pitch + cc);
        const bool ok = c0 + 4 * i < op.channels;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) val[blk] = ok ? (f4){q[0][blk], q[1][blk], q[2][blk], q[3][blk]} : f4_zero();
    } else if (fastp && op.layout == 0) {
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) val[blk] = pw_ld4(base + (size_t)min(c0 + 16 * blk + i, op.channels - 1) * op.pitch + pq);
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
            if (c0 + 16 * blk + i >= op.channels) val[blk] = f4_zero();
    } else {
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
            for (int s = 0; s < 4; ++s) val[blk][s] = pw_load1(op, base, c0 + pw_frag_channel(op.layout, blk, i), pq + s, P);
    }
}

struct PwWgParams {
    int samples, P;
    PwOp dz;                        // (samples, cout, P)
    int nsrc;
    PwOp src[PW_MAXOP + 1];         // col0 = first column of dW; layout 2 = the bias row (its gradient goes to dbias)
    float *dW; int w_pitch;
    float *dbias;
    int nchunks;                    // input chunks of <= 64 channels: chunk -> (src, first channel)
    unsigned char chunk_src[40];
    short chunk_c0[40];
    int tiles_per_wg;               // (sample, 16-position tile) pairs per workgroup
    int ochunks, splits;            // the job's grid: nchunks x ochunks x splits workgroups
    float *partial;                 // splits > 1: [z][block = y * nchunks + x][64 x 64] workgroup partials (pw_wgrad_reduce_kernel adds them)
};

// Several weight gradients in ONE launch (up to PW_WG_JOBS; the parameters of all of them travel in the kernel argument): the backward
// of a step has 35 of them, every one a leaf of the graph (only the optimizer reads it), each a 10-20 us launch that fills the chip
// for a few microseconds.  Deferred to the end of the backward and launched together they are one grid of ~1000 workgroups.
constexpr int PW_WG_JOBS = 8;
struct PwWgMulti {
    int n;
    int first_block[PW_WG_JOBS + 1];      // flat workgroup index -> job
    PwWgParams q[PW_WG_JOBS];
};
static_assert(sizeof(PwWgMulti) <= 4096, "kernel arguments are limited to 4 KiB");

__global__ __launch_bounds__(PW_T, 2) void pw_wgrad_kernel(const PwWgMulti M) {
    __shared__ float s_red[PW_T / 64][64][65];      // one image per wave: written in parallel, added on the way out
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15, wave = threadIdx.x >> 6;
    int ji = 0;
    while (ji + 1 < M.n && (int)blockIdx.x >= M.first_block[ji + 1]) ++ji;
    const PwWgParams &Q = M.q[ji];
    // the job's own (input chunk, output chunk, position split) grid
    const int flat = (int)blockIdx.x - M.first_block[ji];
    const int gx = Q.nchunks, gy = Q.ochunks;
    const int bx = flat % gx, by = (flat / gx) % gy, bz = flat / (gx * gy);
    const PwOp &S = Q.src[Q.chunk_src[bx]];
    const int k0 = Q.chunk_c0[bx];
    const int o0 = by * 64;
    const int P = Q.P, tps = (P + 15) >> 4;                          // tiles per sample
    const long ntiles = (long)Q.samples * tps;
    const long t_begin = (long)bz * Q.tiles_per_wg, t_end = min(ntiles, t_begin + Q.tiles_per_wg);
    f4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = f4_zero();
    const float *sbase = S.layout == 2 ? nullptr : S.ptr;
    auto fetch = [&](long t, f4 (&A)[4], f4 (&B)[4]) {
        const int b = (int)(t / tps), pt = (int)(t - (long)b * tps) * 16;
        const int pq = pt + 4 * g;
        const bool pfull = pt + 16 <= P;
        pw_load_frags(Q.dz, Q.dz.ptr + (size_t)b * Q.dz.sample_stride, o0, j, pq, P, pfull, A);
        pw_load_frags(S, sbase ? sbase + (size_t)b * S.sample_stride : nullptr, k0, j, pq, P, pfull, B);
    };
    // operands of the next two tiles are in flight during a tile's 64 MFMAs (a tile is 0.85 us of matrix work, a round trip to HBM more)
    constexpr int STEP = PW_T / 64;
    f4 A[4], B[4], A1[4], B1[4];
    long t = t_begin + wave;
    if (t < t_end) fetch(t, A, B);
    if (t + STEP < t_end) fetch(t + STEP, A1, B1);
    for (; t < t_end; t += STEP) {
        f4 A2[4], B2[4];
        if (t + 2 * STEP < t_end) fetch(t + 2 * STEP, A2, B2);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[a][c] = mfma4(A[a][q], B[c][q], acc[a][c]);
#pragma unroll
        for (int q = 0; q < 4; ++q) { A[q] = A1[q]; B[q] = B1[q]; A1[q] = A2[q]; B1[q] = B2[q]; }
    }
    // D layout: acc[a][c][r] = dW[o0 + chanA(a, 4g + r)][k0 + chanB(c, j)]; every wave stores its own LDS image (one round, one
    // barrier; adding into a single image in turn was four), the images are added on the way out
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                s_red[wave][pw_frag_channel(Q.dz.layout, a, 4 * g + r)][pw_frag_channel(S.layout, c, j)] = acc[a][c][r];
    __syncthreads();
    auto total = [&](int o, int k) { return (s_red[0][o][k] + s_red[1][o][k]) + (s_red[2][o][k] + s_red[3][o][k]); };
    static_assert(PW_T == 256, "four waves");
    if (Q.splits > 1) {              // a partial block per workgroup, no atomics (a 64 x 64 block of float atomics per workgroup on
                                     // 64..256 contended addresses cost more than the whole product)
        float *dst = Q.partial + ((size_t)bz * gy * gx + (size_t)by * gx + bx) * 4096;
        for (int e = threadIdx.x; e < 64 * 64; e += PW_T) dst[e] = total(e >> 6, e & 63);
        return;
    }
    const int no = min(64, Q.dz.channels - o0), nk = min(64, S.channels - k0);
    for (int e = threadIdx.x; e < 64 * 64; e += PW_T) {
        const int o = e >> 6, k = e & 63;
        if (o < no && k < nk) {
            float *dst = S.layout == 2 ? Q.dbias + o0 + o : Q.dW + (size_t)(o0 + o) * Q.w_pitch + S.col0 + k0 + k;
            *dst += total(o, k);                                   // this workgroup owns the block
        }
    }
}

// dW block (y, x) += sum over z of the workgroup partials.  A workgroup = 32 consecutive elements x 8 lanes over z (every load
// independent: the sum over a few hundred partials is a latency problem, not a bandwidth one), then a fixed-order LDS reduction.
__global__ __launch_bounds__(256) void pw_wgrad_reduce_kernel(const PwWgMulti M) {      // first_block: in units of 128 workgroups (one 64 x 64 block)
    __shared__ float s_part[8][33];
    const int el = threadIdx.x & 31, zl = threadIdx.x >> 5;
    int ji = 0;
    while (ji + 1 < M.n && (int)(blockIdx.x >> 7) >= M.first_block[ji + 1]) ++ji;
    const PwWgParams &Q = M.q[ji];
    const int splits = Q.splits, ochunks = Q.ochunks;
    const int blk = (int)(blockIdx.x >> 7) - M.first_block[ji], e = (blockIdx.x & 127) * 32 + el;              // 128 workgroups per 64 x 64 block
    const float *src = Q.partial + (size_t)blk * 4096 + e;
    const size_t stride = (size_t)ochunks * Q.nchunks * 4096;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int z = zl;
    for (; z + 24 < splits; z += 32) {
        a0 += src[(size_t)z * stride]; a1 += src[(size_t)(z + 8) * stride]; a2 += src[(size_t)(z + 16) * stride]; a3 += src[(size_t)(z + 24) * stride];
    }
    for (; z < splits; z += 8) a0 += src[(size_t)z * stride];
    s_part[zl][el] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (zl) return;
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) sum += s_part[q][el];
    const int x = blk % Q.nchunks, y = blk / Q.nchunks;
    const PwOp &S = Q.src[Q.chunk_src[x]];
    const int k0 = Q.chunk_c0[x], o0 = y * 64, o = e >> 6, k = e & 63;
    if (o >= Q.dz.channels - o0 || k >= S.channels - k0) return;
    float *dst = S.layout == 2 ? Q.dbias + o0 + o : Q.dW + (size_t)(o0 + o) * Q.w_pitch + S.col0 + k0 + k;
    *dst += sum;
}

// ---- weight images (rtk_pack_weights) ---------------------------------------------------------------------------------------
constexpr int PK_MAX = 16;
struct PkJobs { int n; rtk_pack_job_t j[PK_MAX]; int first_block[PK_MAX + 1]; };

__global__ __launch_bounds__(256) void pack_weights_kernel(const PkJobs Q) {
    int ji = 0;
    while (ji + 1 < Q.n && (int)blockIdx.x >= Q.first_block[ji + 1]) ++ji;
    const rtk_pack_job_t &J = Q.j[ji];
    const int e = ((int)blockIdx.x - Q.first_block[ji]) * 256 + threadIdx.x;
    auto at = [&](int o, int k) -> float {
        if (o >= J.rows || k >= J.cols) return 0.f;
        return J.transpose ? J.src[(size_t)k * J.pitch + o] : J.src[(size_t)o * J.pitch + k];
    };
    if (J.kind == 0) {
        const int V = (J.rows + 15) >> 4, U = (J.cols + 15) >> 4;
        if (e >= U * V * 256) return;
        const int r = e & 3, i = (e >> 2) & 15, g = (e >> 6) & 3, uv = e >> 8, v = uv % V, u = uv / V;
        J.dst[e] = at(16 * v + i, 16 * u + 4 * g + r);
    } else if (J.kind == 1) {
        const int V = (J.rows + 15) >> 4;
        if (e >= V * 64) return;
        const int i = e & 15, g = (e >> 4) & 3, v = e >> 6, o = 16 * v + i;
        J.dst[e] = g < 3 ? at(o, g) : ((J.src2 && o < J.rows) ? J.src2[o] : 0.f);
    } else {
        const int n16 = (J.rows + 15) & ~15;
        if (e >= n16) return;
        J.dst[e] = e < J.rows ? J.src[e] : 0.f;
    }
}

int fill_op(PwOp &d, const rtk_pw_operand_t &s, const char *who) {
    if (!(s.layout == 2 || s.ptr) || s.channels <= 0 || s.layout < 0 || s.layout > 2) {
        rtk_set_error("%s: bad operand (channels %d, layout %d)", who, s.channels, s.layout);
        return RTK_ERR_INVALID;
    }
    d.ptr = s.ptr; d.sample_stride = s.sample_stride; d.pitch = s.pitch; d.channels = s.channels; d.layout = s.layout; d.col0 = s.col0;
    return RTK_OK;
}

}  // namespace

extern "C" int rtk_pw_conv(int samples, int positions, int nsrc, const rtk_pw_operand_t *srcs, int ndst, const rtk_pw_operand_t *dsts,
                           const float *w, int w_pitch, int transpose_w, const float *bias, int accumulate, const float *row_weight,
                           int groups, double *sums, int stat_channels, rtk_stream_t stream) {
    RTK_REQUIRE(samples > 0 && positions > 0 && nsrc >= 1 && nsrc <= PW_MAXOP && ndst >= 1 && ndst <= PW_MAXOP && srcs && dsts && w,
                "pw_conv: bad arguments");
    RTK_REQUIRE(samples <= 65535 && groups >= 1 && samples % groups == 0, "pw_conv: bad batch (%d samples, %d groups)", samples, groups);
    PwParams Q = {};
    Q.samples = samples; Q.P = positions; Q.groups = groups; Q.nsrc = nsrc; Q.ndst = ndst;
    for (int i = 0; i < nsrc; ++i) {
        if (fill_op(Q.src[i], srcs[i], "pw_conv")) return RTK_ERR_INVALID;
        RTK_REQUIRE(srcs[i].layout != 2, "pw_conv: the constant-one operand is a wgrad source");
    }
    // four waves sharing one staged 64-channel weight tile -- unless that leaves most of the chip idle (small batches): then one
    // wave and one 16-channel block per workgroup
    long big_wgs = 0;
    for (int i = 0; i < ndst; ++i) big_wgs += (long)rtk_divup(dsts[i].channels, 64) * rtk_divup(positions, 256) * samples;
    const bool small = big_wgs < 192;
    const int vb = small ? 1 : 4;
    const int oc = 16 * vb;
    int nch = 0;
    for (int i = 0; i < ndst; ++i) {
        if (fill_op(Q.dst[i], dsts[i], "pw_conv")) return RTK_ERR_INVALID;
        RTK_REQUIRE(dsts[i].layout != 2, "pw_conv: bad destination layout");
        for (int c0 = 0; c0 < dsts[i].channels; c0 += oc) {
            RTK_REQUIRE(nch < 96, "pw_conv: more than 96 output chunks");
            Q.chunk_dst[nch] = (unsigned char)i;
            Q.chunk_c0[nch++] = (short)c0;
        }
    }
    Q.nchunks = nch;
    Q.W = w; Q.w_pitch = w_pitch; Q.transpose_w = transpose_w; Q.bias = bias; Q.rw = row_weight; Q.sums = sums;
    Q.stat_channels = stat_channels; Q.accumulate = accumulate;
    if (!small && vb == 4) pw_conv_kernel<4, 4><<<dim3(rtk_divup(positions, 256), nch, samples), 256, 0, (hipStream_t)stream>>>(Q);
    else if (!small && vb == 2) pw_conv_kernel<4, 2><<<dim3(rtk_divup(positions, 256), nch, samples), 256, 0, (hipStream_t)stream>>>(Q);
    else if (!small) pw_conv_kernel<4, 1><<<dim3(rtk_divup(positions, 256), nch, samples), 256, 0, (hipStream_t)stream>>>(Q);
    else pw_conv_kernel<1, 1><<<dim3(rtk_divup(positions, 64), nch, samples), 64, 0, (hipStream_t)stream>>>(Q);
    RTK_CHECK_LAUNCH("pw_conv");
    return RTK_OK;
}

// one job's kernel parameters; want_wgs = workgroups to aim for (position splits), ws = its share of the partial workspace
static int wgrad_job(PwWgParams &Q, const rtk_pw_wgrad_job_t &J, int want_wgs, float *ws, long ws_floats) {
    RTK_REQUIRE(J.samples > 0 && J.positions > 0 && J.dz && J.nsrc >= 1 && J.nsrc <= PW_MAXOP && J.srcs && J.dw, "pw_wgrad: bad arguments");
    Q = PwWgParams{};
    Q.samples = J.samples; Q.P = J.positions; Q.nsrc = J.nsrc; Q.dW = J.dw; Q.w_pitch = J.w_pitch; Q.dbias = J.dbias;
    if (fill_op(Q.dz, *J.dz, "pw_wgrad")) return RTK_ERR_INVALID;
    RTK_REQUIRE(J.dz->layout != 2, "pw_wgrad: bad dz layout");
    int nch = 0;
    for (int i = 0; i < J.nsrc + (J.dbias ? 1 : 0); ++i) {
        if (i < J.nsrc) {
            if (fill_op(Q.src[i], J.srcs[i], "pw_wgrad")) return RTK_ERR_INVALID;
            RTK_REQUIRE(J.srcs[i].layout != 2, "pw_wgrad: pass dbias instead of a constant-one source");
        } else {
            Q.src[i] = PwOp{nullptr, 0, 0, 1, 2, 0};
        }
        for (int c0 = 0; c0 < Q.src[i].channels; c0 += 64) {
            RTK_REQUIRE(nch < 40, "pw_wgrad: more than 40 input chunks");
            Q.chunk_src[nch] = (unsigned char)i;
            Q.chunk_c0[nch++] = (short)c0;
        }
    }
    Q.nchunks = nch;
    const long ntiles = (long)J.samples * ((J.positions + 15) / 16);
    Q.ochunks = rtk_divup(J.dz->channels, 64);
    // position splits: towards want_wgs workgroups (more only adds partial blocks: tools/experiments/exp_pw.py), at least one tile per
    // wave, and no more than the workspace holds partial blocks for
    const long blocks = (long)nch * Q.ochunks;
    long splits = (want_wgs + blocks - 1) / blocks;
    if (splits > (ntiles + 3) / 4) splits = (ntiles + 3) / 4;      // ... and at least one tile per wave (tiny batches: the serial depth counts)
    if (splits > ws_floats / (blocks * 4096)) splits = ws ? ws_floats / (blocks * 4096) : 1;
    if (splits < 1) splits = 1;
    Q.tiles_per_wg = (int)((ntiles + splits - 1) / splits);
    Q.splits = (int)((ntiles + Q.tiles_per_wg - 1) / Q.tiles_per_wg);
    Q.partial = ws;
    return RTK_OK;
}

extern "C" int rtk_pw_wgrad_multi(int njobs, const rtk_pw_wgrad_job_t *jobs, float *workspace, long workspace_floats, rtk_stream_t stream) {
    RTK_REQUIRE(njobs >= 1 && jobs, "pw_wgrad_multi: bad arguments");
    for (int j0 = 0; j0 < njobs; j0 += PW_WG_JOBS) {
        const int n = njobs - j0 < PW_WG_JOBS ? njobs - j0 : PW_WG_JOBS;
        PwWgMulti M = {}, R = {};
        M.n = R.n = n;
        // about four workgroups per CU for the launch as a whole, shared out in proportion to the jobs' work (64 x 64 blocks x position
        // tiles): with equal shares the launch lasts as long as its largest job on an eighth of the chip.  The workspace in equal shares.
        double work[PW_WG_JOBS], total = 0.0;
        for (int k = 0; k < n; ++k) {
            const rtk_pw_wgrad_job_t &J = jobs[j0 + k];
            RTK_REQUIRE(J.dz && J.srcs && J.nsrc >= 1 && J.nsrc <= PW_MAXOP, "pw_wgrad: bad arguments");
            long cin = J.dbias ? 1 : 0;
            for (int i = 0; i < J.nsrc; ++i) cin += (J.srcs[i].channels + 63) / 64;
            work[k] = (double)cin * ((J.dz->channels + 63) / 64) * J.samples * ((J.positions + 15) / 16);
            total += work[k];
        }
        const long share = workspace ? (workspace_floats / n) & ~4095L : 0;
        int wgs = 0, rblocks = 0, nred = 0;
        for (int k = 0; k < n; ++k) {
            const int want = n == 1 ? 512 : (int)(1024.0 * work[k] / total) + 1;
            if (int rc = wgrad_job(M.q[k], jobs[j0 + k], want, workspace ? workspace + (size_t)k * share : nullptr, share)) return rc;
            M.first_block[k] = wgs;
            wgs += M.q[k].nchunks * M.q[k].ochunks * M.q[k].splits;
            if (M.q[k].splits > 1) {
                R.q[nred] = M.q[k];
                R.first_block[nred++] = rblocks;
                rblocks += M.q[k].nchunks * M.q[k].ochunks;
            }
        }
        M.first_block[n] = wgs;
        pw_wgrad_kernel<<<wgs, PW_T, 0, (hipStream_t)stream>>>(M);
        RTK_CHECK_LAUNCH("pw_wgrad");
        if (nred) {
            R.n = nred;
            R.first_block[nred] = rblocks;
            pw_wgrad_reduce_kernel<<<(unsigned)(rblocks * 128), 256, 0, (hipStream_t)stream>>>(R);
            RTK_CHECK_LAUNCH("pw_wgrad_reduce");
        }
    }
    return RTK_OK;
}

extern "C" int rtk_pw_wgrad(int samples, int positions, const rtk_pw_operand_t *dz, int nsrc, const rtk_pw_operand_t *srcs, float *dw,
                            int w_pitch, float *dbias, float *workspace, long workspace_floats, rtk_stream_t stream) {
    const rtk_pw_wgrad_job_t job = {samples, positions, dz, nsrc, srcs, dw, w_pitch, dbias};
    return rtk_pw_wgrad_multi(1, &job, workspace, workspace_floats, stream);
}

extern "C" int rtk_pack_weights(int njobs, const rtk_pack_job_t *jobs, rtk_stream_t stream) {
    RTK_REQUIRE(njobs >= 1 && njobs <= PK_MAX && jobs, "pack_weights: 1..%d jobs", PK_MAX);
    PkJobs Q = {};
    Q.n = njobs;
    int blocks = 0;
    for (int i = 0; i < njobs; ++i) {
        const rtk_pack_job_t &J = jobs[i];
        RTK_REQUIRE(J.src && J.dst && J.rows > 0 && (J.kind == 2 || J.cols > 0) && J.kind >= 0 && J.kind <= 2, "pack_weights: bad job %d", i);
        Q.j[i] = J;
        Q.first_block[i] = blocks;
        const long elems = J.kind == 0 ? (long)((J.rows + 15) / 16) * ((J.cols + 15) / 16) * 256 : J.kind == 1 ? (long)((J.rows + 15) / 16) * 64
                                                                                                          : (long)((J.rows + 15) & ~15);
        blocks += (int)((elems + 255) / 256);
    }
    Q.first_block[njobs] = blocks;
    pack_weights_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(Q);
    RTK_CHECK_LAUNCH("pack_weights");
    return RTK_OK;
}
