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
    unsigned char chunk_dst[40];
    short chunk_c0[40];
    int accumulate;                 // stores add to the destination (dgrad into a tensor that already holds a partial gradient)
};

__device__ __forceinline__ float pw_load1(const PwOp &op, const float *base, int c, int p, int P) {
    if (c >= op.channels || p >= P) return 0.f;
    return op.layout ? base[(size_t)p * op.pitch + c] : base[(size_t)c * op.pitch + p];
}

// x[t][r] = X[channel c0 + 4g + r][position p + t] of one 16-channel block.  `fastp` (wave-uniform): the wave's 64 positions are in
// range and the operand is 16-byte aligned, so every access is a vector load guarded only by the lane's channel range.
__device__ __forceinline__ void pw_load_block(const PwOp &op, const float *base, int c0, int g, int p, int P, bool fastp, f4 (&x)[4]) {
    const int c = c0 + 4 * g;
    if (fastp && op.layout && c + 4 <= op.channels) {
#pragma unroll
        for (int t = 0; t < 4; ++t) x[t] = pw_ld4(base + (size_t)(p + t) * op.pitch + c);
    } else if (fastp && !op.layout) {
        f4 q[4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
            q[r] = c + r < op.channels ? pw_ld4(base + (size_t)(c + r) * op.pitch + p) : f4_zero();
#pragma unroll
        for (int t = 0; t < 4; ++t) x[t] = (f4){q[0][t], q[1][t], q[2][t], q[3][t]};
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) x[t][r] = pw_load1(op, base, c + r, p + t, P);
    }
}


template <int NW>      // waves per workgroup: 4 (they share a weight tile) or 1 (small batches: four times the workgroups)
__global__ __launch_bounds__(64 * NW) void pw_conv_kernel(const PwParams Q) {
    constexpr int NT = 64 * NW;
    __shared__ __attribute__((aligned(16))) f4 s_w[2 * 16 * 64];    // double-buffered (64 out, 64 in) weight tile, fragment (u, v)
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15, wave = threadIdx.x >> 6;
    const int b = blockIdx.z, P = Q.P;
    const PwOp &D = Q.dst[Q.chunk_dst[blockIdx.y]];
    const int oc0 = Q.chunk_c0[blockIdx.y];                         // first channel of this chunk inside its destination
    const int orow0 = D.col0 + oc0;                                 // ... and its row of W
    const int nout = min(64, D.channels - oc0);
    const int p0 = (blockIdx.x * NW + wave) * 64;
    const int p = p0 + 4 * j;
    const bool pfull = p0 + 64 <= P;
    f4 acc[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[t][v] = f4_zero();
    int buf = 0;
    for (int s = 0; s < Q.nsrc; ++s) {
        const PwOp &S = Q.src[s];
        const float *sb = S.ptr + (size_t)b * S.sample_stride;
        for (int k0 = 0; k0 < S.channels; k0 += 64) {
            // stage the (64 out, 64 in) weight tile as MFMA A fragments into the other half of the double buffer: fragment (u, v),
            // lane (fg, fi) = W[o = 16v + fi][k = 16u + 4fg .. +3].  One barrier per tile: a wave can only be one tile ahead.
            f4 *sw = s_w + buf * (16 * 64);
            if (!Q.transpose_w) {
                for (int e = threadIdx.x; e < 16 * 64; e += NT) {
                    const int f = e >> 6, l = e & 63, u = f >> 2, v = f & 3, fg = l >> 4, fi = l & 15;
                    const int o = 16 * v + fi, k = k0 + 16 * u + 4 * fg;
                    f4 w = f4_zero();
                    if (o < nout) {
                        const float *src = Q.W + (size_t)(orow0 + o) * Q.w_pitch + S.col0 + k;
                        if (k + 4 <= S.channels) w = pw_ld4(src);
                        else {
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                if (k + q < S.channels) w[q] = src[q];
                        }
                    }
                    sw[e] = w;
                }
            } else {      // element (o, k) = W[k][o]: read rows of W along o (contiguous), scatter the four values into their fragments
                float *swf = reinterpret_cast<float *>(sw);
                for (int e = threadIdx.x; e < 64 * 16; e += NT) {
                    const int kk = e >> 4, o4 = (e & 15) * 4, k = k0 + kk;
                    f4 w = f4_zero();
                    if (k < S.channels) {
                        const float *src = Q.W + (size_t)(S.col0 + k) * Q.w_pitch + orow0 + o4;
                        if (o4 + 4 <= nout) w = pw_ld4(src);
                        else {
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                if (o4 + q < nout) w[q] = src[q];
                        }
                    }
                    const int u = kk >> 4, fg = (kk >> 2) & 3, kq = kk & 3;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int o = o4 + q;
                        swf[(((u * 4 + (o >> 4)) * 64) + fg * 16 + (o & 15)) * 4 + kq] = w[q];
                    }
                }
            }
            // the tile's four input blocks are requested up front (all in flight across the barrier and the first MFMAs)
            const int nu = min(4, (S.channels - k0 + 15) >> 4);
            f4 x[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (u < nu) pw_load_block(S, sb, k0 + 16 * u, g, p, P, pfull, x[u]);
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (u >= nu) break;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    if (16 * v >= nout) continue;                     // workgroup-uniform: narrow layers skip the empty blocks
                    const f4 wf = sw[(u * 4 + v) * 64 + lane];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        acc[t][v] = mfma4(wf.x, x[u][t].x, acc[t][v]);
                        acc[t][v] = mfma4(wf.y, x[u][t].y, acc[t][v]);
                        acc[t][v] = mfma4(wf.z, x[u][t].z, acc[t][v]);
                        acc[t][v] = mfma4(wf.w, x[u][t].w, acc[t][v]);
                    }
                }
            }
            buf ^= 1;
        }
    }
    // ---- epilogue: acc[t][v][r] = Z[channel oc0 + 16v + 4g + r][position p + t] -------------------------------------------------
    float wl[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) wl[t] = (p + t < P) ? (Q.rw ? Q.rw[(size_t)b * P + p + t] : 1.f) : 0.f;
    float *db = const_cast<float *>(D.ptr) + (size_t)b * D.sample_stride;
    const bool dfast = pfull;
    __shared__ double s_red[NW][64][2];
    const bool stats = Q.sums != nullptr;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
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
            double *dst = Q.sums + ((size_t)grp * Q.stat_channels + orow0 + threadIdx.x) * 2;
            atomicAdd(dst, a0);
            atomicAdd(dst + 1, a1);
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
    if (fastp && op.layout == 1 && c0 + 4 * i + 4 <= op.channels) {
        f4 q[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) q[s] = pw_ld4(base + (size_t)(pq + s) * op.pitch + c0 + 4 * i);
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) val[blk] = (f4){q[0][blk], q[1][blk], q[2][blk], q[3][blk]};
    } else if (fastp && op.layout == 0) {
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
            val[blk] = c0 + 16 * blk + i < op.channels ? pw_ld4(base + (size_t)(c0 + 16 * blk + i) * op.pitch + pq) : f4_zero();
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
};

__global__ __launch_bounds__(PW_T, 2) void pw_wgrad_kernel(const PwWgParams Q) {
    __shared__ float s_red[64][65];
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15, wave = threadIdx.x >> 6;
    const PwOp &S = Q.src[Q.chunk_src[blockIdx.x]];
    const int k0 = Q.chunk_c0[blockIdx.x];
    const int o0 = blockIdx.y * 64;
    const int P = Q.P, tps = (P + 15) >> 4;                          // tiles per sample
    const long ntiles = (long)Q.samples * tps;
    const long t_begin = (long)blockIdx.z * Q.tiles_per_wg, t_end = min(ntiles, t_begin + Q.tiles_per_wg);
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
    f4 A[4], B[4];
    long t = t_begin + wave;
    if (t < t_end) fetch(t, A, B);
    for (; t < t_end; t += PW_T / 64) {
        f4 An[4], Bn[4];
        const bool more = t + PW_T / 64 < t_end;
        if (more) fetch(t + PW_T / 64, An, Bn);       // the next tile's operands are in flight during this tile's 64 MFMAs
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc[a][c] = mfma4(A[a].x, B[c].x, acc[a][c]);
                acc[a][c] = mfma4(A[a].y, B[c].y, acc[a][c]);
                acc[a][c] = mfma4(A[a].z, B[c].z, acc[a][c]);
                acc[a][c] = mfma4(A[a].w, B[c].w, acc[a][c]);
            }
        if (more) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { A[q] = An[q]; B[q] = Bn[q]; }
        }
    }
    // D layout: acc[a][c][r] = dW[o0 + chanA(a, 4g + r)][k0 + chanB(c, j)]; the four waves add into one LDS image in turn
    for (int w = 0; w < PW_T / 64; ++w) {
        if (wave == w) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float &d = s_red[pw_frag_channel(Q.dz.layout, a, 4 * g + r)][pw_frag_channel(S.layout, c, j)];
                        d = w == 0 ? acc[a][c][r] : d + acc[a][c][r];
                    }
        }
        __syncthreads();
    }
    const int no = min(64, Q.dz.channels - o0), nk = min(64, S.channels - k0);
    for (int e = threadIdx.x; e < 64 * 64; e += PW_T) {
        const int o = e >> 6, k = e & 63;
        if (o < no && k < nk) {
            float *dst = S.layout == 2 ? Q.dbias + o0 + o : Q.dW + (size_t)(o0 + o) * Q.w_pitch + S.col0 + k0 + k;
            atomicAdd(dst, s_red[o][k]);
        }
    }
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
    int nch = 0;
    for (int i = 0; i < ndst; ++i) {
        if (fill_op(Q.dst[i], dsts[i], "pw_conv")) return RTK_ERR_INVALID;
        RTK_REQUIRE(dsts[i].layout != 2, "pw_conv: bad destination layout");
        for (int c0 = 0; c0 < dsts[i].channels; c0 += 64) {
            RTK_REQUIRE(nch < 40, "pw_conv: more than 40 output chunks");
            Q.chunk_dst[nch] = (unsigned char)i;
            Q.chunk_c0[nch++] = (short)c0;
        }
    }
    Q.nchunks = nch;
    Q.W = w; Q.w_pitch = w_pitch; Q.transpose_w = transpose_w; Q.bias = bias; Q.rw = row_weight; Q.sums = sums;
    Q.stat_channels = stat_channels; Q.accumulate = accumulate;
    // four waves sharing one staged weight tile -- unless that leaves most of the chip idle (small batches): then one wave each
    static const int force_nw = getenv("RTK_PW_NW") ? atoi(getenv("RTK_PW_NW")) : 0;      // experiment knob (tools/exp_pw.py)
    if (force_nw == 2) {
        pw_conv_kernel<2><<<dim3(rtk_divup(positions, 128), nch, samples), 128, 0, (hipStream_t)stream>>>(Q);
    } else if (force_nw != 1 && (force_nw == 4 || (long)rtk_divup(positions, 256) * nch * samples >= 192)) {
        pw_conv_kernel<4><<<dim3(rtk_divup(positions, 256), nch, samples), 256, 0, (hipStream_t)stream>>>(Q);
    } else {
        pw_conv_kernel<1><<<dim3(rtk_divup(positions, 64), nch, samples), 64, 0, (hipStream_t)stream>>>(Q);
    }
    RTK_CHECK_LAUNCH("pw_conv");
    return RTK_OK;
}

extern "C" int rtk_pw_wgrad(int samples, int positions, const rtk_pw_operand_t *dz, int nsrc, const rtk_pw_operand_t *srcs, float *dw,
                            int w_pitch, float *dbias, rtk_stream_t stream) {
    RTK_REQUIRE(samples > 0 && positions > 0 && dz && nsrc >= 1 && nsrc <= PW_MAXOP && srcs && dw, "pw_wgrad: bad arguments");
    PwWgParams Q = {};
    Q.samples = samples; Q.P = positions; Q.nsrc = nsrc; Q.dW = dw; Q.w_pitch = w_pitch; Q.dbias = dbias;
    if (fill_op(Q.dz, *dz, "pw_wgrad")) return RTK_ERR_INVALID;
    RTK_REQUIRE(dz->layout != 2, "pw_wgrad: bad dz layout");
    int nch = 0;
    for (int i = 0; i < nsrc + (dbias ? 1 : 0); ++i) {
        if (i < nsrc) {
            if (fill_op(Q.src[i], srcs[i], "pw_wgrad")) return RTK_ERR_INVALID;
            RTK_REQUIRE(srcs[i].layout != 2, "pw_wgrad: pass dbias instead of a constant-one source");
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
    const long ntiles = (long)samples * ((positions + 15) / 16);
    const int ochunks = rtk_divup(dz->channels, 64);
    // enough workgroups to fill the chip, few enough that the 64 x 64 atomics per workgroup stay a small share of its work
    // about one workgroup per CU; at least one tile per wave behind every 64 x 64 block of atomics
    long splits = (256 + (long)nch * ochunks - 1) / ((long)nch * ochunks);
    if (splits > (ntiles + 3) / 4) splits = (ntiles + 3) / 4;
    if (splits < 1) splits = 1;
    Q.tiles_per_wg = (int)((ntiles + splits - 1) / splits);
    const dim3 grid(nch, ochunks, (unsigned)((ntiles + Q.tiles_per_wg - 1) / Q.tiles_per_wg));
    pw_wgrad_kernel<<<grid, PW_T, 0, (hipStream_t)stream>>>(Q);
    RTK_CHECK_LAUNCH("pw_wgrad");
    return RTK_OK;
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
