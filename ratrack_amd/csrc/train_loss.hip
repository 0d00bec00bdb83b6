// Multi-task loss of the backbone trainer and its gradients in ONE launch (include/rtk_train.h: rtk_backbone_loss).
// Reference: losses/loss.py:8-31,85-89,124-146 as generalised to the batch mean in ratrack_amd/loss.py (backbone_loss):
//     sf_b  = mean_n || pc1 + flow - gt ||_2                         sf  = mean_b sf_b      (NaN -> 0)
//     seg_b = 0.4 sum(bce g)/npos + 0.6 sum(bce (1-g))/nneg           seg = mean_b seg_b     (a sample without positives or
//                                                                                            without negatives contributes 0)
//     total = seg                        while pre-training
//           = 0.5 sf + 0.5 trk + seg     otherwise (trk = 0 on the backbone path)
// with bce = -(g max(log p, -100) + (1-g) max(log(1-p), -100)) (torch.nn.functional.binary_cross_entropy).  The framework
// formulation costs ~50 tiny kernels forward + backward; here one workgroup per sample reduces its sample and writes the
// gradients d total / d flow and d total / d cls directly:
//     d/dflow[b,c,n] = 0.5 / (B N) diff_c / ||diff||                  (0 where ||diff|| == 0)
//     d/dcls[b,n]    = [defined_b] / B (0.4 g / npos + 0.6 (1-g) / nneg) (p - g) / max(p (1-p), 1e-12)
// the last factor being torch's own binary_cross_entropy backward.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdlib.h>

#include "rtk_common.h"
#include "rtk_train.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float *s_red) {      // 256 threads -> every thread gets the sum
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

__global__ __launch_bounds__(256) void backbone_loss_kernel(int B, int N, const float *__restrict__ pc1, const float *__restrict__ flow,
                                                            const float *__restrict__ gt, const float *__restrict__ cls,
                                                            const unsigned char *__restrict__ gt_cls, int gt_cls_stride, int pretrain,
                                                            float *__restrict__ items, float *__restrict__ dflow,
                                                            float *__restrict__ dcls, const int *__restrict__ n_valid) {
    __shared__ float s_red[4];
    __shared__ int s_last;
    const int b = blockIdx.x, t = threadIdx.x;
    const int nv = n_valid ? n_valid[b] : N;        // padded batch: the sample's own point count (padding columns: zero gradient)
    const float *p1 = pc1 + (size_t)b * 3 * N, *fl = flow + (size_t)b * 3 * N, *g3 = gt + (size_t)b * 3 * N;
    const float *pc = cls + (size_t)b * N;
    const unsigned char *gc = gt_cls + (size_t)b * gt_cls_stride;
    float sf = 0.f, npos = 0.f, nneg = 0.f, spos = 0.f, sneg = 0.f;
    for (int n = t; n < N; n += 256) {
        if (n >= nv) {
            if (dflow && !pretrain) {
                float *o = dflow + (size_t)b * 3 * N;
                o[n] = 0.f; o[N + n] = 0.f; o[2 * N + n] = 0.f;
            }
            continue;
        }
        const float dx = (p1[n] + fl[n]) - g3[n], dy = (p1[N + n] + fl[N + n]) - g3[N + n], dz = (p1[2 * N + n] + fl[2 * N + n]) - g3[2 * N + n];
        const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
        sf += nrm;
        if (dflow && !pretrain) {
            const float k = nrm > 0.f ? 0.5f / ((float)B * (float)nv) / nrm : 0.f;
            float *o = dflow + (size_t)b * 3 * N;
            o[n] = k * dx; o[N + n] = k * dy; o[2 * N + n] = k * dz;
        }
        const float p = pc[n];
        const bool pos = gc[n] != 0;
        const float bce = -(pos ? fmaxf(logf(p), -100.f) : fmaxf(logf(1.f - p), -100.f));
        if (pos) { npos += 1.f; spos += bce; } else { nneg += 1.f; sneg += bce; }
    }
    sf = block_sum(sf, s_red);
    npos = block_sum(npos, s_red);
    nneg = block_sum(nneg, s_red);
    spos = block_sum(spos, s_red);
    sneg = block_sum(sneg, s_red);
    const bool defined = npos > 0.f && nneg > 0.f;
    const float wp = 0.4f / fmaxf(npos, 1.f), wn = 0.6f / fmaxf(nneg, 1.f);
    if (dcls) {
        for (int n = t; n < N; n += 256) {
            const float p = pc[n], g = gc[n] != 0 ? 1.f : 0.f;
            const float w = defined && n < nv ? (g != 0.f ? wp : wn) / (float)B : 0.f;
            dcls[(size_t)b * N + n] = w * (p - g) / fmaxf((1.f - p) * p, 1e-12f);
        }
    }
    if (t == 0) {
        float sfb = sf / (float)nv;
        sfb = sfb != sfb ? 0.f : sfb;                               // NaN -> 0 (losses/loss.py:15-20)
        const float segb = defined ? wp * spos + wn * sneg : 0.f;
        const float sfm = sfb / (float)B, segm = segb / (float)B;
        // the batch means: every sample leaves its share, the last workgroup to arrive adds them up sample 0 first (float atomics
        // summed in arrival order: the loss differed from run to run in its last bit)
        float *share = items + 5;
        int *ticket = reinterpret_cast<int *>(items + 4);
        share[2 * b] = sfm;
        share[2 * b + 1] = segm;
        __threadfence();
        s_last = atomicAdd(ticket, 1) == B - 1;
    }
    __syncthreads();
    if (s_last && t < 64) {                                  // (workgroup-uniform: the last workgroup's first wave)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const float *share = items + 5;
        float s_sf = 0.f, s_seg = 0.f, s_all = 0.f;
        for (int k = t; k < B; k += 64) {                             // lane l: samples l, l + 64, ...; then the xor tree -- a fixed order
            const float a = __builtin_nontemporal_load(share + 2 * k), c = __builtin_nontemporal_load(share + 2 * k + 1);
            s_sf += a;
            s_seg += c;
            s_all += pretrain ? c : 0.5f * a + c;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s_sf += __shfl_xor(s_sf, o, 64); s_seg += __shfl_xor(s_seg, o, 64); s_all += __shfl_xor(s_all, o, 64); }
        if (t == 0) {
            items[1] = s_sf;                                          // SceneFlowLoss
            items[3] = s_seg;                                         // SegLoss
            items[0] = s_all;                                         // Loss  (items[2] = TrackingLoss stays 0)
            *reinterpret_cast<int *>(items + 4) = 0;
        }
    }
}

}  // namespace

extern "C" int rtk_backbone_loss(int b, int n, const float *pc1, const float *flow, const float *gt_warp, const float *cls,
                                 const unsigned char *gt_cls, int gt_cls_stride, int pretrain, float *items, float *dflow, float *dcls,
                                 const int *n_valid, rtk_stream_t stream) {
    RTK_REQUIRE(b > 0 && n > 0 && pc1 && flow && gt_warp && cls && gt_cls && items, "backbone_loss: bad arguments");
    backbone_loss_kernel<<<b, 256, 0, (hipStream_t)stream>>>(b, n, pc1, flow, gt_warp, cls, gt_cls, gt_cls_stride, pretrain, items, dflow, dcls,
                                                             n_valid);
    RTK_CHECK_LAUNCH("backbone_loss");
    return RTK_OK;
}

// ---- WeightNet parameter gradients (rtk_weightnet_bwd) --------------------------------------------------------------------------
// WeightNet(3 -> 8 -> 8 -> C, ReLU after every layer, utils/model_utils/model_utils.py:359-390) of the cost volume / patch
// aggregation.  The backward kernels of those operators emit, per (point, neighbour) position m: d4[m] = (direction, 1),
// dq3[m] (C) = gradient of the last pre-activation (ReLU mask applied) and dt2[m] (8) = Wc^T dq3[m].  Here the two hidden
// activations are recomputed and all six parameter gradients accumulated in one pass over dq3 (the only large operand):
//     t1 = relu(Wa d + ba), t2 = relu(Wb t1 + bb)
//     dWc = sum dq3 t2^T, dbc = sum dq3;   g2 = dt2 [t2 > 0]: dWb = sum g2 t1^T, dbb = sum g2;
//     g1 = (Wb^T g2) [t1 > 0]: dWa = sum g1 d^T, dba = sum g1.
// Round 1 did this with ~20 framework kernels per call (addmm, relu, cat, three slab-split GEMMs + reductions, masks).
namespace {

constexpr int WN_SUB = 256;       // positions per LDS sub-block

__global__ __launch_bounds__(256) void weightnet_bwd_kernel(long M, int C, int sub_per_wg, const float *__restrict__ d4,
                                                            const float *__restrict__ dq3, const float *__restrict__ dt2,
                                                            const float *__restrict__ wa, const float *__restrict__ ba,
                                                            const float *__restrict__ wb, const float *__restrict__ bb,
                                                            float *__restrict__ partial, int partial_pitch) {
    __shared__ float s_t1[WN_SUB][9], s_g1[WN_SUB][9], s_g2[WN_SUB][9], s_d[WN_SUB][4];
    __shared__ __attribute__((aligned(16))) float s_t2[WN_SUB][8];      // read as two float4 per position (broadcast) in the dWc loop
    __shared__ float s_wa[8][3], s_ba[8], s_wb[8][8], s_bb[8];
    const int t = threadIdx.x;
    if (t < 24) s_wa[t / 3][t % 3] = wa[t];
    if (t < 8) { s_ba[t] = ba[t]; s_bb[t] = bb[t]; }
    if (t < 64) s_wb[t >> 3][t & 7] = wb[t];
    __syncthreads();
    // dWc / dbc: a thread owns FOUR consecutive channels (one float4 per dq3 row: 4x the bytes in flight of a thread-per-channel
    // mapping, which left this stream at 1.3 TB/s) of the rows p = phase (mod 4), phase = its wave; two passes cover C <= 512
    const int cq = (t & 63) * 4, ph = t >> 6;
    float accc[2][4][8], accb[2][4];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            accb[q][i] = 0.f;
#pragma unroll
            for (int h = 0; h < 8; ++h) accc[q][i][h] = 0.f;
        }
    // The four small gradients as ONE 16 x 16 MFMA accumulator per wave: rows = [g2 (8) | g1 (8)], columns = [t1 (8) | d (3) | 1 | 0..],
    // contracted over the positions (4 per instruction): dWb = g2 x t1, dbb = g2 x 1, dWa = g1 x d, dba = g1 x 1 sit in its blocks.
    // (A serial loop of 104 threads over the sub-block's 256 positions was half of the kernel's run time.)
    typedef float wn_f4 __attribute__((ext_vector_type(4)));
    wn_f4 accs = {0.f, 0.f, 0.f, 0.f};
    for (int sb = 0; sb < sub_per_wg; ++sb) {
        const long m0 = ((long)blockIdx.x * sub_per_wg + sb) * WN_SUB;
        if (m0 >= M) break;
        const int cnt = (int)((M - m0) < WN_SUB ? (M - m0) : WN_SUB);
        __syncthreads();
        if (t < cnt) {       // hidden activations and their gradients of position m0 + t
            const long m = m0 + t;
            const float dx = d4[m * 4 + 0], dy = d4[m * 4 + 1], dz = d4[m * 4 + 2];
            float t1[8], t2[8], g2[8];
            const float4 dta = *reinterpret_cast<const float4 *>(dt2 + m * 8), dtb = *reinterpret_cast<const float4 *>(dt2 + m * 8 + 4);
            const float dtv[8] = {dta.x, dta.y, dta.z, dta.w, dtb.x, dtb.y, dtb.z, dtb.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) t1[k] = fmaxf(s_wa[k][0] * dx + s_wa[k][1] * dy + s_wa[k][2] * dz + s_ba[k], 0.f);
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                float a = s_bb[h];
#pragma unroll
                for (int k = 0; k < 8; ++k) a += s_wb[h][k] * t1[k];
                t2[h] = fmaxf(a, 0.f);
                g2[h] = t2[h] > 0.f ? dtv[h] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float a = 0.f;
#pragma unroll
                for (int h = 0; h < 8; ++h) a += s_wb[h][k] * g2[h];
                s_g1[t][k] = t1[k] > 0.f ? a : 0.f;
                s_t1[t][k] = t1[k]; s_t2[t][k] = t2[k]; s_g2[t][k] = g2[k];
            }
            s_d[t][0] = dx; s_d[t][1] = dy; s_d[t][2] = dz;
        }
        __syncthreads();
        // eight rows in flight per thread: unconditional loads from clamped rows, zeros selected afterwards
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = cq + 256 * q;
            if (c >= C) break;
            const float *col = dq3 + m0 * C + c;
            for (int p0 = ph; p0 < cnt; p0 += 32) {
                float4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float4 *>(col + (long)min(p0 + 4 * k, cnt - 1) * C);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int row = p0 + 4 * k;
                    if (row >= cnt) v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    const int rr = min(row, cnt - 1);
                    const float4 ta = *reinterpret_cast<const float4 *>(&s_t2[rr][0]), tb = *reinterpret_cast<const float4 *>(&s_t2[rr][4]);
                    const float th[8] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w};
                    const float vv[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        accb[q][i] += vv[i];
#pragma unroll
                        for (int h = 0; h < 8; ++h) accc[q][i][h] += vv[i] * th[h];
                    }
                }
            }
        }
        {
            const int lane = t & 63, g = lane >> 4, i = lane & 15;
#pragma unroll 4
            for (int q = 0; q < 16; ++q) {
                const int p = 64 * ph + 4 * q + g;                 // wave ph contracts positions [64 ph, 64 ph + 64)
                const bool ok = p < cnt;
                const int pc = ok ? p : 0;
                const float a = i < 8 ? s_g2[pc][i] : s_g1[pc][i - 8];
                const float bq = i < 8 ? s_t1[pc][i] : i < 11 ? s_d[pc][i - 8] : (i == 11 ? 1.f : 0.f);
                accs = __builtin_amdgcn_mfma_f32_16x16x4f32(ok ? a : 0.f, ok ? bq : 0.f, accs, 0, 0, 0);
            }
        }
    }
    // this workgroup's partial gradients [dbc (C) | dWc (8C) | dWb (64) dbb (8) dWa (24) dba (8)]: plain stores, added up in a fixed
    // order by weightnet_bwd_reduce_kernel.  (With float atomics on the 9C + 104 outputs -- 512 workgroups deep on every address --
    // the kernel took 200 us for 38 us of streaming.)
    // the four waves (row phases) add into one LDS image in turn, then the workgroup stores it
    __shared__ float s_acc[9 * 512];
    for (int w = 0; w < 4; ++w) {
        if (ph == w) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int c = cq + 256 * q;
                if (c < C) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        s_acc[c + i] = w == 0 ? accb[q][i] : s_acc[c + i] + accb[q][i];
#pragma unroll
                        for (int h = 0; h < 8; ++h) {
                            float &d = s_acc[C + (c + i) * 8 + h];
                            d = w == 0 ? accc[q][i][h] : d + accc[q][i][h];
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    float *part = partial + (size_t)blockIdx.x * partial_pitch;
    for (int e = t; e < 9 * C; e += 256) part[e] = s_acc[e];
    // accs[r] = D[row 4g + r][column i] of this wave; the four waves' tiles are added through LDS (s_acc is free again)
    __syncthreads();
    {
        const int lane = t & 63, g = lane >> 4, i = lane & 15;
#pragma unroll
        for (int r = 0; r < 4; ++r) s_acc[ph * 256 + (4 * g + r) * 16 + i] = accs[r];
    }
    __syncthreads();
    if (t < 104) {
        int row, col;
        if (t < 64) { row = t >> 3; col = t & 7; }                        // dWb[h][k]
        else if (t < 72) { row = t - 64; col = 11; }                     // dbb[h]
        else if (t < 96) { row = 8 + (t - 72) / 3; col = 8 + (t - 72) % 3; }      // dWa[k][j]
        else { row = 8 + (t - 96); col = 11; }                           // dba[k]
        const int e = row * 16 + col;
        part[9 * C + t] = (s_acc[e] + s_acc[256 + e]) + (s_acc[512 + e] + s_acc[768 + e]);
    }
}

__global__ __launch_bounds__(256) void weightnet_bwd_reduce_kernel(int C, int wgs, const float *__restrict__ partial, int partial_pitch,
                                                                   float *__restrict__ dwa, float *__restrict__ dba, float *__restrict__ dwb,
                                                                   float *__restrict__ dbb, float *__restrict__ dwc, float *__restrict__ dbc) {
    // 16 outputs x 16 lanes over the workgroup axis, eight independent loads per lane and round: the sum over ~500 partial vectors
    // is a latency problem (a few rounds of HBM/L2 round trips), not a bandwidth one
    __shared__ float s_part[16][17];
    const int el = threadIdx.x & 15, zl = threadIdx.x >> 4, L = 9 * C + 104;
    const int e = blockIdx.x * 16 + el;
    const float *src = partial + (e < L ? e : L - 1);
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = 0.f;
    for (int z0 = zl; z0 < wgs; z0 += 128) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int z = z0 + 16 * k;
            const float v = src[(size_t)(z < wgs ? z : 0) * partial_pitch];
            a[k] += z < wgs ? v : 0.f;
        }
    }
    s_part[zl][el] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    __syncthreads();
    if (zl || e >= L) return;
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) sum += s_part[q][el];
    float *dst;
    if (e < C) dst = dbc + e;
    else if (e < 9 * C) dst = dwc + (e - C);
    else {
        const int r = e - 9 * C;
        dst = r < 64 ? dwb + r : r < 72 ? dbb + (r - 64) : r < 96 ? dwa + (r - 72) : dba + (r - 96);
    }
    *dst += sum;
}

}  // namespace

extern "C" int rtk_weightnet_bwd(long positions, int channels, const float *d4, const float *dq3, const float *dt2, const float *wa,
                                 const float *ba, const float *wb, const float *bb, float *dwa, float *dba, float *dwb, float *dbb,
                                 float *dwc, float *dbc, float *workspace, long workspace_floats, rtk_stream_t stream) {
    RTK_REQUIRE(positions > 0 && channels > 0 && channels <= 512 && (channels & 3) == 0 && d4 && dq3 && dt2 && wa && ba && wb && bb && dwa && dba && dwb && dbb &&
                dwc && dbc && workspace, "weightnet_bwd: bad arguments");
    const int pitch = (9 * channels + 104 + 3) & ~3;
    RTK_REQUIRE(workspace_floats >= pitch, "weightnet_bwd: workspace of %ld floats < %d", workspace_floats, pitch);
    const long subs = (positions + WN_SUB - 1) / WN_SUB;
    long want = 512;                                // workgroups (two per CU keep the dq3 stream near HBM speed), each one partial vector
    if (want > workspace_floats / pitch) want = workspace_floats / pitch;
    int per = (int)((subs + want - 1) / want);
    if (per < 1) per = 1;
    const int wgs = (int)((subs + per - 1) / per);
    weightnet_bwd_kernel<<<wgs, 256, 0, (hipStream_t)stream>>>(positions, channels, per, d4, dq3, dt2, wa, ba, wb, bb, workspace, pitch);
    RTK_CHECK_LAUNCH("weightnet_bwd");
    weightnet_bwd_reduce_kernel<<<(9 * channels + 104 + 15) / 16, 256, 0, (hipStream_t)stream>>>(channels, wgs, workspace, pitch, dwa, dba, dwb,
                                                                                                  dbb, dwc, dbc);
    RTK_CHECK_LAUNCH("weightnet_bwd_reduce");
    return RTK_OK;
}
