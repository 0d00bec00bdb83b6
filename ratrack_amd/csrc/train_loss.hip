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
                                                            float *__restrict__ dcls) {
    __shared__ float s_red[4];
    const int b = blockIdx.x, t = threadIdx.x;
    const float *p1 = pc1 + (size_t)b * 3 * N, *fl = flow + (size_t)b * 3 * N, *g3 = gt + (size_t)b * 3 * N;
    const float *pc = cls + (size_t)b * N;
    const unsigned char *gc = gt_cls + (size_t)b * gt_cls_stride;
    float sf = 0.f, npos = 0.f, nneg = 0.f, spos = 0.f, sneg = 0.f;
    for (int n = t; n < N; n += 256) {
        const float dx = (p1[n] + fl[n]) - g3[n], dy = (p1[N + n] + fl[N + n]) - g3[N + n], dz = (p1[2 * N + n] + fl[2 * N + n]) - g3[2 * N + n];
        const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
        sf += nrm;
        if (dflow && !pretrain) {
            const float k = nrm > 0.f ? 0.5f / ((float)B * (float)N) / nrm : 0.f;
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
            const float w = defined ? (g != 0.f ? wp : wn) / (float)B : 0.f;
            dcls[(size_t)b * N + n] = w * (p - g) / fmaxf((1.f - p) * p, 1e-12f);
        }
    }
    if (t == 0) {
        float sfb = sf / (float)N;
        sfb = sfb != sfb ? 0.f : sfb;                               // NaN -> 0 (losses/loss.py:15-20)
        const float segb = defined ? wp * spos + wn * sneg : 0.f;
        const float sfm = sfb / (float)B, segm = segb / (float)B;
        atomicAdd(items + 1, sfm);                                   // SceneFlowLoss
        atomicAdd(items + 3, segm);                                  // SegLoss
        atomicAdd(items + 0, pretrain ? segm : 0.5f * sfm + segm);   // Loss  (items[2] = TrackingLoss stays 0)
    }
}

}  // namespace

extern "C" int rtk_backbone_loss(int b, int n, const float *pc1, const float *flow, const float *gt_warp, const float *cls,
                                 const unsigned char *gt_cls, int gt_cls_stride, int pretrain, float *items, float *dflow, float *dcls,
                                 rtk_stream_t stream) {
    RTK_REQUIRE(b > 0 && n > 0 && pc1 && flow && gt_warp && cls && gt_cls && items, "backbone_loss: bad arguments");
    backbone_loss_kernel<<<b, 256, 0, (hipStream_t)stream>>>(b, n, pc1, flow, gt_warp, cls, gt_cls, gt_cls_stride, pretrain, items, dflow, dcls);
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
                                                            float *__restrict__ dwa, float *__restrict__ dba, float *__restrict__ dwb,
                                                            float *__restrict__ dbb, float *__restrict__ dwc, float *__restrict__ dbc) {
    __shared__ float s_t1[WN_SUB][9], s_t2[WN_SUB][9], s_g1[WN_SUB][9], s_g2[WN_SUB][9], s_d[WN_SUB][4];
    __shared__ float s_wa[8][3], s_ba[8], s_wb[8][8], s_bb[8];
    const int t = threadIdx.x;
    if (t < 24) s_wa[t / 3][t % 3] = wa[t];
    if (t < 8) { s_ba[t] = ba[t]; s_bb[t] = bb[t]; }
    if (t < 64) s_wb[t >> 3][t & 7] = wb[t];
    __syncthreads();
    float accc[8][2];      // this thread's channels c = t and t + 256 (C <= 512): dWc[c][0..7]
    float accb[2] = {0.f, 0.f};
#pragma unroll
    for (int h = 0; h < 8; ++h) { accc[h][0] = 0.f; accc[h][1] = 0.f; }
    float small = 0.f;     // threads 0..103: one element of dWb (64) | dbb (8) | dWa (24) | dba (8)
    for (int sb = 0; sb < sub_per_wg; ++sb) {
        const long m0 = ((long)blockIdx.x * sub_per_wg + sb) * WN_SUB;
        if (m0 >= M) break;
        const int cnt = (int)((M - m0) < WN_SUB ? (M - m0) : WN_SUB);
        __syncthreads();
        if (t < cnt) {       // hidden activations and their gradients of position m0 + t
            const long m = m0 + t;
            const float dx = d4[m * 4 + 0], dy = d4[m * 4 + 1], dz = d4[m * 4 + 2];
            float t1[8], t2[8], g2[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t1[k] = fmaxf(s_wa[k][0] * dx + s_wa[k][1] * dy + s_wa[k][2] * dz + s_ba[k], 0.f);
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                float a = s_bb[h];
#pragma unroll
                for (int k = 0; k < 8; ++k) a += s_wb[h][k] * t1[k];
                t2[h] = fmaxf(a, 0.f);
                g2[h] = t2[h] > 0.f ? dt2[m * 8 + h] : 0.f;
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
        // dWc / dbc: thread = channel, loop over the sub-block's positions (rows of dq3 are read coalesced)
        for (int q = 0; q < 2; ++q) {
            const int c = t + 256 * q;
            if (c >= C) break;
            const float *col = dq3 + m0 * C + c;
            for (int p = 0; p < cnt; ++p) {
                const float v = col[(long)p * C];
                accb[q] += v;
#pragma unroll
                for (int h = 0; h < 8; ++h) accc[h][q] += v * s_t2[p][h];
            }
        }
        if (t < 104) {
            for (int p = 0; p < cnt; ++p) {
                if (t < 64) small += s_g2[p][t >> 3] * s_t1[p][t & 7];
                else if (t < 72) small += s_g2[p][t - 64];
                else if (t < 96) small += s_g1[p][(t - 72) / 3] * s_d[p][(t - 72) % 3];
                else small += s_g1[p][t - 96];
            }
        }
    }
    for (int q = 0; q < 2; ++q) {
        const int c = t + 256 * q;
        if (c >= C) break;
        atomicAdd(dbc + c, accb[q]);
#pragma unroll
        for (int h = 0; h < 8; ++h) atomicAdd(dwc + c * 8 + h, accc[h][q]);
    }
    if (t < 64) atomicAdd(dwb + t, small);
    else if (t < 72) atomicAdd(dbb + t - 64, small);
    else if (t < 96) atomicAdd(dwa + t - 72, small);
    else if (t < 104) atomicAdd(dba + t - 96, small);
}

}  // namespace

extern "C" int rtk_weightnet_bwd(long positions, int channels, const float *d4, const float *dq3, const float *dt2, const float *wa,
                                 const float *ba, const float *wb, const float *bb, float *dwa, float *dba, float *dwb, float *dbb,
                                 float *dwc, float *dbc, rtk_stream_t stream) {
    RTK_REQUIRE(positions > 0 && channels > 0 && channels <= 512 && d4 && dq3 && dt2 && wa && ba && wb && bb && dwa && dba && dwb && dbb &&
                dwc && dbc, "weightnet_bwd: bad arguments");
    const long subs = (positions + WN_SUB - 1) / WN_SUB;
    int per = (int)((subs + 511) / 512);           // ~512 workgroups, each a few sub-blocks: its atomics stay a small share
    if (per < 1) per = 1;
    const int wgs = (int)((subs + per - 1) / per);
    weightnet_bwd_kernel<<<wgs, 256, 0, (hipStream_t)stream>>>(positions, channels, per, d4, dq3, dt2, wa, ba, wb, bb, dwa, dba, dwb, dbb, dwc,
                                                              dbc);
    RTK_CHECK_LAUNCH("weightnet_bwd");
    return RTK_OK;
}
