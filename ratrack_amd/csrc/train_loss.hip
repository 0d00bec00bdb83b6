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
