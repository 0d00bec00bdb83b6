// fused_group.hip -- the neighbourhood stages of the backbone as single kernels:
//
//   rtk_sa_scale     QueryAndGroup + SharedMLP + max_pool2d of one MSG scale
//                    (lib/pointnet2_utils.py:269-292, lib/pointnet2_modules.py:37-53)
//   rtk_cost_volume  point-to-patch cost volume of FeatureCorrelator (model_utils.py:216-236)
//   rtk_patch_cost   patch-to-patch aggregation (model_utils.py:238-248)
//
// The reference materialises the grouped tensor (B, C+3, S, ns) -- 539 MB for the 514-channel level --
// and every activation after it.  Here a wave owns 16 (centroid, neighbour) pairs; the grouped
// features never exist in memory: layer 1 is  q[idx] + Wx.(xyz[idx] - centroid) + b  where q is the
// per-POINT projection of the features (linearity of the 1x1 conv) and the 3-channel offset term is
// one MFMA k-step whose C-in is the gathered q row; every following layer runs register-to-register
// (fused_common.h) and the neighbourhood reduction (max / weighted sum) is a DPP row reduction.
#include <string.h>

#include "rtk_common.h"
#include "fused_common.h"
#include "rtk_fused.h"
#include "rtk_train.h"

// The single k-step "offset" layer: A operand image [V][64] floats, lane (g, i) holds W4[16v + i][g] with
// W4 = [Wx | b] (Cout x 4); B operand: lane (g, j) holds (dx, dy, dz, 1)[g] of pair j.

__device__ __forceinline__ f4 f4_max(f4 a, f4 b) { return (f4){fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)}; }
__device__ __forceinline__ f4 f4_relu(f4 a) { return (f4){fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f)}; }

// =================================================================================================
// rtk_sa_scale
// =================================================================================================
#ifndef SA_MAX_WGS
#define SA_MAX_WGS 1024
#endif
struct SaParams {
    int samples, n, npoint;
    const float *xyz, *new_xyz;
    const int *idx;
    const float *q;
    int q_pitch;
    const float *w1;        // offset layer image [V1][64]
    const f4 *blob;         // layers 2(,3): packed, contiguous
    const float *bias2, *bias3;
    float *out;
    int out_pitch, out_offset;
    const int *src_nuniq, *dst_nuniq;
    int gx;                 // > 0: XCD-aware 1-D grid (rtk_decode_block)
};

// NS = neighbours per centroid; V1/V2/V3 = layer widths / 16 (V3 = 0: two-layer MLP).
template <int NS, int V1, int V2, int V3>
__global__ __launch_bounds__(256) void sa_scale_kernel(const SaParams P) {
    constexpr int NF = V1 * V2 + V2 * V3;
    constexpr int VL = V3 ? V3 : V2;                 // width of the last layer
    constexpr int TILES = NS > 16 ? NS / 16 : 1;     // tiles per work unit (NS = 32: one centroid = 2 tiles)
    constexpr int CPT = NS >= 16 ? 1 : 16 / NS;      // centroids per tile
    __shared__ __attribute__((aligned(16))) f4 s_w[NF * 64];
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    // A workgroup owns one sample and strides over its centroid groups -- the index arithmetic inside the loop is 32-bit and
    // division-free (a 64-bit divide per tile cost more issue slots than the MFMAs of the small scales).
    // Workgroups whose groups are all duplicates (>= nuniq[b], copies of centroid 0) exit at once; the dispatcher
    // back-fills them, which spreads the live work evenly.
    int b, bx, nbx;
    rtk_decode_block(P.gx, b, bx, nbx);
    const int dst_e = P.dst_nuniq ? __builtin_amdgcn_readfirstlane(P.dst_nuniq[b]) : P.npoint;
    const int live_groups = (min(dst_e, P.npoint) + CPT - 1) / CPT;
    if (bx * 4 >= live_groups) return;
    for (int i = threadIdx.x; i < NF * 64; i += blockDim.x) s_w[i] = P.blob[i];
    float w1[V1];
#pragma unroll
    for (int v = 0; v < V1; ++v) w1[v] = P.w1[v * 64 + lane];
    __syncthreads();
    const int slot0 = j % NS;                         // neighbour slot of this lane within the first tile
    for (int unit = bx * 4 + (threadIdx.x >> 6); unit < live_groups; unit += nbx * 4) {
        int cl = unit * CPT + (NS >= 16 ? 0 : j / NS);        // centroid of this lane within the sample
        const bool valid = cl < P.npoint;
        if (!valid) cl = P.npoint - 1;
        const int c = b * P.npoint + cl;                      // global centroid index (fits 32 bits: asserted by the launcher)
        const int src_e = P.src_nuniq ? P.src_nuniq[b] : 0x7fffffff;
        const float cg = g < 3 ? P.new_xyz[(long)c * 3 + g] : 0.f;
        f4 best[VL];
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            const int slot = slot0 + 16 * t;
            const int id = P.idx[(long)c * NS + slot];
            const int src = b * P.n + id;
            // offset operand: (neighbour - centroid) component g, or 1 for the bias column
            const float bop = g < 3 ? __fsub_rn(P.xyz[(long)src * 3 + g], cg) : 1.0f;
            f4 a1[V1];
            const float *qrow = P.q + (long)(b * P.n + (id < src_e ? id : 0)) * P.q_pitch + 4 * g;   // duplicate source rows alias row 0
#pragma unroll
            for (int v = 0; v < V1; ++v) a1[v] = *reinterpret_cast<const f4 *>(qrow + 16 * v);
#pragma unroll
            for (int v = 0; v < V1; ++v) a1[v] = mfma4(w1[v], bop, a1[v]);      // + Wx.d + b1 (one MFMA k-step, C-in = q row)
#pragma unroll
            for (int v = 0; v < V1; ++v) a1[v] = f4_relu(a1[v]);
            f4 a2[V2];
            if constexpr (V3 == 0) {
#pragma unroll
                for (int v = 0; v < V2; ++v) a2[v] = f4_zero();     // last layer: bias + ReLU after the max
                mlp_layer_res<V1, V2>(s_w, lane, a1, a2);
#pragma unroll
                for (int v = 0; v < V2; ++v) best[v] = t == 0 ? a2[v] : f4_max(best[v], a2[v]);
            } else {
#pragma unroll
                for (int v = 0; v < V2; ++v) a2[v] = bias_frag(P.bias2, v, g);
                mlp_layer_res<V1, V2>(s_w, lane, a1, a2);
#pragma unroll
                for (int v = 0; v < V2; ++v) a2[v] = f4_relu(a2[v]);
                f4 a3[V3 ? V3 : 1];
#pragma unroll
                for (int v = 0; v < V3; ++v) a3[v] = f4_zero();
                mlp_layer_res<V2, (V3 ? V3 : 1)>(s_w + V1 * V2 * 64, lane, a2, a3);
#pragma unroll
                for (int v = 0; v < V3; ++v) best[v] = t == 0 ? a3[v] : f4_max(best[v], a3[v]);
            }
        }
        // bias, max over the neighbours of each centroid (DPP within the 16-lane row), ReLU.  The bias goes in BEFORE the maximum
        // (max(x) + b == max(x + b) bit for bit: rounding is monotonic): with one tile per centroid best[] is the raw MFMA result, and
        // the inline-asm reduction must sit behind a VALU instruction the compiler knows -- it inserts no MFMA -> VALU wait states in
        // front of inline asm (the lesson of the register-capped split kernel).
        const float *bl = V3 ? P.bias3 : P.bias2;
#pragma unroll
        for (int v = 0; v < VL; ++v) {
            f4 m = best[v] + bias_frag(bl, v, g);
            row_max_group_f4<(NS > 16 ? 16 : NS)>(m);
            best[v] = f4_relu(m);
        }
        if (valid && slot0 == 0) {
            float *o = P.out + (long)c * P.out_pitch + P.out_offset + 4 * g;
#pragma unroll
            for (int v = 0; v < VL; ++v) *reinterpret_cast<f4 *>(o + 16 * v) = best[v];
        }
    }
}

extern "C" int rtk_sa_scale(int samples, int n, int npoint, int nsample, const float *xyz, const float *new_xyz,
                            const int *idx, const float *q, int q_pitch, int c1_16, const float *w1xyz_packed,
                            int nlayers, const rtk_layer_t *layers, float *out, int out_pitch, int out_offset,
                            const int *src_nuniq, const int *dst_nuniq, rtk_stream_t stream) {
    RTK_REQUIRE(samples > 0 && n > 0 && npoint > 0 && xyz && new_xyz && idx && q && w1xyz_packed && layers && out,
                "sa_scale: bad arguments");
    RTK_REQUIRE(nlayers == 1 || nlayers == 2, "sa_scale: nlayers=%d (1 or 2 layers after the offset layer)", nlayers);
    RTK_REQUIRE(q_pitch % 4 == 0 && out_pitch % 4 == 0 && out_offset % 4 == 0, "sa_scale: pitches/offset must be multiples of 4");
    RTK_REQUIRE(layers[0].cin16 == c1_16 && (nlayers == 1 || (layers[1].cin16 == layers[0].cout16 &&
                layers[1].w_packed == layers[0].w_packed + (size_t)layers[0].cin16 * layers[0].cout16 * 256)),
                "sa_scale: layer chain mismatch / not contiguous");
    SaParams P;
    P.samples = samples; P.n = n; P.npoint = npoint;
    P.xyz = xyz; P.new_xyz = new_xyz; P.idx = idx; P.q = q; P.q_pitch = q_pitch;
    P.w1 = w1xyz_packed;
    P.blob = reinterpret_cast<const f4 *>(layers[0].w_packed);
    P.bias2 = layers[0].bias;
    P.bias3 = nlayers == 2 ? layers[1].bias : nullptr;
    P.out = out; P.out_pitch = out_pitch; P.out_offset = out_offset;
    P.src_nuniq = src_nuniq; P.dst_nuniq = dst_nuniq;
    const int v1 = c1_16, v2 = layers[0].cout16, v3 = nlayers == 2 ? layers[1].cout16 : 0;
    hipStream_t s = (hipStream_t)stream;
    RTK_REQUIRE((long)samples * npoint * nsample < 0x7fffffffL && (long)samples * n < 0x7fffffffL && samples <= 65535,
                "sa_scale: problem too large for 32-bit indexing");
    const int cpt = nsample >= 16 ? 1 : 16 / nsample;
    const int groups = (npoint + cpt - 1) / cpt;
    int bx = (groups + 3) / 4;                 // one unit per wave per pass ...
    while ((long)bx * samples > SA_MAX_WGS && bx > 1) bx = (bx + 1) / 2;   // few, fat workgroups: the LDS weight fill is paid per workgroup
    P.gx = samples % 8 == 0 ? bx : 0;
    const dim3 blocks = P.gx ? dim3(bx * samples) : dim3(bx, samples);
    const long key = (((long)nsample * 32 + v1) * 32 + v2) * 32 + v3;
#define SA_CASE(ns, a, b, c)                                              \
    case (((long)(ns) * 32 + (a)) * 32 + (b)) * 32 + (c):                 \
        sa_scale_kernel<ns, a, b, c><<<blocks, 256, 0, s>>>(P);           \
        break;
    switch (key) {
        SA_CASE(4, 1, 1, 2)     // sa1 scale 0: r=2,  ns=4,  16-16-32
        SA_CASE(8, 1, 1, 2)     // sa1 scale 1: r=4,  ns=8,  16-16-32
        SA_CASE(8, 2, 2, 0)     // sa2 scale 0: r=4,  ns=8,  32-32
        SA_CASE(16, 2, 4, 0)    // sa2 scale 1: r=8,  ns=16, 32-64
        SA_CASE(16, 4, 4, 0)    // sa3 scale 0: r=8,  ns=16, 64-64
        SA_CASE(32, 4, 4, 0)    // sa3 scale 1: r=16, ns=32, 64-64
        default:
            rtk_set_error("sa_scale: no kernel instance for nsample=%d widths=(%d,%d,%d)x16", nsample, v1, v2, v3);
            return RTK_ERR_UNSUPPORTED;
    }
#undef SA_CASE
    RTK_CHECK_LAUNCH("sa_scale");
    return RTK_OK;
}

// =================================================================================================
// WeightNet on direction vectors (model_utils.py:359-390, bn=False): 3 -> 8 -> 8 -> C, ReLU after
// every conv.  Layer a is the single k-step offset layer ([Wa | ba] image), layer b one 16x16
// fragment, layer c [1][VC] fragments.  Returns relu(Wc.relu(Wb.relu(Wa.d+ba)+bb)+bc)[16v..] for one v.
// =================================================================================================
struct WnWeights {
    const float *wa;   // [1][64] offset image (8 outputs padded to 16)
    const f4 *wb;      // 1 fragment
    const f4 *wc;      // VC fragments
    const float *bb, *bc;
};

__device__ __forceinline__ f4 weightnet_hidden(const WnWeights &W, int lane, int g, float bop) {
    f4 t1 = f4_zero();
    t1 = mfma4(W.wa[lane], bop, t1);
    t1 = f4_relu(t1);
    f4 t2 = bias_frag(W.bb, 0, g);
    const f4 fb = W.wb[lane];
    t2 = mfma4(fb.x, t1.x, t2);
    t2 = mfma4(fb.y, t1.y, t2);
    t2 = mfma4(fb.z, t1.z, t2);
    t2 = mfma4(fb.w, t1.w, t2);
    return f4_relu(t2);
}

__device__ __forceinline__ f4 weightnet_out(const WnWeights &W, int lane, int g, int v, f4 t2) {
    f4 o = bias_frag(W.bc, v, g);
    const f4 fc = W.wc[v * 64 + lane];
    o = mfma4(fc.x, t2.x, o);
    o = mfma4(fc.y, t2.y, o);
    o = mfma4(fc.z, t2.z, o);
    o = mfma4(fc.w, t2.w, o);
    return f4_relu(o);
}

// =================================================================================================
// rtk_cost_volume
// =================================================================================================
#ifndef CV_NW
#define CV_NW 4          // waves per workgroup (all share one weight stream)
#endif
#ifndef CV_F
#define CV_F 8           // fragments (KiB) per half of the LDS double buffer of the FORWARD kernel.  Standalone the kernel is the same
                         // speed with 8, 16 or 32 (0.60-0.62 ms), but with two batches in flight the small footprint (2 x 16 KiB per CU
                         // instead of 2 x 64) lets the other batch's kernels co-reside: 1.421 vs 1.440 ms per step (tools/experiments/exp_cvf.sh)
#endif
#ifndef CVB_F
#define CVB_F 32         // ... of the backward kernel (runs alone in the train step)
#endif
#ifndef CV_WGS_PER_CU
#define CV_WGS_PER_CU 2
#endif
#define CV_V 16          // 256 channels

__device__ __forceinline__ f4 *cv_at(float *base, unsigned byte_off) {
    return reinterpret_cast<f4 *>(reinterpret_cast<char *>(base) + byte_off);
}

__device__ __forceinline__ const f4 *cv_at(const float *base, unsigned byte_off) {
    return reinterpret_cast<const f4 *>(reinterpret_cast<const char *>(base) + byte_off);
}

struct CvParams {
    int samples, n1, n2;
    const float *xyz1, *xyz2;
    const int64_t *knn;
    const float *p1, *p2;
    const float *wd;              // offset layer image [16][64] ([Wd | 0])
    const f4 *blob;               // layers 2,3 (256 + 256 fragments)
    const float *bias2, *bias3;
    WnWeights wn;
    float *out;
    int out_pitch;
    int gx;                       // workgroups per sample; > 0 selects the XCD-aware 1-D grid (rtk_decode_block)
    float *sv1, *sv2, *sv3;       // training forward (SAVE): the three activations, (positions, 256) each, kept for the backward
    uint2 *mk1, *mk2;             // ... and the sign masks of a1, a2: 64 bits per lane slot (position, g), bit 4v + r = [a[v][r] > 0]
};

// sign mask of an activation fragment set (the backward's leaky' selector: 8 bytes instead of the lane's 256 bytes of a1 / a2)
template <int V>
__device__ __forceinline__ uint2 sign_mask(const f4 (&a)[V]) {
    unsigned lo = 0, hi = 0;
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned bit = a[v][r] > 0.f ? 1u : 0u;
            if (v < 8) lo |= bit << (4 * v + r);
            else hi |= bit << (4 * (v - 8) + r);
        }
    return make_uint2(lo, hi);
}

// SAVE: the training forward (rtk_cost_volume_train) also stores a1, a2, a3 -- the backward then needs no recomputation (two of
// its four 256 x 256 products) and its weight-gradient GEMMs read the same tensors.
template <bool SAVE>
__global__ __launch_bounds__(64 * CV_NW, (CV_NW * CV_WGS_PER_CU) / 4) void cost_volume_kernel(const CvParams P) {
    __shared__ __attribute__((aligned(16))) f4 s_w[2 * CV_F * 64];
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const int wave_in_wg = threadIdx.x >> 6;
    // A workgroup owns one sample and strides over its points, CV_NW points (one per wave) per iteration.  The trip count is
    // the same for every wave of the workgroup (barriers in the weight stream).
    int b, bx, nbx;
    rtk_decode_block(P.gx, b, bx, nbx);
    const int groups = (P.n1 + CV_NW - 1) / CV_NW;
    constexpr int NF = 2 * CV_V * CV_V;
    WStream<CV_NW, CV_F, NF> ws;
    ws.start(P.blob, s_w, wave_in_wg, lane);
    for (int G = bx; G < groups; G += nbx) {
        asm volatile("" ::: "memory");   // keep loop-invariant weight/bias loads inside the loop (registers are the scarce resource)
        const int pt = G * CV_NW + wave_in_wg;                                  // query point within the sample
        const bool valid = pt < P.n1;
        const long i = (long)b * P.n1 + (valid ? pt : P.n1 - 1);               // global query index
        const long nb = (long)b * P.n2 + (long)P.knn[i * 16 + j];               // neighbour j in pc2
        const float bop = g < 3 ? __fsub_rn(P.xyz2[nb * 3 + g], P.xyz1[i * 3 + g]) : 1.0f;
        // layer 1: leaky(p1[i] + p2[nb] + Wd.d)     (bias folded into p1)
        f4 h[CV_V];
        {
            const float *r1 = P.p1 + i * 256 + 4 * g, *r2 = P.p2 + nb * 256 + 4 * g;
#pragma unroll
            for (int v = 0; v < CV_V; ++v)
                h[v] = *reinterpret_cast<const f4 *>(r1 + 16 * v) + *reinterpret_cast<const f4 *>(r2 + 16 * v);
#pragma unroll
            for (int v = 0; v < CV_V; ++v) h[v] = mfma4(P.wd[v * 64 + lane], bop, h[v]);
            apply_act<CV_V>(h, RTK_ACT_LEAKY);
        }
        // byte offset of this lane's 16-byte slot in a (position, 256) row (one 32-bit VGPR, added to uniform base pointers)
        const unsigned ro = (unsigned)(i * 16 + j) * 1024u + 16u * g;
        if (SAVE && valid) {
#pragma unroll
            for (int v = 0; v < CV_V; ++v) *cv_at(P.sv1, ro + 64u * v) = h[v];
            P.mk1[(i * 16 + j) * 4 + g] = sign_mask<CV_V>(h);
        }
        f4 a[CV_V];
#pragma unroll
        for (int v = 0; v < CV_V; ++v) a[v] = bias_frag(P.bias2, v, g);
        mlp_layer_ws<CV_V, CV_V, 0>(ws, h, a);
        apply_act<CV_V>(a, RTK_ACT_LEAKY);
        if (SAVE && valid) {
#pragma unroll
            for (int v = 0; v < CV_V; ++v) *cv_at(P.sv2, ro + 64u * v) = a[v];
            P.mk2[(i * 16 + j) * 4 + g] = sign_mask<CV_V>(a);
        }
#pragma unroll
        for (int v = 0; v < CV_V; ++v) h[v] = bias_frag(P.bias3, v, g);
        mlp_layer_ws<CV_V, CV_V, CV_V * CV_V>(ws, a, h);
        apply_act<CV_V>(h, RTK_ACT_LEAKY);
        ws.next();   // wrap the stream to chunk 0 (NF > F)
        if (SAVE && valid) {
#pragma unroll
            for (int v = 0; v < CV_V; ++v) *cv_at(P.sv3, ro + 64u * v) = h[v];
        }
        // WeightNet(direction) and the weighted sum over the 16 neighbours
        const f4 t2 = weightnet_hidden(P.wn, lane, g, bop);
        float *o = P.out + i * P.out_pitch + 4 * g;
#pragma unroll
        for (int v = 0; v < CV_V; ++v) {
            const f4 w = weightnet_out(P.wn, lane, g, v, t2);
            f4 r = w * h[v];
            row_sum16_f4(r);
            if (valid && j == 0) *reinterpret_cast<f4 *>(o + 16 * v) = r;
        }
    }
    ws.finish();
}

static int fill_wn(WnWeights &W, const rtk_layer_t *wn, const char *who) {
    // wn[0]: offset image (w_packed = [1][64] floats, cout16 = 1); wn[1]: 16->16 (1 fragment); wn[2]: 16 -> C
    if (!wn || !wn[0].w_packed || !wn[1].w_packed || !wn[2].w_packed || !wn[1].bias || !wn[2].bias ||
        wn[1].cin16 != 1 || wn[1].cout16 != 1 || wn[2].cin16 != 1) {
        rtk_set_error("%s: bad WeightNet layers", who);
        return RTK_ERR_INVALID;
    }
    W.wa = wn[0].w_packed;
    W.wb = reinterpret_cast<const f4 *>(wn[1].w_packed);
    W.wc = reinterpret_cast<const f4 *>(wn[2].w_packed);
    W.bb = wn[1].bias;
    W.bc = wn[2].bias;
    return RTK_OK;
}

static int cost_volume_launch(const char *who, int samples, int n1, int n2, const float *xyz1, const float *xyz2, const int64_t *knn_idx,
                              const float *p1, const float *p2, const float *wd_packed, const rtk_layer_t *layers, const rtk_layer_t *wn,
                              float *out, int out_pitch, float *a1, float *a2, float *a3, void *mask1, void *mask2, rtk_stream_t stream) {
    RTK_REQUIRE(samples > 0 && n1 > 0 && n2 >= 16 && xyz1 && xyz2 && knn_idx && p1 && p2 && wd_packed && layers && out,
                "%s: bad arguments", who);
    RTK_REQUIRE(layers[0].cin16 == 16 && layers[0].cout16 == 16 && layers[1].cin16 == 16 && layers[1].cout16 == 16 &&
                layers[1].w_packed == layers[0].w_packed + 256 * 256, "%s: expects two contiguous 256x256 layers", who);
    RTK_REQUIRE(out_pitch % 4 == 0 && out_pitch >= 256, "%s: bad out_pitch", who);
    const bool save = a1 != nullptr;
    RTK_REQUIRE(!save || ((double)samples * n1 * 16.0 * 1024.0 < 4294967296.0), "%s: more than 4 GiB per saved activation (32-bit row "
                "offsets): split the batch", who);
    CvParams P;
    P.samples = samples; P.n1 = n1; P.n2 = n2;
    P.xyz1 = xyz1; P.xyz2 = xyz2; P.knn = knn_idx; P.p1 = p1; P.p2 = p2; P.wd = wd_packed;
    P.blob = reinterpret_cast<const f4 *>(layers[0].w_packed);
    P.bias2 = layers[0].bias; P.bias3 = layers[1].bias;
    if (fill_wn(P.wn, wn, who) != RTK_OK) return RTK_ERR_INVALID;
    RTK_REQUIRE(wn[2].cout16 == 16, "%s: WeightNet must produce 256 channels", who);
    P.out = out; P.out_pitch = out_pitch;
    P.sv1 = a1; P.sv2 = a2; P.sv3 = a3; P.mk1 = (uint2 *)mask1; P.mk2 = (uint2 *)mask2;
    RTK_REQUIRE(samples <= 65535, "%s: too many samples", who);
    const int groups = (n1 + CV_NW - 1) / CV_NW;
    int gx = 256 * CV_WGS_PER_CU / samples;           // resident workgroups; the rest is looped
    if (gx < 1) gx = 1;
    if (gx > groups) gx = groups;
    P.gx = samples % 8 == 0 ? gx : 0;
    const dim3 grid = P.gx ? dim3(gx * samples) : dim3(gx, samples);
    if (save) cost_volume_kernel<true><<<grid, 64 * CV_NW, 0, (hipStream_t)stream>>>(P);
    else cost_volume_kernel<false><<<grid, 64 * CV_NW, 0, (hipStream_t)stream>>>(P);
    RTK_CHECK_LAUNCH(who);
    return RTK_OK;
}

extern "C" int rtk_cost_volume(int samples, int n1, int n2, const float *xyz1, const float *xyz2, const int64_t *knn_idx,
                               const float *p1, const float *p2, const float *wd_packed, const rtk_layer_t *layers,
                               const rtk_layer_t *wn, float *out, int out_pitch, rtk_stream_t stream) {
    return cost_volume_launch("cost_volume", samples, n1, n2, xyz1, xyz2, knn_idx, p1, p2, wd_packed, layers, wn, out, out_pitch, nullptr,
                              nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int rtk_cost_volume_train(int samples, int n1, int n2, const float *xyz1, const float *xyz2, const int64_t *knn_idx,
                                     const float *p1, const float *p2, const float *wd_packed, const rtk_layer_t *layers,
                                     const rtk_layer_t *wn, float *out, int out_pitch, float *a1, float *a2, float *a3,
                                     void *mask1, void *mask2, rtk_stream_t stream) {
    RTK_REQUIRE(a1 && a2 && a3 && mask1 && mask2, "cost_volume_train: null activation buffer");
    return cost_volume_launch("cost_volume_train", samples, n1, n2, xyz1, xyz2, knn_idx, p1, p2, wd_packed, layers, wn, out, out_pitch, a1,
                              a2, a3, mask1, mask2, stream);
}

// =================================================================================================
// rtk_cost_volume_bwd: backward of rtk_cost_volume_train (include/rtk_train.h).
//
// Same tiling as the forward (one wave = one query point x its 16 neighbours, gradients stationary in registers).  The
// forward saved the three activations a1, a2, a3 (round 2, first half: they were recomputed here -- two more 256 x 256
// products per position, a kernel at 0.55 of the MFMA peak with 184 bytes of scratch per lane); the gradient walks back
// through the two 256x256 layers with the TRANSPOSED packed weights streamed through the LDS double buffer (W3^T, W2^T).
// Weight gradients are contractions over all B*N*16 positions -- plain GEMMs -- whose operands are exactly the saved
// activations and what this kernel materialises, point-major (position, 256): the pre-activation gradients dz1, dz2, dz3
// (+ dq3 for the WeightNet and the 4-vector (dx, dy, dz, 1) of every position); the host multiplies.
// dp1 (gradient of the per-query projection) is the sum of dz1 over the 16 neighbours, reduced in registers.
// =================================================================================================
struct CvBwdParams {
    CvParams f;                   // forward arguments (p1, p2, out unused); f.blob = W3^T, W2^T
    const float *dout;
    int dout_pitch;
    const f4 *wct;                // packed Wc^T (WeightNet last layer transposed): [16][1] fragments
    const float *a3;              // saved by the forward: the last activation ...
    const uint2 *mk1, *mk2;       // ... and the sign masks of the first two (sign_mask)
    float *dz1, *dz2, *dz3, *dq3, *d4, *dp1, *dpd, *dt2;
    float *dbrows;                // optional (queries, 2, 256): per-query sums over the 16 neighbours of dz3 | dz2 (bias gradients = their column sums)
};

__device__ __forceinline__ f4 leaky_grad_bits(f4 d, unsigned bits) {     // d * leaky'(z), [z > 0] in the low four bits
    f4 r;
    r.x = (bits & 1u) ? d.x : 0.1f * d.x;
    r.y = (bits & 2u) ? d.y : 0.1f * d.y;
    r.z = (bits & 4u) ? d.z : 0.1f * d.z;
    r.w = (bits & 8u) ? d.w : 0.1f * d.w;
    return r;
}

__device__ __forceinline__ f4 leaky_grad(f4 d, f4 a) {     // d * leaky'(z), the sign of z read off a = leaky(z)
    f4 r;
    r.x = a.x > 0.f ? d.x : 0.1f * d.x;
    r.y = a.y > 0.f ? d.y : 0.1f * d.y;
    r.z = a.z > 0.f ? d.z : 0.1f * d.z;
    r.w = a.w > 0.f ? d.w : 0.1f * d.w;
    return r;
}

#ifndef CVB_MIN_WAVES
#define CVB_MIN_WAVES ((CV_NW * CV_WGS_PER_CU) / 4)
#endif
__global__ __launch_bounds__(64 * CV_NW, CVB_MIN_WAVES) void cost_volume_bwd_kernel(const CvBwdParams Q) {
    __shared__ __attribute__((aligned(16))) f4 s_w[2 * CVB_F * 64];
    const CvParams &P = Q.f;
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const int wave_in_wg = threadIdx.x >> 6;
    int b, bx, nbx;
    rtk_decode_block(P.gx, b, bx, nbx);
    const int groups = (P.n1 + CV_NW - 1) / CV_NW;
    constexpr int NF = 2 * CV_V * CV_V;
    constexpr int L = CV_V * CV_V;
    WStream<CV_NW, CVB_F, NF> ws;
    ws.start(P.blob, s_w, wave_in_wg, lane);
    for (int G = bx; G < groups; G += nbx) {
        asm volatile("" ::: "memory");
        const int pt = G * CV_NW + wave_in_wg;
        const bool valid = pt < P.n1;
        const long i = (long)b * P.n1 + (valid ? pt : P.n1 - 1);
        const long nb = (long)b * P.n2 + (long)P.knn[i * 16 + j];
        const float bop = g < 3 ? __fsub_rn(P.xyz2[nb * 3 + g], P.xyz1[i * 3 + g]) : 1.0f;
        const long pos = i * 16 + j;                                     // row of the (position, 256) tensors
        // byte offset of this lane's 16-byte slot in a (position, 256) row: one 32-bit VGPR, added to the tensors' uniform
        // base pointers (SGPR-base addressing; 64-bit per-tensor addresses would spill)
        const unsigned ro = (unsigned)pos * 1024u + 16u * g;
        if (valid) Q.d4[pos * 4 + g] = bop;
        const uint2 m2 = Q.mk2[pos * 4 + g], m1 = Q.mk1[pos * 4 + g];     // requested now, used after the first / second product
        f4 h[CV_V], a[CV_V];
#pragma unroll
        for (int v = 0; v < CV_V; ++v) h[v] = *cv_at(Q.a3, ro + 64u * v);       // a3 = leaky(z3)
        // ---- out = sum_k wn * a3:  dz3 = dout wn leaky'(z3),  dq3 = dout a3 [wn > 0] ---------------------------------
        const f4 t2 = weightnet_hidden(P.wn, lane, g, bop);
        const float *dor = Q.dout + i * Q.dout_pitch + 4 * g;
        f4 dt2 = f4_zero();                                              // Wc^T dq3: gradient of the WeightNet's hidden layer
#pragma unroll
        for (int v = 0; v < CV_V; ++v) {
            const f4 w = weightnet_out(P.wn, lane, g, v, t2);
            const f4 d = *reinterpret_cast<const f4 *>(dor + 16 * v);
            f4 q;
            q.x = w.x > 0.f ? d.x * h[v].x : 0.f;
            q.y = w.y > 0.f ? d.y * h[v].y : 0.f;
            q.z = w.z > 0.f ? d.z * h[v].z : 0.f;
            q.w = w.w > 0.f ? d.w * h[v].w : 0.f;
            h[v] = leaky_grad(d * w, h[v]);
            const f4 ft = Q.wct[v * 64 + lane];
            dt2 = mfma4(ft.x, q.x, dt2);
            dt2 = mfma4(ft.y, q.y, dt2);
            dt2 = mfma4(ft.z, q.z, dt2);
            dt2 = mfma4(ft.w, q.w, dt2);
            if (valid) {
                *cv_at(Q.dq3, ro + 64u * v) = q;
                *cv_at(Q.dz3, ro + 64u * v) = h[v];
            }
            if (Q.dbrows) {                          // (uniform) per-query neighbour sum: the host's bias sum shrinks 16x
                f4 r = h[v];
                row_sum16_f4(r);
                if (valid && j == 0) *reinterpret_cast<f4 *>(Q.dbrows + i * 512 + 16 * v + 4 * g) = r;
            }
        }
        if (valid && g < 2) *reinterpret_cast<f4 *>(Q.dt2 + pos * 8 + 4 * g) = dt2;     // 8 hidden units
        // ---- da2 = W3^T dz3;  dz2 = da2 leaky'(z2) ---------------------------------------------------------------------
#pragma unroll
        for (int v = 0; v < CV_V; ++v) a[v] = f4_zero();
        mlp_layer_ws<CV_V, CV_V, 0>(ws, h, a);
#pragma unroll
        for (int v = 0; v < CV_V; ++v) {
            a[v] = leaky_grad_bits(a[v], (v < 8 ? m2.x : m2.y) >> (4 * (v & 7)));
            if (valid) *cv_at(Q.dz2, ro + 64u * v) = a[v];
            if (Q.dbrows) {
                f4 r = a[v];
                row_sum16_f4(r);
                if (valid && j == 0) *reinterpret_cast<f4 *>(Q.dbrows + i * 512 + 256 + 16 * v + 4 * g) = r;
            }
        }
        // ---- da1 = W2^T dz2;  dz1 = da1 leaky'(z1);  dp1 = sum over the 16 neighbours ----------------------------------
#pragma unroll
        for (int v = 0; v < CV_V; ++v) h[v] = f4_zero();
        mlp_layer_ws<CV_V, CV_V, L>(ws, a, h);
        ws.next();   // wrap the stream to chunk 0
        // dz1 -> its store, dp1 (sum over the 16 neighbours) and the per-query partials of dWd = dz1^T d.  Direction components of
        // position j are held by lanes (0,j), (1,j), (2,j) as their MFMA B operand.
        float *dpr = Q.dp1 + i * 256 + 4 * g;
        float *dpd = Q.dpd + i * 768 + 4 * g;
        const float dx = __shfl(bop, j, 64), dy = __shfl(bop, 16 + j, 64), dzc = __shfl(bop, 32 + j, 64);
#pragma unroll
        for (int v = 0; v < CV_V; ++v) {
            f4 r = leaky_grad_bits(h[v], (v < 8 ? m1.x : m1.y) >> (4 * (v & 7)));
            if (valid) *cv_at(Q.dz1, ro + 64u * v) = r;
            f4 rx = r * dx, ry = r * dy, rz = r * dzc;
            row_sum16_f4(r);
            row_sum16_f4(rx);
            row_sum16_f4(ry);
            row_sum16_f4(rz);
            if (valid && j == 0) {
                *reinterpret_cast<f4 *>(dpr + 16 * v) = r;
                *reinterpret_cast<f4 *>(dpd + 16 * v) = rx;
                *reinterpret_cast<f4 *>(dpd + 256 + 16 * v) = ry;
                *reinterpret_cast<f4 *>(dpd + 512 + 16 * v) = rz;
            }
        }
    }
    ws.finish();
}

extern "C" int rtk_cost_volume_bwd(int samples, int n1, int n2, const float *xyz1, const float *xyz2, const int64_t *knn_idx,
                                   const rtk_layer_t *layers_t, const rtk_layer_t *wn, const float *wct_packed, const float *dout,
                                   int dout_pitch, const float *a3, const void *mask1, const void *mask2, float *dz1, float *dz2, float *dz3,
                                   float *dq3, float *d4, float *dp1, float *dpd, float *dt2, float *dbias_rows, rtk_stream_t stream) {
    RTK_REQUIRE(samples > 0 && n1 > 0 && n2 >= 16 && xyz1 && xyz2 && knn_idx && layers_t && dout && mask1 && mask2 && a3 && dz1 && dz2 && dz3 &&
                dq3 && d4 && dp1 && dpd && dt2 && wct_packed, "cost_volume_bwd: bad arguments");
    RTK_REQUIRE((double)samples * n1 * 16.0 * 1024.0 < 4294967296.0, "cost_volume_bwd: more than 4 GiB per (position, 256) tensor "
                "(32-bit row offsets): split the batch");
    RTK_REQUIRE(layers_t[0].cin16 == 16 && layers_t[0].cout16 == 16 && layers_t[1].cin16 == 16 && layers_t[1].cout16 == 16 &&
                layers_t[1].w_packed == layers_t[0].w_packed + 256 * 256, "cost_volume_bwd: expects two contiguous 256x256 layers (W3^T, W2^T)");
    RTK_REQUIRE(dout_pitch % 4 == 0 && dout_pitch >= 256, "cost_volume_bwd: bad dout_pitch");
    RTK_REQUIRE(samples <= 65535, "cost_volume_bwd: too many samples");
    CvBwdParams Q;
    CvParams &P = Q.f;
    P.samples = samples; P.n1 = n1; P.n2 = n2;
    P.xyz1 = xyz1; P.xyz2 = xyz2; P.knn = knn_idx; P.p1 = nullptr; P.p2 = nullptr; P.wd = nullptr;
    P.blob = reinterpret_cast<const f4 *>(layers_t[0].w_packed);
    P.bias2 = nullptr; P.bias3 = nullptr;
    if (fill_wn(P.wn, wn, "cost_volume_bwd") != RTK_OK) return RTK_ERR_INVALID;
    RTK_REQUIRE(wn[2].cout16 == 16, "cost_volume_bwd: WeightNet must produce 256 channels");
    P.out = nullptr; P.out_pitch = 0; P.sv1 = P.sv2 = P.sv3 = nullptr; P.mk1 = P.mk2 = nullptr;
    Q.dout = dout; Q.dout_pitch = dout_pitch;
    Q.wct = reinterpret_cast<const f4 *>(wct_packed);
    Q.a3 = a3; Q.mk1 = (const uint2 *)mask1; Q.mk2 = (const uint2 *)mask2; Q.dz1 = dz1; Q.dz2 = dz2; Q.dz3 = dz3; Q.dq3 = dq3; Q.d4 = d4; Q.dp1 = dp1; Q.dpd = dpd; Q.dt2 = dt2;
    Q.dbrows = dbias_rows;
    const int groups = (n1 + CV_NW - 1) / CV_NW;
    int gx = 256 * CV_WGS_PER_CU / samples;
    if (gx < 1) gx = 1;
    if (gx > groups) gx = groups;
    P.gx = samples % 8 == 0 ? gx : 0;
    const dim3 grid = P.gx ? dim3(gx * samples) : dim3(gx, samples);
    cost_volume_bwd_kernel<<<grid, 64 * CV_NW, 0, (hipStream_t)stream>>>(Q);
    RTK_CHECK_LAUNCH("cost_volume_bwd");
    return RTK_OK;
}

// dst[b][idx[b][m]][:] += src[b][m][:]  -- the scatter half of a gather's backward (dp2 of the cost volume: every
// neighbour row collects the dz1 of the positions that gathered it).  One workgroup owns (sample, 32-channel slab):
// the slab of dst is accumulated in LDS with ds_add_f32 and written once -- no global atomics, dst needs no zero-fill.
#define SC_CH 32
__global__ __launch_bounds__(256) void scatter_rows_kernel(int m, int n, int channels, const int64_t *__restrict__ idx,
                                                           const float *__restrict__ src, float *__restrict__ dst) {
    extern __shared__ float s_acc[];                  // [n][SC_CH]
    const int b = blockIdx.y, c0 = blockIdx.x * SC_CH;
    for (int e = threadIdx.x; e < n * SC_CH; e += 256) s_acc[e] = 0.f;
    __syncthreads();
    const int cl = threadIdx.x & (SC_CH - 1), rsub = threadIdx.x / SC_CH;      // 8 rows per pass
    const int64_t *ib = idx + (size_t)b * m;
    const float *sb = src + (size_t)b * m * channels + c0 + cl;
    constexpr int RP = 256 / SC_CH, UN = 8;                                     // UN independent loads in flight per thread
    int r = rsub;
    for (; r + (UN - 1) * RP < m; r += UN * RP) {
        int t[UN];
        float v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            t[u] = (int)ib[r + u * RP];
            v[u] = sb[(size_t)(r + u * RP) * channels];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) atomicAdd(&s_acc[t[u] * SC_CH + cl], v[u]);
    }
    for (; r < m; r += RP) atomicAdd(&s_acc[(int)ib[r] * SC_CH + cl], sb[(size_t)r * channels]);
    __syncthreads();
    float *db = dst + (size_t)b * n * channels + c0;
    for (int e = threadIdx.x; e < n * SC_CH; e += 256) db[(size_t)(e / SC_CH) * channels + (e % SC_CH)] = s_acc[e];
}

// 256-channel variant, partitioned by DESTINATION: one workgroup owns 32 destination rows of one sample (32 KiB of LDS
// accumulators, thread = channel, so no atomics at all) and walks the index list once; matching source rows are compacted
// per 256-entry chunk (ballot + prefix, position order) and fetched as full coalesced 1 KiB rows, four in flight.
// Deterministic: every destination element is accumulated by one thread in position order.
// SC_RB = 32 destination rows per workgroup at large batches; 8 at small ones (B = 1, the reference's regime: 12 workgroups of 32 rows
// leave the chip empty and one hot row -- real frames have exact duplicate points -- holds a whole workgroup up: 150 us on the real
// pairs against 66 us on synthetic ones; every workgroup scans the whole index list either way, which is cheap at these sizes).
template <int SC_RB>
__global__ __launch_bounds__(256) void scatter_rows256_kernel(int m, int n, const int64_t *__restrict__ idx,
                                                              const float *__restrict__ src, float *__restrict__ dst) {
    __shared__ float s_acc[SC_RB][256];
    __shared__ int s_list[256];
    __shared__ int s_cnt[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, r0 = blockIdx.x * SC_RB;
#pragma unroll
    for (int r = 0; r < SC_RB; ++r) s_acc[r][tid] = 0.f;
    const int64_t *ib = idx + (size_t)b * m;
    const float *sb = src + (size_t)b * m * 256 + tid;
    for (int base = 0; base < m; base += 256) {
        const int p = base + tid;
        const int t = p < m ? (int)ib[p] - r0 : -1;
        const bool hit = t >= 0 && t < SC_RB;
        const unsigned long long bal = __ballot(hit);
        if (lane == 0) s_cnt[wave] = __popcll(bal);
        __syncthreads();
        int off = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int c = s_cnt[w];
            off += w < wave ? c : 0;
            total += c;
        }
        if (hit) s_list[off + __popcll(bal & ((1ull << lane) - 1ull))] = tid | (t << 8);
        __syncthreads();
        int q = 0;
        for (; q + 16 <= total; q += 16) {     // sixteen 1 KiB rows in flight (a chunk holds ~32 hits: two round trips; with four
            int e[16];                          // in flight the kernel ran at 1.9 TB/s, with eight at 3.2)
            float v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) e[k] = s_list[q + k];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = sb[(size_t)(base + (e[k] & 255)) * 256];
#pragma unroll
            for (int k = 0; k < 16; ++k) s_acc[e[k] >> 8][tid] += v[k];
        }
        for (; q + 4 <= total; q += 4) {
            const int e0 = s_list[q], e1 = s_list[q + 1], e2 = s_list[q + 2], e3 = s_list[q + 3];
            const float v0 = sb[(size_t)(base + (e0 & 255)) * 256], v1 = sb[(size_t)(base + (e1 & 255)) * 256];
            const float v2 = sb[(size_t)(base + (e2 & 255)) * 256], v3 = sb[(size_t)(base + (e3 & 255)) * 256];
            s_acc[e0 >> 8][tid] += v0;
            s_acc[e1 >> 8][tid] += v1;
            s_acc[e2 >> 8][tid] += v2;
            s_acc[e3 >> 8][tid] += v3;
        }
        for (; q < total; ++q) {
            const int e = s_list[q];
            s_acc[e >> 8][tid] += sb[(size_t)(base + (e & 255)) * 256];
        }
        __syncthreads();
    }
    float *db = dst + ((size_t)b * n + r0) * 256 + tid;
    for (int r = 0; r < SC_RB && r0 + r < n; ++r) db[(size_t)r * 256] = s_acc[r][tid];
}

extern "C" int rtk_scatter_add_rows(int samples, int m, int n, int channels, const int64_t *idx, const float *src, float *dst,
                                    rtk_stream_t stream) {
    RTK_REQUIRE(samples > 0 && m > 0 && n > 0 && channels > 0 && channels % SC_CH == 0 && idx && src && dst,
                "scatter_add_rows: bad arguments (channels must be a multiple of %d)", SC_CH);
    RTK_REQUIRE(channels == 256 || (size_t)n * SC_CH * 4 <= 128 * 1024, "scatter_add_rows: n (%d) too large for the LDS slab", n);
    RTK_REQUIRE(samples <= 65535, "scatter_add_rows: too many samples");
    if (channels == 256) {
        if ((long)samples * ((n + 31) / 32) >= 256)
            scatter_rows256_kernel<32><<<dim3((n + 31) / 32, samples), 256, 0, (hipStream_t)stream>>>(m, n, idx, src, dst);
        else
            scatter_rows256_kernel<8><<<dim3((n + 7) / 8, samples), 256, 0, (hipStream_t)stream>>>(m, n, idx, src, dst);
        RTK_CHECK_LAUNCH("scatter_add_rows");
        return RTK_OK;
    }
    const size_t lds = (size_t)n * SC_CH * 4;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(scatter_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        RTK_REQUIRE(e == hipSuccess, "scatter_add_rows: cannot raise the LDS limit: %s", hipGetErrorString(e));
    }
    scatter_rows_kernel<<<dim3(channels / SC_CH, samples), 256, lds, (hipStream_t)stream>>>(m, n, channels, idx, src, dst);
    RTK_CHECK_LAUNCH("scatter_add_rows");
    return RTK_OK;
}

// =================================================================================================
// rtk_patch_cost
// =================================================================================================
struct PcParams {
    int samples, n;
    const float *xyz;
    const int64_t *knn;
    const float *feat;
    int feat_pitch;
    WnWeights wn;
    float *out;
    int out_pitch, out_cm;
    int gx;                 // > 0: XCD-aware 1-D grid (rtk_decode_block)
};

// Register budget: five waves per SIMD and the block loop NOT unrolled (58 registers; fully unrolled hipcc hoists all sixteen row
// loads and WeightNet fragments to the top: 200 registers, two waves per SIMD).  69.9 -> 54.6 us alone, and the kernel fits on a
// SIMD next to another batch's forward cost volume: +1.5 % frame-pairs/s.
#ifndef PC_WAVES
#define PC_WAVES 5
#endif
#ifndef PC_WGS_TARGET
#define PC_WGS_TARGET 4096
#endif
__global__ __launch_bounds__(256, PC_WAVES) void patch_cost_kernel(const PcParams P) {
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    int b, bx, nbx;                                                 // one sample per workgroup, its points strided
    rtk_decode_block(P.gx, b, bx, nbx);
    for (int pt = bx * 4 + (threadIdx.x >> 6); pt < P.n; pt += nbx * 4) {
        const long i = (long)b * P.n + pt;
        const long nb = (long)b * P.n + (long)P.knn[i * 16 + j];
        const float bop = g < 3 ? __fsub_rn(P.xyz[nb * 3 + g], P.xyz[i * 3 + g]) : 1.0f;
        const f4 t2 = weightnet_hidden(P.wn, lane, g, bop);
        const float *fr = P.feat + nb * P.feat_pitch + 4 * g;
        // four 16-channel blocks at a time: their neighbour sums by ONE transposing reduction (fused_common.h: 32 cross-lane operations
        // for the four instead of 4 x 16), lane j ends with block 4 vg + 2 (bit 3 of j) + (bit 2 of j) and the first lane of each quad
        // stores.  (The whole row up front: 122 vs 71 us -- registers; the group loop stays rolled.)
        const int vq = 2 * ((j >> 3) & 1) + ((j >> 2) & 1);
#pragma unroll 1
        for (int vg = 0; vg < CV_V / 4; ++vg) {
            f4 r[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f4 w = weightnet_out(P.wn, lane, g, 4 * vg + q, t2);
                const f4 f = *reinterpret_cast<const f4 *>(fr + 16 * (4 * vg + q));
                r[q] = w * f;                       // (a product the compiler sees: it places the wait states the MFMA result needs)
            }
            const f4 t = row_sum16_transpose4(r[0], r[1], r[2], r[3]);
            if ((j & 3) == 0) {
                const int v = 4 * vg + vq;
                if (!P.out_cm) {
                    *reinterpret_cast<f4 *>(P.out + i * P.out_pitch + 16 * v + 4 * g) = t;
                } else {
                    float *o = P.out + ((long)b * 256 + 16 * v + 4 * g) * P.n + (i - (long)b * P.n);
                    o[0] = t.x; o[P.n] = t.y; o[2 * (long)P.n] = t.z; o[3 * (long)P.n] = t.w;
                }
            }
        }
    }
}

extern "C" int rtk_patch_cost(int samples, int n, const float *xyz, const int64_t *knn_idx, const float *feat,
                              int feat_pitch, const rtk_layer_t *wn, float *out, int out_pitch, int out_channel_major,
                              rtk_stream_t stream) {
    RTK_REQUIRE(samples > 0 && n >= 16 && xyz && knn_idx && feat && out && feat_pitch % 4 == 0 && feat_pitch >= 256,
                "patch_cost: bad arguments");
    RTK_REQUIRE(out_channel_major || (out_pitch % 4 == 0 && out_pitch >= 256), "patch_cost: bad out_pitch");
    PcParams P;
    P.samples = samples; P.n = n; P.xyz = xyz; P.knn = knn_idx; P.feat = feat; P.feat_pitch = feat_pitch;
    if (fill_wn(P.wn, wn, "patch_cost") != RTK_OK) return RTK_ERR_INVALID;
    RTK_REQUIRE(wn[2].cout16 == 16, "patch_cost: WeightNet must produce 256 channels");
    P.out = out; P.out_pitch = out_pitch; P.out_cm = out_channel_major;
    RTK_REQUIRE(samples <= 65535, "patch_cost: too many samples");
    int gx = (n + 3) / 4;
    while ((long)gx * samples > PC_WGS_TARGET && gx > 1) gx = (gx + 1) / 2;
    P.gx = samples % 8 == 0 ? gx : 0;
    patch_cost_kernel<<<P.gx ? dim3(gx * samples) : dim3(gx, samples), 256, 0, (hipStream_t)stream>>>(P);
    RTK_CHECK_LAUNCH("patch_cost");
    return RTK_OK;
}

// Backward of rtk_patch_cost: out[i] = sum_k wn(d_ik) * feat[knn[i,k]].  Per position (i,k), point-major (M,256):
//   dxg = dout[i] * wn          -> scattered onto feat's rows by the caller (rtk_scatter_add_rows)
//   dq3 = dout[i] * feat[nb] * [wn > 0]   (gradient of the WeightNet's last pre-activation), dt2 = Wc^T dq3, d4 = (d, 1).
struct PcBwdParams {
    PcParams f;
    const float *dout;
    int dout_pitch;
    const f4 *wct;
    float *dxg, *dq3, *dt2, *d4;      // dxg optional (NULL: the feature gradient is gathered by rtk_patch_dfeat_gather instead)
    float *t2;                        // optional (M, 8): the WeightNet's hidden activation of every position, for that gather
};

__global__ __launch_bounds__(256) void patch_cost_bwd_kernel(const PcBwdParams Q) {
    const PcParams &P = Q.f;
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    int b, bx, nbx;
    rtk_decode_block(P.gx, b, bx, nbx);
    for (int pt = bx * 4 + (threadIdx.x >> 6); pt < P.n; pt += nbx * 4) {
        const long i = (long)b * P.n + pt;
        const long nb = (long)b * P.n + (long)P.knn[i * 16 + j];
        const float bop = g < 3 ? __fsub_rn(P.xyz[nb * 3 + g], P.xyz[i * 3 + g]) : 1.0f;
        const long pos = i * 16 + j;
        Q.d4[pos * 4 + g] = bop;
        const f4 t2 = weightnet_hidden(P.wn, lane, g, bop);
        if (Q.t2 && g < 2) *reinterpret_cast<f4 *>(Q.t2 + pos * 8 + 4 * g) = t2;      // rows 4g .. 4g+3 = hidden units, column j = position
        const float *fr = P.feat + nb * P.feat_pitch + 4 * g;
        const float *dor = Q.dout + i * Q.dout_pitch + 4 * g;
        f4 dt2 = f4_zero();
        // both rows requested in full before the first use (32 loads in flight per lane; one pair per iteration was a round trip each)
        f4 fv[CV_V], dv[CV_V];
#pragma unroll
        for (int v = 0; v < CV_V; ++v) {
            fv[v] = *reinterpret_cast<const f4 *>(fr + 16 * v);
            dv[v] = *reinterpret_cast<const f4 *>(dor + 16 * v);
        }
#pragma unroll
        for (int v = 0; v < CV_V; ++v) {
            const f4 w = weightnet_out(P.wn, lane, g, v, t2);
            const f4 f = fv[v], d = dv[v];
            f4 q;
            q.x = w.x > 0.f ? d.x * f.x : 0.f;
            q.y = w.y > 0.f ? d.y * f.y : 0.f;
            q.z = w.z > 0.f ? d.z * f.z : 0.f;
            q.w = w.w > 0.f ? d.w * f.w : 0.f;
            const f4 ft = Q.wct[v * 64 + lane];
            dt2 = mfma4(ft.x, q.x, dt2);
            dt2 = mfma4(ft.y, q.y, dt2);
            dt2 = mfma4(ft.z, q.z, dt2);
            dt2 = mfma4(ft.w, q.w, dt2);
            *reinterpret_cast<f4 *>(Q.dq3 + pos * 256 + 16 * v + 4 * g) = q;
            if (Q.dxg) *reinterpret_cast<f4 *>(Q.dxg + pos * 256 + 16 * v + 4 * g) = d * w;
        }
        if (g < 2) *reinterpret_cast<f4 *>(Q.dt2 + pos * 8 + 4 * g) = dt2;
    }
}

extern "C" int rtk_patch_cost_bwd(int samples, int n, const float *xyz, const int64_t *knn_idx, const float *feat, int feat_pitch,
                                  const rtk_layer_t *wn, const float *wct_packed, const float *dout, int dout_pitch, float *dxg,
                                  float *dq3, float *dt2, float *d4, float *t2, rtk_stream_t stream) {
    RTK_REQUIRE(samples > 0 && n >= 16 && xyz && knn_idx && feat && dout && (dxg || t2) && dq3 && dt2 && d4 && wct_packed &&
                feat_pitch % 4 == 0 && feat_pitch >= 256 && dout_pitch % 4 == 0 && dout_pitch >= 256, "patch_cost_bwd: bad arguments");
    PcBwdParams Q;
    PcParams &P = Q.f;
    P.samples = samples; P.n = n; P.xyz = xyz; P.knn = knn_idx; P.feat = feat; P.feat_pitch = feat_pitch;
    if (fill_wn(P.wn, wn, "patch_cost_bwd") != RTK_OK) return RTK_ERR_INVALID;
    RTK_REQUIRE(wn[2].cout16 == 16, "patch_cost_bwd: WeightNet must produce 256 channels");
    P.out = nullptr; P.out_pitch = 0; P.out_cm = 0;
    Q.dout = dout; Q.dout_pitch = dout_pitch; Q.wct = reinterpret_cast<const f4 *>(wct_packed);
    Q.dxg = dxg; Q.dq3 = dq3; Q.dt2 = dt2; Q.d4 = d4; Q.t2 = t2;
    RTK_REQUIRE(samples <= 65535, "patch_cost_bwd: too many samples");
    int gx = (n + 3) / 4;
    while ((long)gx * samples > 4096 && gx > 1) gx = (gx + 1) / 2;
    P.gx = samples % 8 == 0 ? gx : 0;
    patch_cost_bwd_kernel<<<P.gx ? dim3(gx * samples) : dim3(gx, samples), 256, 0, (hipStream_t)stream>>>(Q);
    RTK_CHECK_LAUNCH("patch_cost_bwd");
    return RTK_OK;
}

