// fused_pointwise.hip -- rtk_pointwise_mlp: a chain of up to four 1x1-conv (+folded BN, +activation)
// layers applied per point, with the input vector assembled on the fly from
//   [three-NN inverse-distance interpolation of a coarser level]  ||  skip features  ||  per-sample
//   (broadcast) features,
// i.e. the reference's PointnetFPModule.forward (lib/pointnet2_modules.py:140-158), the nn.Linear
// bottlenecks of PNHead (model_utils.py:414-418) and the Flow/Cls predictors (model_utils.py:308-357),
// each as ONE kernel instead of 4-12 framework ops with materialised intermediates.
//
// Structure: see fused_common.h.  A wave owns 16 points; all layers run back to back on the MFMA
// pipe with activations held in registers; HBM sees each input row once and each output row once.
#include <string.h>

#include "rtk_common.h"
#include "fused_common.h"
#include "rtk_fused.h"

struct PwParams {
    int rows, rows_per_sample;
    rtk_interp_t interp;
    int nsrc;
    rtk_src_t src[RTK_MAX_SRC];
    const float *sample_bias;
    rtk_layer_t layer[RTK_MAX_LAYERS];
    float *out;
    int out_pitch, out_channels, out_cm;
    const int *row_nuniq;
    float *colmax;
    int gx;                 // > 0: XCD-aware 1-D grid (rtk_decode_block)
};

template <int V>
__device__ __forceinline__ void init_bias(f4 (&acc)[V], const float *__restrict__ bias, int g) {
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = bias_frag(bias, v, g);
}
template <int V>
__device__ __forceinline__ void init_zero(f4 (&acc)[V]) {
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = f4_zero();
}
// split layers: y = acc c + b, c = this lane's 2^-(kw + kx)
template <int V>
__device__ __forceinline__ void scale_bias(f4 (&acc)[V], float c, const float *__restrict__ bias, int g) {
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = __builtin_elementwise_fma(acc[v], (f4){c, c, c, c}, bias_frag(bias, v, g));
}

// One layer of the chain: fp32-input MFMA with the bias in the accumulators, or (SPLIT) the fp16 split path -- the lane's power-of-two
// scale from its position's largest input, zero accumulators, scale and bias in one fma afterwards.
template <bool SPLIT, int U, int V, int FBASE, class WS>
__device__ __forceinline__ void pw_layer(WS &ws, const f4 (&h)[U], f4 (&acc)[V], const rtk_layer_t &L, int g, const float *sample_bias, bool first) {
    if constexpr (SPLIT) {
        const LaneScale sc = lane_scale16(h);
        init_zero<V>(acc);
        if (first) ws.next_if_deferred();
        mlp_layer_ws_split<U, V, FBASE>(ws, h, sc.s, acc);
        scale_bias<V>(acc, sc.inv * L.inv_scale, L.bias, g);
    } else {
        init_bias<V>(acc, L.bias, g);
        if (first) ws.next_if_deferred();
        mlp_layer_ws<U, V, FBASE>(ws, h, acc);
    }
    if (sample_bias) {
#pragma unroll
        for (int v = 0; v < V; ++v) acc[v] += bias_frag(sample_bias, v, g);
    }
    apply_act<V>(acc, L.act);
}

template <int V>
__device__ __forceinline__ void store_tile(const PwParams &P, const f4 (&acc)[V], int p, int b, int g, bool valid) {
    if (P.colmax) {
        // per-sample column maximum: DPP max over the tile's 16 rows, then one atomic max per channel on the float bits
        // (outputs are >= 0 after ReLU, so unsigned order == float order and the zero-initialised buffer is the identity)
        unsigned *cm = reinterpret_cast<unsigned *>(P.colmax) + (size_t)b * 16 * V + 4 * g;
        const bool lead = (threadIdx.x & 15) == 0;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            f4 m = valid ? acc[v] : f4_zero();
            row_max16_f4(m);
            if (lead) {
                atomicMax(cm + 16 * v + 0, __float_as_uint(m.x));
                atomicMax(cm + 16 * v + 1, __float_as_uint(m.y));
                atomicMax(cm + 16 * v + 2, __float_as_uint(m.z));
                atomicMax(cm + 16 * v + 3, __float_as_uint(m.w));
            }
        }
    }
    if (!valid) return;
    if (!P.out_cm) {
        float *o = P.out + (size_t)p * P.out_pitch;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int c = 16 * v + 4 * g;
            if (c + 3 < P.out_channels) {
                *reinterpret_cast<f4 *>(o + c) = acc[v];
            } else {
                if (c + 0 < P.out_channels) o[c + 0] = acc[v].x;
                if (c + 1 < P.out_channels) o[c + 1] = acc[v].y;
                if (c + 2 < P.out_channels) o[c + 2] = acc[v].z;
            }
        }
    } else {
        const int n = P.rows_per_sample;
        float *o = P.out + (size_t)b * P.out_channels * n + (p - b * n);
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int c = 16 * v + 4 * g;
            if (c + 0 < P.out_channels) o[(size_t)(c + 0) * n] = acc[v].x;
            if (c + 1 < P.out_channels) o[(size_t)(c + 1) * n] = acc[v].y;
            if (c + 2 < P.out_channels) o[(size_t)(c + 2) * n] = acc[v].z;
            if (c + 3 < P.out_channels) o[(size_t)(c + 3) * n] = acc[v].w;
        }
    }
}

#define PW_NW 4    // waves per workgroup
#ifndef PW_WGS_TARGET
#define PW_WGS_TARGET 256      // workgroups per launch: about one per CU, the rest is looped.  Round 6, same box, alternating (ab_knobs.py):
                               // 512 (rounds 2-5): base; 448 / 384 / 320: +0.4 ... +0.5 %; 256: +0.9 % (B=64, N=256), +1.0 % (B=32, N=1024), +-0.1 %
                               // (B=32 / 8 / 1, N=256); 192: -0.5 %; 128: -0.1 %
#endif
#ifndef PW_F
#define PW_F 16    // fragments (KiB) per half of the LDS weight double buffer (16 vs 32: same kernel speed, 1 % more end-to-end
                   // throughput with two batches in flight -- smaller footprints co-reside, tools/experiments/exp_pwf.sh)
#endif

// SPLIT: the layers on the fp16 matrix pipe (two pieces per operand, three products per fp32 product, a power-of-two scale per weight
// matrix and per position: fused_common.h, split_mfma.h): same tile, registers and stream, the blob holds split images.
template <int U, int V1, int V2, int V3, int V4, bool INTERP, bool SPLIT>
__global__ __launch_bounds__(64 * PW_NW, 2) void pointwise_mlp_kernel(const PwParams P) {
    __shared__ __attribute__((aligned(16))) f4 s_w[2 * PW_F * 64];
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const int wave_in_wg = threadIdx.x >> 6;
    // A workgroup owns one sample and strides over its 64-row groups (one 16-row tile per wave).  No division inside the
    // loop; a tile never straddles two samples; workgroups whose groups are all duplicate rows (>= row_nuniq[b]) exit
    // before touching the weight stream.  The trip count is uniform over the workgroup, which the barriers inside the
    // weight stream require.
    int b, bx, nbx;
    rtk_decode_block(P.gx, b, bx, nbx);
    const int rps = P.rows_per_sample;
    const int live_rows = P.row_nuniq ? min(rps, __builtin_amdgcn_readfirstlane(P.row_nuniq[b])) : rps;
    const int live_groups = (live_rows + PW_NW * 16 - 1) / (PW_NW * 16);
    if (bx >= live_groups) return;
    constexpr int F1 = SPLIT ? split16_nf(U, V1) : U * V1, F2 = SPLIT ? split16_nf(V1, V2) : V1 * V2,
                  F3 = SPLIT ? split16_nf(V2, V3) : V2 * V3, F4 = SPLIT ? split16_nf(V3, V4) : V3 * V4;
    constexpr int NF = F1 + F2 + F3 + F4;
    // A blob of several chunks: the stream is entered at the top of every tile AFTER the tile's input loads have been issued (the
    // wrap to chunk 0 -- for the first tile: the arrival of chunk 0 -- and those loads then share one round trip; most workgroups
    // have exactly one tile, and the two round trips in a row were 1-2 us of a 10-25 us launch).
    WStream<PW_NW, PW_F, NF> ws;
    if constexpr (NF > PW_F) ws.start_deferred(reinterpret_cast<const f4 *>(P.layer[0].w_packed), s_w, wave_in_wg, lane);
    else ws.start(reinterpret_cast<const f4 *>(P.layer[0].w_packed), s_w, wave_in_wg, lane);

    // 16-channel slot where each segment starts (wave-uniform)
    int ustart[RTK_MAX_SRC + 1];
    ustart[0] = INTERP ? (P.interp.channels + 15) >> 4 : 0;
#pragma unroll
    for (int s = 0; s < RTK_MAX_SRC; ++s) ustart[s + 1] = ustart[s] + (s < P.nsrc ? (P.src[s].channels + 15) >> 4 : 0);
    const int uend = ustart[RTK_MAX_SRC];

    for (int G = bx; G < live_groups; G += nbx) {
        asm volatile("" ::: "memory");   // keep loop-invariant loads/addresses inside the loop (registers are the scarce resource)
        const int r = G * (PW_NW * 16) + wave_in_wg * 16 + j;           // row within the sample
        const bool valid = r < live_rows;
        const int p = b * rps + (r < rps ? r : rps - 1);                  // global row (clamped: out-of-range lanes compute garbage, never stored)

        f4 h[U];
        // ---- segment 0 (optional): three-NN interpolation, lib/pointnet2_modules.py:141-146 -------------
        if (INTERP) {
            const int *idp = P.interp.idx + (size_t)p * 3;
            int id[3] = {idp[0], idp[1], idp[2]};
            if (P.interp.nuniq) {       // known rows beyond the sample's unique count are copies of its row 0
                const int e = P.interp.nuniq[b];
                id[0] = id[0] < e ? id[0] : 0; id[1] = id[1] < e ? id[1] : 0; id[2] = id[2] < e ? id[2] : 0;
            }
            const float *d2 = P.interp.dist2 + (size_t)p * 3;
            const float r0 = __fdiv_rn(1.0f, __fadd_rn(__fsqrt_rn(d2[0]), 1e-8f));
            const float r1 = __fdiv_rn(1.0f, __fadd_rn(__fsqrt_rn(d2[1]), 1e-8f));
            const float r2 = __fdiv_rn(1.0f, __fadd_rn(__fsqrt_rn(d2[2]), 1e-8f));
            const float norm = __fadd_rn(__fadd_rn(r0, r1), r2);
            const float w0 = __fdiv_rn(r0, norm), w1 = __fdiv_rn(r1, norm), w2 = __fdiv_rn(r2, norm);
            const float *k0 = P.interp.known_feats + ((size_t)b * P.interp.m + id[0]) * P.interp.pitch;
            const float *k1 = P.interp.known_feats + ((size_t)b * P.interp.m + id[1]) * P.interp.pitch;
            const float *k2 = P.interp.known_feats + ((size_t)b * P.interp.m + id[2]) * P.interp.pitch;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (u < ustart[0]) {
                    // unconditional loads (lanes past the last channel re-read channel 0 and discard): a load under a per-lane
                    // condition is a basic block of its own and the compiler waits for it before issuing the next one
                    const int c = 16 * u + 4 * g;
                    const bool ok = c < P.interp.channels;
                    const int cc = ok ? c : 0;
                    const f4 a0 = *reinterpret_cast<const f4 *>(k0 + cc);
                    const f4 a1 = *reinterpret_cast<const f4 *>(k1 + cc);
                    const f4 a2 = *reinterpret_cast<const f4 *>(k2 + cc);
                    // interpolate_gpu.cu:168  w0*p0 + w1*p1 + w2*p2  (nvcc: fma(w2,p2,fma(w1,p1,w0*p0)))
                    f4 v;
                    v.x = __fmaf_rn(w2, a2.x, __fmaf_rn(w1, a1.x, __fmul_rn(w0, a0.x)));
                    v.y = __fmaf_rn(w2, a2.y, __fmaf_rn(w1, a1.y, __fmul_rn(w0, a0.y)));
                    v.z = __fmaf_rn(w2, a2.z, __fmaf_rn(w1, a1.z, __fmul_rn(w0, a0.z)));
                    v.w = __fmaf_rn(w2, a2.w, __fmaf_rn(w1, a1.w, __fmul_rn(w0, a0.w)));
                    h[u] = ok ? v : f4_zero();
                }
            }
        }
        // ---- plain / per-sample segments --------------------------------------------------------------
        // per-lane row base of every source (row = point, or sample for broadcast sources), then one
        // load per 16-channel slot from the source that owns it (selection is wave-uniform)
        const float *rowp[RTK_MAX_SRC];
        const float *dummy = reinterpret_cast<const float *>(P.layer[0].w_packed);      // any readable 16-byte aligned address
#pragma unroll
        for (int q = 0; q < RTK_MAX_SRC; ++q)
            rowp[q] = q < P.nsrc ? P.src[q].ptr + (P.src[q].per_sample ? (size_t)b : (size_t)p) * P.src[q].pitch + 4 * g : nullptr;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!INTERP || u >= ustart[0]) {
                const float *rp = rowp[0];
                int us = ustart[0], ch = P.src[0].channels;
#pragma unroll
                for (int q = 1; q < RTK_MAX_SRC; ++q)
                    if (q < P.nsrc && u >= ustart[q]) { rp = rowp[q]; us = ustart[q]; ch = P.src[q].channels; }
                const int c = 16 * (u - us);
                const bool ok = u < uend && c + 4 * g < ch;
                const f4 v = *reinterpret_cast<const f4 *>(ok ? rp + c : dummy);       // unconditional load, see above
                h[u] = ok ? v : f4_zero();
            }
        }

        // ---- layer chain ------------------------------------------------------------------------------
        // (the first layer enters the weight stream: chunk 0's request has been travelling with the loads above)
        f4 a1[V1];
        pw_layer<SPLIT, U, V1, 0>(ws, h, a1, P.layer[0], g, P.sample_bias ? P.sample_bias + (size_t)b * 16 * V1 : nullptr, true);
        if constexpr (V2 == 0) {
            store_tile<V1>(P, a1, p, b, g, valid);
        } else {
            f4 a2[V2];
            pw_layer<SPLIT, V1, V2, F1>(ws, a1, a2, P.layer[1], g, nullptr, false);
            if constexpr (V3 == 0) {
                store_tile<V2>(P, a2, p, b, g, valid);
            } else {
                f4 a3[V3];
                pw_layer<SPLIT, V2, V3, F1 + F2>(ws, a2, a3, P.layer[2], g, nullptr, false);
                if constexpr (V4 == 0) {
                    store_tile<V3>(P, a3, p, b, g, valid);
                } else {
                    f4 a4[V4];
                    pw_layer<SPLIT, V3, V4, F1 + F2 + F3>(ws, a3, a4, P.layer[3], g, nullptr, false);
                    store_tile<V4>(P, a4, p, b, g, valid);
                }
            }
        }
    }
    ws.finish();
}

template <int U, int V1, int V2, int V3, int V4>
static int launch_pw(const PwParams &P0, bool interp, bool split, hipStream_t s) {
    PwParams P = P0;
    const int samples = P.rows / P.rows_per_sample;
    const int groups = (P.rows_per_sample + PW_NW * 16 - 1) / (PW_NW * 16);
    int gx = PW_WGS_TARGET / samples;           // (1024: 5 % slower end to end)
    if (gx < 1) gx = 1;
    if (gx > groups) gx = groups;
    P.gx = samples % 8 == 0 ? gx : 0;
    const dim3 blocks = P.gx ? dim3(gx * samples) : dim3(gx, samples);
    if (interp && split) pointwise_mlp_kernel<U, V1, V2, V3, V4, true, true><<<blocks, 256, 0, s>>>(P);
    else if (interp) pointwise_mlp_kernel<U, V1, V2, V3, V4, true, false><<<blocks, 256, 0, s>>>(P);
    else if (split) pointwise_mlp_kernel<U, V1, V2, V3, V4, false, true><<<blocks, 256, 0, s>>>(P);
    else pointwise_mlp_kernel<U, V1, V2, V3, V4, false, false><<<blocks, 256, 0, s>>>(P);
    return 0;
}

extern "C" int rtk_pointwise_mlp(int rows, int rows_per_sample, const rtk_interp_t *interp, int nsrc,
                                 const rtk_src_t *srcs, const float *sample_bias, int nlayers,
                                 const rtk_layer_t *layers, float *out, int out_pitch, int out_channels,
                                 int out_channel_major, const int *row_nuniq, float *colmax, rtk_stream_t stream) {
    RTK_REQUIRE(rows > 0 && rows_per_sample > 0 && rows % rows_per_sample == 0 && rows / rows_per_sample <= 65535,
                "pointwise_mlp: bad row counts (%d, %d)", rows, rows_per_sample);
    RTK_REQUIRE(nsrc >= 0 && nsrc <= RTK_MAX_SRC && (nsrc == 0 || srcs), "pointwise_mlp: nsrc=%d", nsrc);
    RTK_REQUIRE(nlayers >= 1 && nlayers <= RTK_MAX_LAYERS && layers && out, "pointwise_mlp: nlayers=%d", nlayers);
    PwParams P;
    memset(&P, 0, sizeof(P));
    P.rows = rows;
    P.rows_per_sample = rows_per_sample;
    int U = 0;
    if (interp) {
        RTK_REQUIRE(interp->known_feats && interp->idx && interp->dist2 && interp->pitch % 4 == 0 && interp->channels % 4 == 0,
                    "pointwise_mlp: bad interp segment");
        P.interp = *interp;
        U += (interp->channels + 15) / 16;
    }
    P.nsrc = nsrc;
    for (int s = 0; s < nsrc; ++s) {
        RTK_REQUIRE(srcs[s].ptr && srcs[s].pitch % 4 == 0 && srcs[s].channels > 0 && srcs[s].pitch >= ((srcs[s].channels + 3) / 4) * 4,
                    "pointwise_mlp: bad source %d (pitch %d, channels %d)", s, srcs[s].pitch, srcs[s].channels);
        P.src[s] = srcs[s];
        U += (srcs[s].channels + 15) / 16;
    }
    P.sample_bias = sample_bias;
    int V[RTK_MAX_LAYERS] = {0, 0, 0, 0};
    int cin = U;
    const bool split = (layers[0].act & RTK_LAYER_SPLIT) != 0;      // the chain's images are split images (rtk_fused.h)
    for (int l = 0; l < nlayers; ++l) {
        RTK_REQUIRE(layers[l].w_packed && layers[l].bias && layers[l].cin16 == cin && layers[l].cout16 > 0,
                    "pointwise_mlp: layer %d expects cin16=%d, got %d (cout16=%d)", l, cin, layers[l].cin16, layers[l].cout16);
        RTK_REQUIRE(((layers[l].act & RTK_LAYER_SPLIT) != 0) == split, "pointwise_mlp: split and fp32 images in one chain (layer %d)", l);
        const size_t prev_floats = l == 0 ? 0 : (split ? (size_t)split16_nf(layers[l - 1].cin16, layers[l - 1].cout16) * 256
                                                        : (size_t)layers[l - 1].cin16 * layers[l - 1].cout16 * 256);
        RTK_REQUIRE(l == 0 || layers[l].w_packed == layers[l - 1].w_packed + prev_floats,
                    "pointwise_mlp: the packed weights of a chain must be contiguous (layer %d)", l);
        P.layer[l] = layers[l];
        P.layer[l].act = layers[l].act & 0xff;
        V[l] = layers[l].cout16;
        cin = V[l];
    }
    P.out = out;
    P.out_pitch = out_pitch;
    P.out_channels = out_channels;
    P.out_cm = out_channel_major;
    P.row_nuniq = row_nuniq;
    P.colmax = colmax;
    RTK_REQUIRE(out_channels > 0 && out_channels <= 16 * cin, "pointwise_mlp: out_channels=%d", out_channels);
    RTK_REQUIRE(out_channel_major || (out_pitch % 4 == 0 && out_pitch >= out_channels), "pointwise_mlp: bad out_pitch %d", out_pitch);
    hipStream_t s = (hipStream_t)stream;
    const bool it = interp != nullptr;
    const long key = ((((long)U * 32 + V[0]) * 32 + V[1]) * 32 + V[2]) * 32 + V[3];
#define PW_CASE(u, v1, v2, v3, v4)                                               \
    case ((((long)(u) * 32 + (v1)) * 32 + (v2)) * 32 + (v3)) * 32 + (v4):        \
        launch_pw<u, v1, v2, v3, v4>(P, it, split, s);                           \
        break;
    switch (key) {
        PW_CASE(1, 2, 0, 0, 0)     // raw (RCS, v_r) -> sa1 layer-1 projections (2 scales x 16)
        PW_CASE(4, 6, 0, 0, 0)     // sa1 out 64 -> linear1 (32) || sa2 projections (32+32)
        PW_CASE(6, 12, 0, 0, 0)    // sa2 out 96 -> linear2 (64) || sa3 projections (64+64)
        PW_CASE(8, 4, 0, 0, 0)     // sa3 out 128 -> linear3 (64)
        PW_CASE(8, 8, 0, 0, 0)     // fp3 (interp 64 || skip 64), fp1 (interp 128) -> 128
        PW_CASE(10, 8, 0, 0, 0)    // fp2 (interp 128 || skip 32) -> 128
        PW_CASE(8, 16, 0, 0, 0)    // local features 128 -> cost-volume layer-1 projection 256
        PW_CASE(16, 8, 4, 2, 1)    // cls head 256 -> 128 -> 64 -> 32 -> 1
        PW_CASE(8, 8, 4, 2, 1)     // flow head (prop 128 + per-sample GRU term) -> 128 -> 64 -> 32 -> 3
        PW_CASE(25, 2, 0, 0, 0)    // decoder embeddings (2 || 128 || 256 + per-sample) -> mse.sa1 projections
        PW_CASE(8, 2, 0, 0, 0)
        default:
            rtk_set_error("pointwise_mlp: no kernel instance for shape U=%d V=(%d,%d,%d,%d)", U, V[0], V[1], V[2], V[3]);
            return RTK_ERR_UNSUPPORTED;
    }
#undef PW_CASE
    RTK_CHECK_LAUNCH("pointwise_mlp");
    return RTK_OK;
}
