// fused_split.hip -- kernels on the split-bf16 matrix path (split_mfma.h): fp32 results from v_mfma_f32_32x32x16_bf16.
#include "rtk_common.h"
#include "rtk_fused.h"
#include "split_mfma.h"

namespace {

#ifndef SP_NW
#define SP_NW 4           // waves per workgroup = one per SIMD: the tile keeps ~400 registers per lane
#endif
#ifndef SP_F
#define SP_F 48           // fragments (KiB) per half of the LDS double buffer: a multiple of 6 (one group step)
#endif

// ---- packing: (256 x 256) row-major fp32 weights -> split image (split_mfma.h) ---------------------------------------------
__global__ __launch_bounds__(256) void pack_split_kernel(int cout, int cin, const float *__restrict__ w, u4v *__restrict__ out) {
    const int VB = cout / 32, slot = blockIdx.x * 256 + threadIdx.x;      // slot = (s, v, lane)
    if (slot >= (cin / 16) * VB * 64) return;
    const int lane = slot & 63, v = (slot >> 6) % VB, s = (slot >> 6) / VB, hh = lane >> 5, i = lane & 31;
    const float *row = w + (size_t)(32 * v + i) * cin + 32 * (s >> 1) + 16 * (s & 1) + 4 * hh;
    const f4 x0 = *reinterpret_cast<const f4 *>(row), x1 = *reinterpret_cast<const f4 *>(row + 8);
    u4v b[3];
    split3(x0, x1, b);
#pragma unroll
    for (int p = 0; p < 3; ++p) out[((size_t)(s * VB + v) * 3 + p) * 64 + lane] = b[p];
}

// ---- two 256 x 256 layers with LeakyReLU(0.1) over a list of positions (the inner layers of the cost volume, standalone) ----
__global__ __launch_bounds__(64 * SP_NW) __attribute__((amdgpu_waves_per_eu(1, 1)))
void split_mlp2_kernel(int positions, const float *__restrict__ x, const f4 *__restrict__ blob, const float *__restrict__ bias1,
                       const float *__restrict__ bias2, float *__restrict__ y) {
    __shared__ __attribute__((aligned(16))) f4 s_w[2 * SP_F * 64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), hh = lane >> 5, col = lane & 31;
    constexpr int NF = 2 * SPLIT_NF;
    WStreamA<SP_NW, SP_F, NF> ws;
    ws.start_parts(blob, s_w, wave, lane);
    const int tiles = (positions + 32 * SP_NW - 1) / (32 * SP_NW);
    for (int T = blockIdx.x; T < tiles; T += gridDim.x) {
        asm volatile("" ::: "memory");
        const long pos = (long)T * 32 * SP_NW + 32 * wave + col;
        const bool valid = pos < positions;
        const float *xr = x + (valid ? pos : (long)positions - 1) * 256 + 4 * hh;
        f4 h[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) h[i] = *reinterpret_cast<const f4 *>(xr + 8 * i);      // channels 32 (i/4) + 8 (i%4) + 4 hh ..
        f16v acc[SPLIT_VB];
#ifdef SP_REP      // experiment: the two layers SP_REP times per tile (results are wrong), to time the matrix core without the tile's I/O
        for (int rep = 1; rep < SP_REP; ++rep) {
#pragma unroll
            for (int v = 0; v < SPLIT_VB; ++v) acc[v] = split_bias(bias1, v, hh);
            split_layer<0>(ws, h, acc);
#pragma unroll
            for (int v = 0; v < SPLIT_VB; ++v)
#pragma unroll
                for (int q = 0; q < 4; ++q) h[4 * v + q] = (f4){acc[v][4 * q], acc[v][4 * q + 1], acc[v][4 * q + 2], acc[v][4 * q + 3]};
#pragma unroll
            for (int v = 0; v < SPLIT_VB; ++v) acc[v] = split_bias(bias2, v, hh);
            split_layer<SPLIT_NF>(ws, h, acc);
            ws.sync();
#pragma unroll
            for (int v = 0; v < SPLIT_VB; ++v)
#pragma unroll
                for (int q = 0; q < 4; ++q) h[4 * v + q] = (f4){acc[v][4 * q], acc[v][4 * q + 1], acc[v][4 * q + 2], acc[v][4 * q + 3]};
        }
#endif
#pragma unroll
        for (int v = 0; v < SPLIT_VB; ++v) acc[v] = split_bias(bias1, v, hh);
        split_layer<0>(ws, h, acc);
#pragma unroll
        for (int v = 0; v < SPLIT_VB; ++v)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f4 t = (f4){acc[v][4 * q], acc[v][4 * q + 1], acc[v][4 * q + 2], acc[v][4 * q + 3]};
                h[4 * v + q] = (f4){fmaxf(t.x, 0.1f * t.x), fmaxf(t.y, 0.1f * t.y), fmaxf(t.z, 0.1f * t.z), fmaxf(t.w, 0.1f * t.w)};
            }
#pragma unroll
        for (int v = 0; v < SPLIT_VB; ++v) acc[v] = split_bias(bias2, v, hh);
        split_layer<SPLIT_NF>(ws, h, acc);
        ws.sync();
        float *yr = y + pos * 256 + 4 * hh;
        if (valid) {
#pragma unroll
            for (int v = 0; v < SPLIT_VB; ++v)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f4 t = (f4){acc[v][4 * q], acc[v][4 * q + 1], acc[v][4 * q + 2], acc[v][4 * q + 3]};
                    *reinterpret_cast<f4 *>(yr + 32 * v + 8 * q) = (f4){fmaxf(t.x, 0.1f * t.x), fmaxf(t.y, 0.1f * t.y),
                                                                        fmaxf(t.z, 0.1f * t.z), fmaxf(t.w, 0.1f * t.w)};
                }
        }
    }
    ws.finish();
}

// ---- rtk_cost_volume on the split path ------------------------------------------------------------------------------------
// Same operator as cost_volume_kernel (fused_group.hip); a wave owns TWO query points x their 16 neighbours (the 32 columns of
// the 32x32 tile), a workgroup (one wave per SIMD) eight points per iteration.  Layer 1 (K = 3) and the WeightNet's last
// layer (K = 8) stay on the fp32-input MFMA (v_mfma_f32_32x32x2_f32, same C/D layout); the two 256 x 256 layers -- 99 % of the
// flops -- run split.
struct CvSplitParams {
    int samples, n1, n2, gx;
    const float *xyz1, *xyz2;
    const int64_t *knn;
    const float *p1, *p2;
    const float *wd;              // [16][64] image of [Wd | 0] (fused_common.h): Wd[c][k] = wd[(c / 16) * 64 + 16 k + c % 16]
    const f4 *blob;               // split images of layers 2, 3
    const float *bias2, *bias3;
    const float *wa, *wb, *wc;    // WeightNet images as packed for the 16x16 kernels: [Wa | ba], Wb (one fragment), Wc (16 fragments)
    const float *bb, *bc;
    float *out;
    int out_pitch;
};

__device__ __forceinline__ f16v mfma_f32x2(float a, float b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

__device__ __forceinline__ f4 leaky4(f4 t) { return (f4){fmaxf(t.x, 0.1f * t.x), fmaxf(t.y, 0.1f * t.y), fmaxf(t.z, 0.1f * t.z), fmaxf(t.w, 0.1f * t.w)}; }

__global__ __launch_bounds__(64 * SP_NW) __attribute__((amdgpu_waves_per_eu(1, 1))) void cost_volume_split_kernel(const CvSplitParams P) {
    __shared__ __attribute__((aligned(16))) f4 s_w[2 * SP_F * 64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), hh = lane >> 5, col = lane & 31, pp = col >> 4,
              j = col & 15;      // wave index in an SGPR: the DMA's LDS destination (M0) is then scalar arithmetic
    int b, bx, nbx;
    rtk_decode_block(P.gx, b, bx, nbx);
    constexpr int PPW = 2 * SP_NW;                                  // points per workgroup iteration
    const int groups = (P.n1 + PPW - 1) / PPW;
    WStreamA<SP_NW, SP_F, 2 * SPLIT_NF> ws;
    ws.start_parts(P.blob, s_w, wave, lane);
    for (int G = bx; G < groups; G += nbx) {
        asm volatile("" ::: "memory");
        const int pt = G * PPW + 2 * wave + pp;
        const bool valid = pt < P.n1;
        const long i = (long)b * P.n1 + (valid ? pt : P.n1 - 1);
        const long nb = (long)b * P.n2 + (long)P.knn[i * 16 + j];
        const float dx = __fsub_rn(P.xyz2[nb * 3], P.xyz1[i * 3]), dy = __fsub_rn(P.xyz2[nb * 3 + 1], P.xyz1[i * 3 + 1]),
                    dz = __fsub_rn(P.xyz2[nb * 3 + 2], P.xyz1[i * 3 + 2]);
        // layer 1: leaky(p1[i] + p2[nb] + Wd.d)     (bias folded into p1)
        f4 h[32];
        {
            const float *r1 = P.p1 + i * 256 + 4 * hh, *r2 = P.p2 + nb * 256 + 4 * hh;
            const float b0 = hh ? dy : dx, b1 = hh ? 0.f : dz;      // B[k = hh][col] of the two k-steps (k = 3: the zero column)
#pragma unroll
            for (int v = 0; v < SPLIT_VB; ++v) {
                f16v c;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f4 t = *reinterpret_cast<const f4 *>(r1 + 32 * v + 8 * q) + *reinterpret_cast<const f4 *>(r2 + 32 * v + 8 * q);
                    c[4 * q] = t.x; c[4 * q + 1] = t.y; c[4 * q + 2] = t.z; c[4 * q + 3] = t.w;
                }
                const int ch = 32 * v + col;                         // A[i = col][k = hh]
                const float *wr = P.wd + (ch >> 4) * 64 + (ch & 15);
                c = mfma_f32x2(wr[16 * hh], b0, c);
                c = mfma_f32x2(wr[16 * (2 + hh)], b1, c);
#pragma unroll
                for (int q = 0; q < 4; ++q) h[4 * v + q] = leaky4((f4){c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]});
            }
        }
        f16v acc[SPLIT_VB];
#pragma unroll
        for (int v = 0; v < SPLIT_VB; ++v) acc[v] = split_bias(P.bias2, v, hh);
        split_layer<0>(ws, h, acc);
#pragma unroll
        for (int v = 0; v < SPLIT_VB; ++v)
#pragma unroll
            for (int q = 0; q < 4; ++q) h[4 * v + q] = leaky4((f4){acc[v][4 * q], acc[v][4 * q + 1], acc[v][4 * q + 2], acc[v][4 * q + 3]});
#pragma unroll
        for (int v = 0; v < SPLIT_VB; ++v) acc[v] = split_bias(P.bias3, v, hh);
        split_layer<SPLIT_NF>(ws, h, acc);
        ws.sync();                                                   // wrap the stream to chunk 0
        // WeightNet hidden layers (3 -> 8 -> 8) of this lane's position: uniform weights, every lane its own direction
        float t2[8];
        {
            float t1[8];
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                float a = __fmaf_rn(P.wa[o], dx, 0.f);
                a = __fmaf_rn(P.wa[16 + o], dy, a);
                a = __fmaf_rn(P.wa[32 + o], dz, a);
                t1[o] = fmaxf(__fadd_rn(a, P.wa[48 + o]), 0.f);
            }
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                float a = P.bb[o];
#pragma unroll
                for (int c = 0; c < 8; ++c) a = __fmaf_rn(P.wb[(16 * (c >> 2) + o) * 4 + (c & 3)], t1[c], a);
                t2[o] = fmaxf(a, 0.f);
            }
        }
        // out[i] = sum over the 16 neighbours of relu(Wc.t2 + bc) * a3, one 32-channel block at a time
        float *o = P.out + i * P.out_pitch + 4 * hh;
#pragma unroll
        for (int v = 0; v < SPLIT_VB; ++v) {
            f16v w = split_bias(P.bc, v, hh);
            const int ch = 32 * v + col;
            const float *wr = P.wc + ((ch >> 4) * 64 + (ch & 15)) * 4;      // Wc[ch][k] = wr[64 (k / 4) + k % 4]
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int k = 2 * st + hh;
                w = mfma_f32x2(wr[64 * (k >> 2) + (k & 3)], hh ? t2[2 * st + 1] : t2[2 * st], w);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f4 r;
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = fmaxf(w[4 * q + e], 0.f) * fmaxf(acc[v][4 * q + e], 0.1f * acc[v][4 * q + e]);
                row_sum16_f4(r);
                if (valid && j == 0) *reinterpret_cast<f4 *>(o + 32 * v + 8 * q) = r;
            }
        }
    }
    ws.finish();
}

}  // namespace

extern "C" int rtk_pack_split_layer(int cout, int cin, const float *w, void *image, rtk_stream_t stream) {
    RTK_REQUIRE(cout > 0 && cin > 0 && cout % 32 == 0 && cin % 32 == 0 && w && image, "pack_split_layer: bad arguments");
    const int slots = (cin / 16) * (cout / 32) * 64;
    pack_split_kernel<<<(slots + 255) / 256, 256, 0, (hipStream_t)stream>>>(cout, cin, w, (u4v *)image);
    RTK_CHECK_LAUNCH("pack_split_layer");
    return RTK_OK;
}

extern "C" int rtk_split_mlp2(int positions, const float *x, const void *images, const float *bias1, const float *bias2, float *y,
                              rtk_stream_t stream) {
    RTK_REQUIRE(positions > 0 && x && images && bias1 && bias2 && y, "split_mlp2: bad arguments");
    const int tiles = (positions + 32 * SP_NW - 1) / (32 * SP_NW);
    split_mlp2_kernel<<<tiles < 256 ? tiles : 256, 64 * SP_NW, 0, (hipStream_t)stream>>>(positions, x, (const f4 *)images, bias1, bias2, y);
    RTK_CHECK_LAUNCH("split_mlp2");
    return RTK_OK;
}

extern "C" int rtk_cost_volume_split(int samples, int n1, int n2, const float *xyz1, const float *xyz2, const int64_t *knn_idx,
                                     const float *p1, const float *p2, const float *wd_packed, const void *split_images,
                                     const float *bias2, const float *bias3, const rtk_layer_t *wn, float *out, int out_pitch,
                                     rtk_stream_t stream) {
    RTK_REQUIRE(samples > 0 && samples <= 65535 && n1 > 0 && n2 >= 16 && xyz1 && xyz2 && knn_idx && p1 && p2 && wd_packed && split_images &&
                bias2 && bias3 && out, "cost_volume_split: bad arguments");
    RTK_REQUIRE(out_pitch % 4 == 0 && out_pitch >= 256, "cost_volume_split: bad out_pitch");
    RTK_REQUIRE(wn && wn[0].w_packed && wn[1].w_packed && wn[2].w_packed && wn[1].bias && wn[2].bias && wn[1].cin16 == 1 &&
                wn[1].cout16 == 1 && wn[2].cin16 == 1 && wn[2].cout16 == 16, "cost_volume_split: bad WeightNet layers");
    CvSplitParams P;
    P.samples = samples; P.n1 = n1; P.n2 = n2;
    P.xyz1 = xyz1; P.xyz2 = xyz2; P.knn = knn_idx; P.p1 = p1; P.p2 = p2; P.wd = wd_packed;
    P.blob = reinterpret_cast<const f4 *>(split_images); P.bias2 = bias2; P.bias3 = bias3;
    P.wa = wn[0].w_packed; P.wb = wn[1].w_packed; P.wc = wn[2].w_packed; P.bb = wn[1].bias; P.bc = wn[2].bias;
    P.out = out; P.out_pitch = out_pitch;
    const int groups = (n1 + 2 * SP_NW - 1) / (2 * SP_NW);
    int gx = 256 / samples;                           // one workgroup per CU (the tile keeps the whole register file); the rest is looped
    if (gx < 1) gx = 1;
    if (gx > groups) gx = groups;
    P.gx = samples % 8 == 0 ? gx : 0;
    const dim3 grid = P.gx ? dim3(gx * samples) : dim3(gx, samples);
    cost_volume_split_kernel<<<grid, 64 * SP_NW, 0, (hipStream_t)stream>>>(P);
    RTK_CHECK_LAUNCH("cost_volume_split");
    return RTK_OK;
}
