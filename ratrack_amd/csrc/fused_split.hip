// fused_split.hip -- kernels on the split matrix path (split_mfma.h): fp32 results from v_mfma_f32_32x32x16_f16, two fp16 pieces
// per operand, three products, a power-of-two scale per weight matrix and per position.
#include "rtk_common.h"
#include "rtk_fused.h"
#include "split_mfma.h"

namespace {

#ifndef SP_NW
#define SP_NW 4           // waves per workgroup = one per SIMD: the tile keeps ~400 registers per lane
#endif
#ifndef SP_F
#define SP_F 32           // fragments (KiB) per half of the LDS double buffer: a power of two (a layer is 256 fragments)
#endif
// The forward cost volume streams in halves of 32 KiB (eight group steps of six MFMAs per chunk: the 1 536 matrix clocks the
// 24 KiB halves of the six-product kernel lasted), next to the 64 KiB of staged rows of layer 1.
constexpr int CV_F = 32;

// ---- packing: (cout x cin) row-major fp32 weights -> split image + the inverse of its power-of-two scale (split_mfma.h) ---------
// Every workgroup scans the whole matrix for max|w| itself (at most 256 KiB, L2-resident: one launch, no atomics, no scratch).
__global__ __launch_bounds__(256) void pack_split_kernel(int cout, int cin, const float *__restrict__ w, int transposed, u4v *__restrict__ out,
                                                         float *__restrict__ inv_scale) {
    __shared__ unsigned s_m[4];
    float m = 0.f;
    const f4 *w4 = reinterpret_cast<const f4 *>(w);
    for (int i = threadIdx.x; i < cout * cin / 4; i += 256) {
        const f4 t = w4[i];
        m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(t.x)), __builtin_fabsf(t.y));
        m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(t.z)), __builtin_fabsf(t.w));
    }
    unsigned mb = __float_as_uint(m);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const unsigned o = __shfl_xor(mb, d, 64); mb = o > mb ? o : mb; }
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = mb;
    __syncthreads();
    mb = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3]));
    const LaneScale sc = lane_scale_of(mb);
    if (blockIdx.x == 0 && threadIdx.x == 0) *inv_scale = sc.inv;
    const int VB = cout / 32, slot = blockIdx.x * 256 + threadIdx.x;      // slot = (s, v, lane)
    if (slot >= (cin / 16) * VB * 64) return;
    const int lane = slot & 63, v = (slot >> 6) % VB, s = (slot >> 6) / VB, hh = lane >> 5, i = lane & 31;
    const int o = 32 * v + i, c0 = 32 * (s >> 1) + 16 * (s & 1) + 4 * hh;      // this slot: W[o][c0 .. c0 + 3], W[o][c0 + 8 .. c0 + 11]
    f4 x0, x1;
    if (transposed) {                                                          // W = w^T, w (cin, cout) row-major
#pragma unroll
        for (int r = 0; r < 4; ++r) { x0[r] = w[(size_t)(c0 + r) * cout + o]; x1[r] = w[(size_t)(c0 + 8 + r) * cout + o]; }
    } else {
        const float *row = w + (size_t)o * cin + c0;
        x0 = *reinterpret_cast<const f4 *>(row); x1 = *reinterpret_cast<const f4 *>(row + 8);
    }
    u4v b[2];
    split2(x0, x1, sc.s, b);
#pragma unroll
    for (int p = 0; p < 2; ++p) out[((size_t)(s * VB + v) * 2 + p) * 64 + lane] = b[p];
}

// y = leaky(acc c + b) for a whole accumulator set (c = this lane's 2^-(kw + kx)): the epilogue between two split layers
__device__ __forceinline__ f4 leaky_med4(f4 t, float inf) {
    const f4 u = t * 0.1f;                                                  // two v_pk_mul_f32
    return (f4){__builtin_amdgcn_fmed3f(t.x, u.x, inf), __builtin_amdgcn_fmed3f(t.y, u.y, inf), __builtin_amdgcn_fmed3f(t.z, u.z, inf),
                __builtin_amdgcn_fmed3f(t.w, u.w, inf)};
}
__device__ __forceinline__ f4 scale_bias4(f4 a, float c, f4 b) { return __builtin_elementwise_fma(a, (f4){c, c, c, c}, b); }
__device__ __forceinline__ float rtk_hidden_inf();

// ---- two 256 x 256 layers with LeakyReLU(0.1) over a list of positions (the inner layers of the cost volume, standalone) ----
__global__ __launch_bounds__(64 * SP_NW) __attribute__((amdgpu_waves_per_eu(1, 1)))
void split_mlp2_kernel(int positions, const float *__restrict__ x, const f4 *__restrict__ blob, const float *__restrict__ wsc,
                       const float *__restrict__ bias1, const float *__restrict__ bias2, float *__restrict__ y) {
    const float kinf = rtk_hidden_inf();
    __shared__ __attribute__((aligned(16))) f4 s_w[2 * SP_F * 64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), hh = lane >> 5, col = lane & 31;
    constexpr int NF = 2 * SPLIT_NF;
    WStreamA<SP_NW, SP_F, NF> ws;
    ws.start_parts(blob, s_w, wave, lane);
    const float wi1 = ldc(wsc), wi2 = ldc(wsc + 1);
    const int tiles = (positions + 32 * SP_NW - 1) / (32 * SP_NW);
    for (int T = blockIdx.x; T < tiles; T += gridDim.x) {
        asm volatile("" ::: "memory");
        const long pos = (long)T * 32 * SP_NW + 32 * wave + col;
        const bool valid = pos < positions;
        const float *xr = x + (valid ? pos : (long)positions - 1) * 256 + 4 * hh;
        f4 h[32], bq[2][8];
#pragma unroll
        for (int i = 0; i < 32; ++i) h[i] = *reinterpret_cast<const f4 *>(xr + 8 * i);      // channels 32 (i/4) + 8 (i%4) + 4 hh ..
        f16v acc[SPLIT_VB];
        LaneScale sc = lane_scale32(h);
        split_layer<0>(ws, h, sc.s, acc, BiasSide{bias1 + 4 * hh, bq});
        float c = sc.inv * wi1;
        split_epilogue(acc, bias1 + 4 * hh, bq, [&](int e, f4 a, f4 b) { h[e] = leaky_med4(scale_bias4(a, c, b), kinf); });
        sc = lane_scale32(h);
        split_layer<SPLIT_NF>(ws, h, sc.s, acc, BiasSide{bias2 + 4 * hh, bq});
        ws.sync();
        c = sc.inv * wi2;
        float *yr = y + pos * 256 + 4 * hh;
        split_epilogue(acc, bias2 + 4 * hh, bq, [&](int e, f4 a, f4 b) {
            if (valid) *reinterpret_cast<f4 *>(yr + 8 * e) = leaky_med4(scale_bias4(a, c, b), kinf);
        });
    }
    ws.finish();
}

// ---- rtk_cost_volume on the split path ------------------------------------------------------------------------------------
// Same operator as cost_volume_kernel (fused_group.hip); a wave owns TWO query points x their 16 neighbours (the 32 columns of
// the 32x32 tile), a workgroup (one wave per SIMD) eight points per iteration.  Layer 1 (K = 3) and the WeightNet's last
// layer (K = 8) stay on the fp32-input MFMA (v_mfma_f32_32x32x2_f32, same C/D layout); the two 256 x 256 layers -- 99 % of the
// flops -- run split.
__device__ __forceinline__ f16v mfma_f32x2(float a, float b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

struct WnSplit {                  // WeightNet images as packed for the 16x16 kernels (fused_group.hip)
    const float *wa;              // [Wa | ba]: Wa[o][k] = wa[16 k + o], ba[o] = wa[48 + o]
    const float *wb, *wc;         // Wb[o][c] = wb[(16 (c / 4) + o) 4 + c % 4];  Wc[ch][k] = wc[((ch / 16) 64 + 16 (k / 4) + ch % 16) 4 + k % 4]
    const float *bb, *bc;
};

struct CvSplitParams {
    int samples, n1, n2, gx;
    const float *xyz1, *xyz2;
    const int64_t *knn;
    const float *p1, *p2;
    const float *wd;              // [16][64] image of [Wd | 0] (fused_common.h): Wd[c][k] = wd[(c / 16) * 64 + 16 k + c % 16]
    const f4 *blob;               // split images of layers 2, 3 ...
    const float *wsc;             // ... and the inverses of their power-of-two weight scales (rtk_pack_split_layer)
    const float *bias2, *bias3;
    WnSplit wn;
    float *out;
    int out_pitch;
    float *amax;                  // training kernels: [2] zero-initialised, the largest |element| of (a1, a2) -- forward -- / (dz3, dz2) -- backward
    float *sv1, *sv2, *sv3;       // training forward (SAVE): the three activations (positions, 256) ...
    uint2 *mk1, *mk2;             // ... and the sign masks of a1, a2 in the format of cost_volume_kernel<true> (fused_group.hip): word
                                  // (position, g), bit 4 v16 + r = [a[channel 16 v16 + 4 g + r] > 0]
};

// This lane's channels 32 v + 8 q + 4 hh + r are the 16-wide kernels' (v16 = 2 v + q / 2, g = 2 (q % 2) + hh, r): it owns the two
// complete mask words g = hh (q even) and g = 2 + hh (q odd) of its position.
__device__ __forceinline__ void split_sign_masks(const f4 (&a)[32], uint2 (&m)[2]) {
    m[0] = m[1] = make_uint2(0u, 0u);
#pragma unroll
    for (int v = 0; v < SPLIT_VB; ++v)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned bit = (a[4 * v + q][r] > 0.f ? 1u : 0u) << (4 * ((2 * v + (q >> 1)) & 7) + r);
                if (v < 4) m[q & 1].x |= bit; else m[q & 1].y |= bit;
            }
}
__device__ __forceinline__ unsigned split_mask_bits(const uint2 (&m)[2], int v, int q) {      // [z > 0] of (v, q, r = 0..3) in the low four bits
    return ((v < 4 ? m[q & 1].x : m[q & 1].y) >> (4 * ((2 * v + (q >> 1)) & 7))) & 15u;
}

__device__ __forceinline__ f4 *cv_at(float *base, unsigned byte_off) { return reinterpret_cast<f4 *>(reinterpret_cast<char *>(base) + byte_off); }
__device__ __forceinline__ const f4 *cv_at(const float *base, unsigned byte_off) {
    return reinterpret_cast<const f4 *>(reinterpret_cast<const char *>(base) + byte_off);
}

// WeightNet hidden layers (3 -> 8 -> 8, ReLU) of this lane's position: uniform weights, every lane its own direction
__device__ __forceinline__ void wn_hidden(const WnSplit &W, float dx, float dy, float dz, float (&t2)[8]) {
    float t1[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        float a = __fmaf_rn(ldc(W.wa + o), dx, 0.f);
        a = __fmaf_rn(ldc(W.wa + 16 + o), dy, a);
        a = __fmaf_rn(ldc(W.wa + 32 + o), dz, a);
        t1[o] = fmaxf(__fadd_rn(a, ldc(W.wa + 48 + o)), 0.f);
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        float a = ldc(W.bb + o);
#pragma unroll
        for (int c = 0; c < 8; ++c) a = __fmaf_rn(ldc(W.wb + (16 * (c >> 2) + o) * 4 + (c & 3)), t1[c], a);
        t2[o] = fmaxf(a, 0.f);
    }
}
// relu(Wc.t2 + bc) for the 32-channel block v in the tile layout (K = 8 on the fp32-input MFMA: four k-steps of two), in two
// halves: this lane's operands of the block (bias, four weights: 20 registers), and the product.  The epilogues request block
// v + 1's operands before they work on block v -- the conditional stores of a block end a basic block each, and hipcc's scheduler
// moves no load across them, so every block started with an exposed round trip.
struct WnBlock {
    f16v bias;
    float w[4];
};
__device__ __forceinline__ WnBlock wn_block(const WnSplit &W, int v, int hh, int col) {
    WnBlock k;
    k.bias = split_bias(W.bc, v, hh);
    const int ch = 32 * v + col;                                         // A[i = col][k = hh]
    const float *wr = W.wc + ((ch >> 4) * 64 + (ch & 15)) * 4;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        const int kk = 2 * st + hh;
        k.w[st] = ldc(wr + 64 * (kk >> 2) + (kk & 3));
    }
    return k;
}
__device__ __forceinline__ f16v wn_pre(const WnBlock &k, int hh, const float (&t2)[8]) {      // Wc.t2 + bc, before the ReLU
    f16v w = k.bias;
#pragma unroll
    for (int st = 0; st < 4; ++st) w = mfma_f32x2(k.w[st], hh ? t2[2 * st + 1] : t2[2 * st], w);
    return w;
}
// max(x, 0.1 x) and max(x, 0) in ONE instruction each.  fmaxf() costs two under IEEE mode -- hipcc first quiets a possible signalling
// NaN in every operand it did not compute itself (v_max_f32 x, x, x on each accumulator read): 32 extra VALU instructions per
// 32-channel block of the epilogue, 256 per layer boundary -- and a median with a literal +inf (v_med3_f32) is folded back into
// exactly that maxnum.  Both are the median with a +inf the optimiser cannot see (an SGPR written by a volatile asm, once per
// kernel: rtk_hidden_inf): ReLU = med3(x, 0, inf), LeakyReLU = med3(x, 0.1 x, inf) (leaky_med above) -- instructions the compiler
// knows, so it places the wait states a matrix-core result needs itself (round 4's LeakyReLU was a written-out v_max_f32 that
// relied on the product in front of it for that).  Same bits as fmaxf for every non-NaN input.
__device__ __forceinline__ float rtk_hidden_inf() {
    float v;
    asm volatile("s_mov_b32 %0, 0x7f800000" : "=s"(v));
    return v;
}
__device__ __forceinline__ float relu1(float x, float inf) { return __builtin_amdgcn_fmed3f(x, 0.f, inf); }
__device__ __forceinline__ f4 relu_med4(f4 t, float inf) { return (f4){relu1(t.x, inf), relu1(t.y, inf), relu1(t.z, inf), relu1(t.w, inf)}; }

// ---- layer 1's operands ---------------------------------------------------------------------------------------------------
// The tile layout gives a lane 32 bytes of a gathered p2 row per load instruction (its own position's row, two lanes per position):
// 32 partial cache lines per instruction, each fetched whole from L2 and -- four instructions in a row touching the same line
// while it is still in flight -- fetched again (tools/experiments/cv_ticks.py: 12 k of a tile's 86 k clocks went into layer 1, 6 k
// of them gone when every lane reads the same row).  So the rows come through LDS instead: one global_load_lds per two positions
// moves 2 x 512 contiguous bytes (channels 128 HALF .. 128 HALF + 127 of both rows) into the wave's own 16 KiB of LDS, every line
// fetched once, and the lanes read their slots from there.  Two rounds per tile (HALF = 0, 1: 64 KiB per workgroup next to the
// weight stream's 48, so that a CU keeps 48 KiB for other kernels' workgroups); round 0 of the NEXT tile is requested right after layer 1 and lands under layers 2 and 3.
// Slot p of a position's 512 bytes holds source chunk p ^ (position & 15): the 16 lanes that read together (one hh, 16 positions)
// then hit 16 different bank groups.
constexpr int CV_ROWS_F4 = 32 * 32;      // f4 per wave: 32 positions x 32 slots of 16 bytes
struct CvRowsRequest {
    const char *p2;                // the p2 rows (wave-uniform; byte offsets are 32-bit: at most 2^22 rows)
    f4 *rows;                      // this wave's 16 KiB
    int nbrow;                     // this lane's gathered row (lane col and lane 32 + col hold the same)
    unsigned lanepart;             // 512 HALF + ((slot ^ sub) << 4): (slot ^ ((2 t + sub) & 15)) << 4 == lanepart ^ ((2 t & 15) << 4) below 512
    int sub;
    __device__ __forceinline__ CvRowsRequest(const float *p2_, f4 *rows_, int nbrow_, int half, int lane)
        : p2(reinterpret_cast<const char *>(p2_)), rows(rows_), nbrow(nbrow_), lanepart(512u * half + (unsigned)(((lane & 31) ^ (lane >> 5)) << 4)), sub(lane >> 5) {}
    // lanes 0..31: position 2 T, lanes 32..63: position 2 T + 1 (their row indices live in lanes 2 T, 2 T + 1)
    template <int T>
    __device__ __forceinline__ void one() const {
        const int r0 = __builtin_amdgcn_readlane(nbrow, 2 * T), r1 = __builtin_amdgcn_readlane(nbrow, 2 * T + 1);
        const unsigned off = ((unsigned)(sub ? r1 : r0) << 10) + (lanepart ^ (unsigned)(((2 * T) & 15) << 4));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p2 + off),
                                         (__attribute__((address_space(3))) void *)(rows + T * 64), 16, 0, 0);
    }
    template <int... T>
    __device__ __forceinline__ void all(std::integer_sequence<int, T...>) const { (one<T>(), ...); }
};
template <int HALF>
__device__ __forceinline__ void cv_rows_request(const float *p2, int nbrow, f4 *rows, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the previous round's reads have returned before their slots are overwritten
    CvRowsRequest(p2, rows, nbrow, HALF, lane).all(std::make_integer_sequence<int, 16>{});
}
// Side job of layer 2 (split_mfma.h): the NEXT tile's round 0, three requests in each of the first group steps of the layer's first two
// chunks -- where the weight stream issues its own, so that they are as old as those at the chunk's closing vmcnt(0) -- and, in the
// training forward, the a1 stores.
template <bool SAVE>
struct CvLayer2Side {
    StoreRowsSide st;
    CvRowsRequest rq;
    template <int GI>
    __device__ __forceinline__ void at(const f4 (&h)[32]) const {
        if constexpr (SAVE) st.template at<GI>(h);
        constexpr int per_chunk = CV_F / SPLIT_GF, ig = split_issue_groups(CV_F), g = GI % per_chunk, n = (GI / per_chunk) * ig + g;
        if constexpr (g < ig && 3 * n < 16) {
            rq.template one<3 * n>();
            if constexpr (3 * n + 1 < 16) rq.template one<3 * n + 1>();
            if constexpr (3 * n + 2 < 16) rq.template one<3 * n + 2>();
        }
    }
};
// this lane's 16 slots of the round: channels 128 HALF + 8 e + 4 hh .. + 3, e = 0..15
template <int HALF>
__device__ __forceinline__ void cv_rows_read(const f4 *rows, int col, int hh, f4 (&h)[32]) {
#pragma unroll
    for (int e = 0; e < 16; ++e) h[16 * HALF + e] = rows[col * 32 + ((2 * e + hh) ^ (col & 15))];
}
// r + q as held by lane K of this lane's row of 16 (the p1 row of a point is the same for its 16 neighbours: each of them loads
// two of its 32 slots and the additions pick the owner's copy -- 2 loads per lane instead of 32 returning the same bytes 16 times)
template <int K>
__device__ __forceinline__ f4 add_row_bcast(const f4 q, const f4 r) {
    f4 o;
    asm("s_nop 1\n"
        "v_add_f32_dpp %0, %4, %8 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %1, %5, %9 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %2, %6, %10 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %3, %7, %11 row_newbcast:%12 row_mask:0xf bank_mask:0xf"
        : "=&v"(o.x), "=&v"(o.y), "=&v"(o.z), "=&v"(o.w)
        : "v"(q.x), "v"(q.y), "v"(q.z), "v"(q.w), "v"(r.x), "v"(r.y), "v"(r.z), "v"(r.w), "n"(K));
    return o;
}
// r * q as held by lane K of this lane's row of 16 (the backward's dout row: the same for a point's 16 neighbours)
template <int K>
__device__ __forceinline__ f4 mul_row_bcast(const f4 q, const f4 r) {
    f4 o;
    asm("s_nop 1\n"
        "v_mul_f32_dpp %0, %4, %8 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n"
        "v_mul_f32_dpp %1, %5, %9 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n"
        "v_mul_f32_dpp %2, %6, %10 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n"
        "v_mul_f32_dpp %3, %7, %11 row_newbcast:%12 row_mask:0xf bank_mask:0xf"
        : "=&v"(o.x), "=&v"(o.y), "=&v"(o.z), "=&v"(o.w)
        : "v"(q.x), "v"(q.y), "v"(q.z), "v"(q.w), "v"(r.x), "v"(r.y), "v"(r.z), "v"(r.w), "n"(K));
    return o;
}
template <int V0, int V1>
__device__ __forceinline__ void cv_layer1_blocks(const CvSplitParams &P, const f4 q0, const f4 q1, float b0, float b1, int hh, int col, float kinf, f4 (&h)[32]) {
    static_assert(V1 <= 8, "eight 32-channel blocks");
    auto block = [&](auto vc) {
        constexpr int v = decltype(vc)::value;
        f16v c;
        auto slot = [&](auto qc) {
            constexpr int q = decltype(qc)::value, e = 4 * v + q;
            const f4 t = add_row_bcast<e & 15>(e < 16 ? q0 : q1, h[e]);
            c[4 * q] = t.x; c[4 * q + 1] = t.y; c[4 * q + 2] = t.z; c[4 * q + 3] = t.w;
        };
        slot(std::integral_constant<int, 0>{}); slot(std::integral_constant<int, 1>{});
        slot(std::integral_constant<int, 2>{}); slot(std::integral_constant<int, 3>{});
        const int ch = 32 * v + col;                         // A[i = col][k = hh]
        const float *wr = P.wd + (ch >> 4) * 64 + (ch & 15);
        c = mfma_f32x2(ldc(wr + 16 * hh), b0, c);
        c = mfma_f32x2(ldc(wr + 16 * (2 + hh)), b1, c);
#pragma unroll
        for (int q = 0; q < 4; ++q) h[4 * v + q] = leaky_med4((f4){c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]}, kinf);
    };
    block(std::integral_constant<int, V0>{}); block(std::integral_constant<int, V0 + 1>{});
    block(std::integral_constant<int, V0 + 2>{}); block(std::integral_constant<int, V0 + 3>{});
}

// Register cap of the forward kernel.  Left alone (512) hipcc spreads the tile over 500 registers and nothing else fits on the SIMD;
// capped it allocates 404 without a spill, and the small-register geometry kernels of the other batches in flight (FPS 20, ball
// query 16, three-NN 12, kNN 40 registers) can share the SIMDs with it: +0.7 % frame-pairs/s, the kernel alone unchanged.
// (Round 6: with the hipcc of ROCm 7.2 the attribute no longer binds -- the code object reports .vgpr_count 496 (256 + 240 accumulation
// registers) for any cap from 320 to 512, with or without amdgpu_waves_per_eu: one wave owns the SIMD's register file.)
#ifndef CV_FWD_VGPRS
#define CV_FWD_VGPRS 448
#endif
template <bool SAVE>
__global__ __launch_bounds__(64 * SP_NW) __attribute__((amdgpu_waves_per_eu(1, 1), amdgpu_num_vgpr(CV_FWD_VGPRS)))
void cost_volume_split_kernel(const CvSplitParams P) {
    const float kinf = rtk_hidden_inf();
    __shared__ __attribute__((aligned(16))) f4 s_w[2 * CV_F * 64];
    __shared__ __attribute__((aligned(16))) f4 s_rows[SP_NW * CV_ROWS_F4];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), hh = lane >> 5, col = lane & 31, pp = col >> 4,
              j = col & 15;      // wave index in an SGPR: the DMA's LDS destination (M0) is then scalar arithmetic
    f4 *rows = s_rows + wave * CV_ROWS_F4;
    constexpr int PPW = 2 * SP_NW;                                  // points per workgroup iteration
    const int groups = (P.n1 + PPW - 1) / PPW;
    // Tiles = (sample, group of eight points).  P.gx > 0 (samples % 8 == 0): a 1-D grid of P.gx workgroups, ALL tiles of sample s on
    // XCD s % 8 -- workgroup L serves XCD L % 8 and strides over that XCD's tiles (sample-major) with the XCD's P.gx / 8 workgroups,
    // so that the launcher can give the kernel any share of the CUs (cv_split_fill).  Else a plain 2-D grid, one sample per row.
    const bool flat = P.gx > 0;
    const int xcd = blockIdx.x & 7, t0 = flat ? (int)(blockIdx.x >> 3) : (int)blockIdx.x, tstep = flat ? P.gx >> 3 : (int)gridDim.x;
    const int ntiles = flat ? (P.samples >> 3) * groups : groups;
    auto locate = [&](int t, int &b_, int &G_) {
        if (flat) { const int sk = t / groups; b_ = sk * 8 + xcd; G_ = t - sk * groups; }
        else { b_ = blockIdx.y; G_ = t; }
    };
    int b, bx;
    locate(t0 < ntiles ? t0 : 0, b, bx);
    WStreamA<SP_NW, CV_F, 2 * SPLIT_NF> ws;
    ws.start_parts(P.blob, s_w, wave, lane);
    const float wi2 = ldc(P.wsc), wi3 = ldc(P.wsc + 1);
    // The tile loop is software-pipelined by one tile (round 4): the NEXT tile's neighbour index is requested at the top of the
    // CURRENT tile, the first half of its gathered rows (cv_rows_request<0>) right after layer 1, its direction and its two p1 slots
    // inside the epilogue, whose WeightNet / neighbour-sum arithmetic (VALU + DPP) runs while they arrive.
    int pt = bx * PPW + 2 * wave + pp;
    bool valid = pt < P.n1;
    long i = (long)b * P.n1 + (valid ? pt : P.n1 - 1);
    long nb = 0;
    float dx = 0.f, dy = 0.f, dz = 0.f;
    f4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = q0;                          // p1 slots j and 16 + j of this lane's point
    if (t0 < ntiles) {
        nb = (long)b * P.n2 + (long)P.knn[i * 16 + j];
        cv_rows_request<0>(P.p2, (int)nb, rows, lane);
        q0 = ldc4(P.p1 + i * 256 + 4 * hh + 8 * j); q1 = ldc4(P.p1 + i * 256 + 4 * hh + 8 * (16 + j));
        dx = __fsub_rn(P.xyz2[nb * 3], P.xyz1[i * 3]); dy = __fsub_rn(P.xyz2[nb * 3 + 1], P.xyz1[i * 3 + 1]);
        dz = __fsub_rn(P.xyz2[nb * 3 + 2], P.xyz1[i * 3 + 2]);
    }
    for (int t = t0; t < ntiles; t += tstep) {
        asm volatile("" ::: "memory");
        // the next tile's neighbour index (wave-uniform condition)
        const bool more = t + tstep < ntiles;
        int bn, Gn;
        locate(more ? t + tstep : t, bn, Gn);
        const int ptn = Gn * PPW + 2 * wave + pp;
        const bool validn = ptn < P.n1;
        const long in_ = (long)bn * P.n1 + (validn ? ptn : P.n1 - 1);
        long knn_next = 0;
        if (more) knn_next = (long)P.knn[in_ * 16 + j];
        // layer 1: leaky(p1[i] + p2[nb] + Wd.d)     (bias folded into p1)
        f4 h[32];
        float t2[8];
        {
            const float b0 = hh ? dy : dx, b1 = hh ? 0.f : dz;      // B[k = hh][col] of the two k-steps (k = 3: the zero column)
            cv_rows_read<0>(rows, col, hh, h);
            cv_rows_request<1>(P.p2, (int)nb, rows, lane);
            cv_layer1_blocks<0, 4>(P, q0, q1, b0, b1, hh, col, kinf, h);
            // the WeightNet's hidden layers (3 -> 8 -> 8: ~100 VALU instructions on the direction only) HERE, where the wave would
            // otherwise wait for round 1 of the rows -- they used to open the output epilogue, exposed (8 registers across the layers)
            wn_hidden(P.wn, dx, dy, dz, t2);
            __builtin_amdgcn_sched_barrier(0);      // (round 1's reads wait for the DMA: hipcc would hoist them, and the wait, above the four blocks)
            cv_rows_read<1>(rows, col, hh, h);
            cv_layer1_blocks<4, 8>(P, q0, q1, b0, b1, hh, col, kinf, h);
        }
        const long nbn = (long)bn * P.n2 + knn_next;      // (no next tile: row 0 of the sample, requested and never read)
        // byte offset of this lane's first 16-byte slot in a (position, 256) row (one 32-bit VGPR on uniform base pointers)
        const long pos = i * 16 + j;
        const unsigned ro = (unsigned)pos * 1024u + 16u * hh;
        if (SAVE && valid) {
            uint2 m[2];
            split_sign_masks(h, m);
            P.mk1[pos * 4 + hh] = m[0];
            P.mk1[pos * 4 + 2 + hh] = m[1];
        }
        f16v acc[SPLIT_VB];
        f4 bq[2][8];
        // a1 goes out while it is being consumed; the next tile's first round of rows is requested; the bias arrives during the last k-step
        LaneScale sc = lane_scale32(h);
        if (SAVE && P.amax) tensor_amax_update(P.amax, sc.mb);
        split_layer<0>(ws, h, sc.s, acc, SidePair<CvLayer2Side<SAVE>, BiasSide>{{StoreRowsSide{P.sv1, ro, valid}, CvRowsRequest(P.p2, rows, (int)nbn, 0, lane)},
                                                                              BiasSide{P.bias2 + 4 * hh, bq}});
        float c = sc.inv * wi2;
        split_epilogue(acc, P.bias2 + 4 * hh, bq, [&](int e, f4 a, f4 b) { h[e] = leaky_med4(scale_bias4(a, c, b), kinf); });
        if (SAVE && valid) {
            uint2 m[2];
            split_sign_masks(h, m);
            P.mk2[pos * 4 + hh] = m[0];
            P.mk2[pos * 4 + 2 + hh] = m[1];
        }
        sc = lane_scale32(h);
        if (SAVE && P.amax) tensor_amax_update(P.amax + 1, sc.mb);
        if (SAVE) split_layer<SPLIT_NF>(ws, h, sc.s, acc, SidePair<StoreRowsSide, BiasSide>{StoreRowsSide{P.sv2, ro, valid}, BiasSide{P.bias3 + 4 * hh, bq}});
        else split_layer<SPLIT_NF>(ws, h, sc.s, acc, BiasSide{P.bias3 + 4 * hh, bq});
        ws.sync();                                                   // wrap the stream to chunk 0
        c = sc.inv * wi3;
        split_epilogue(acc, P.bias3 + 4 * hh, bq, [&](int e, f4 a, f4 b) { h[e] = leaky_med4(scale_bias4(a, c, b), kinf); });      // a3
        if (SAVE && valid) {
#pragma unroll
            for (int e = 0; e < 32; ++e) *cv_at(P.sv3, ro + 32u * e) = h[e];
        }
        WnBlock wk = wn_block(P.wn, 0, hh, col);
        // out[i] = sum over the 16 neighbours of relu(Wc.t2 + bc) * a3, one 32-channel block at a time; block v + 1's four
        // dependent MFMAs (K = 8 in steps of 2) run under block v's VALU / DPP work, block v + 2's operands travel meanwhile
        float *o = P.out + i * P.out_pitch + 4 * hh;
        f16v wpre = wn_pre(wk, hh, t2);
        wk = wn_block(P.wn, 1, hh, col);
        auto out_block = [&](int v) {
            const f16v w = wpre;
            if (v + 1 < SPLIT_VB) wpre = wn_pre(wk, hh, t2);
            if (v + 2 < SPLIT_VB) wk = wn_block(P.wn, v + 2, hh, col);
            f4 r[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) r[q][e] = relu1(w[4 * q + e], kinf) * h[4 * v + q][e];
            // sum over the 16 neighbours by the transposing reduction (fused_common.h): lane j ends with slot q = 2 (bit 3 of j) + (bit 2
            // of j) of the block, the first lane of each quad stores -- 32 cross-lane operations and one store instead of 64 and four
            const f4 t = row_sum16_transpose4(r[0], r[1], r[2], r[3]);
            if (valid && (j & 3) == 0) *reinterpret_cast<f4 *>(o + 32 * v + 8 * (2 * ((j >> 3) & 1) + ((j >> 2) & 1))) = t;
        };
        out_block(0);
        out_block(1);
        // ---- next tile: its direction and its p1 slots -------------------------------------------------------------------------
        float cn[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};              // the six coordinates; subtracted after the last block (the
        f4 q0n = q0, q1n = q1;                                        // subtraction is where the wave waits for them)
        if (more) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { cn[c] = ldc(P.xyz2 + nbn * 3 + c); cn[3 + c] = ldc(P.xyz1 + in_ * 3 + c); }
            q0n = ldc4(P.p1 + in_ * 256 + 4 * hh + 8 * j); q1n = ldc4(P.p1 + in_ * 256 + 4 * hh + 8 * (16 + j));
        }
#pragma unroll
        for (int v = 2; v < SPLIT_VB; ++v) out_block(v);
        pt = ptn; valid = validn; i = in_; nb = nbn; q0 = q0n; q1 = q1n;
        dx = __fsub_rn(cn[0], cn[3]); dy = __fsub_rn(cn[1], cn[4]); dz = __fsub_rn(cn[2], cn[5]);
    }
    ws.finish();
}

// ---- rtk_cost_volume_bwd on the split path ------------------------------------------------------------------------------
// Same outputs as cost_volume_bwd_kernel (fused_group.hip) from the same saved tensors; the tile of the split forward.  The two
// transposed 256 x 256 products (W3^T dz3, W2^T dz2) run split; the WeightNet's hidden gradient dt2 = Wc^T dq3 (8 x 256 per
// position) runs on the fp32-input MFMA against an LDS image of Wc^T built once per workgroup.
struct CvSplitBwdParams {
    CvSplitParams f;              // geometry, WeightNet, blob = split images of W3^T, W2^T
    const float *dout;
    int dout_pitch;
    const float *a3;
    const uint2 *mk1, *mk2;
    float *dz1, *dz2, *dz3, *dq3, *d4, *dp1, *dpd, *dt2, *dbrows;
};

__device__ __forceinline__ f4 leaky_grad_bits4(f4 d, unsigned bits) {     // d * leaky'(z), [z > 0] in the low four bits
    return (f4){(bits & 1u) ? d.x : 0.1f * d.x, (bits & 2u) ? d.y : 0.1f * d.y, (bits & 4u) ? d.z : 0.1f * d.z, (bits & 8u) ? d.w : 0.1f * d.w};
}

__global__ __launch_bounds__(64 * SP_NW) __attribute__((amdgpu_waves_per_eu(1, 1))) void cost_volume_bwd_split_kernel(const CvSplitBwdParams Q) {
    const float kinf = rtk_hidden_inf();
    __shared__ __attribute__((aligned(16))) f4 s_w[2 * SP_F * 64];
    __shared__ float s_wct[128 * 64];                                    // [(v, q, r)][lane = 32 hh + o]: Wc[32 v + 8 q + 4 hh + r][o] (o < 8, else 0)
    const CvSplitParams &P = Q.f;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), hh = lane >> 5, col = lane & 31, pp = col >> 4,
              j = col & 15;
    const int jq = 2 * ((j >> 3) & 1) + ((j >> 2) & 1);               // the slot a lane holds after a transposing reduction of four
    for (int e = threadIdx.x; e < 128 * 64; e += 64 * SP_NW) {
        const int l = e & 63, st = e >> 6, o = l & 31, ch = 32 * (st >> 4) + 8 * ((st >> 2) & 3) + 4 * (l >> 5) + (st & 3);
        s_wct[e] = o < 8 ? P.wn.wc[((ch >> 4) * 64 + (o >> 2) * 16 + (ch & 15)) * 4 + (o & 3)] : 0.f;
    }
    int b, bx, nbx;
    rtk_decode_block(P.gx, b, bx, nbx);
    constexpr int PPW = 2 * SP_NW;
    const int groups = (P.n1 + PPW - 1) / PPW;
    WStreamA<SP_NW, SP_F, 2 * SPLIT_NF> ws;
    ws.start_parts(P.blob, s_w, wave, lane);                             // (its barrier also publishes s_wct)
    const float wi3 = ldc(P.wsc), wi2 = ldc(P.wsc + 1);                  // images: W3^T, W2^T
    // software-pipelined by one tile like the forward kernel: the next tile's neighbour index and direction are requested in this
    // tile's epilogue (the dz1 / neighbour-sum phase), not in front of its own first loads
    int pt = bx * PPW + 2 * wave + pp;
    bool valid = pt < P.n1;
    long i = (long)b * P.n1 + (valid ? pt : P.n1 - 1);
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (bx < groups) {
        const long nb = (long)b * P.n2 + (long)P.knn[i * 16 + j];
        dx = __fsub_rn(P.xyz2[nb * 3], P.xyz1[i * 3]); dy = __fsub_rn(P.xyz2[nb * 3 + 1], P.xyz1[i * 3 + 1]);
        dz = __fsub_rn(P.xyz2[nb * 3 + 2], P.xyz1[i * 3 + 2]);
    }
    for (int G = bx; G < groups; G += nbx) {
        asm volatile("" ::: "memory");
        const long pos = i * 16 + j;
        const unsigned ro = (unsigned)pos * 1024u + 16u * hh;
        if (valid && hh == 0) *reinterpret_cast<f4 *>(Q.d4 + pos * 4) = (f4){dx, dy, dz, 1.0f};
        uint2 m2[2] = {Q.mk2[pos * 4 + hh], Q.mk2[pos * 4 + 2 + hh]}, m1[2] = {Q.mk1[pos * 4 + hh], Q.mk1[pos * 4 + 2 + hh]};
        // a3 = leaky(z3).  (Through LDS as the forward's p2 rows -- whole lines, once -- the phase gains 3.6 k clocks per tile and the
        // two layers lose 6 k: next to this kernel's 32 KiB Wc^T image the rows only fit with 24 KiB halves of the weight buffer,
        // whose twice as many chunk boundaries each wait for the dz stores in flight; and the next tile's rows come from HBM, not L2:
        // requested as a layer's side job they are too young at three boundaries in a row: +9 k.  Not kept.)
        f4 h[32];
#pragma unroll
        for (int e = 0; e < 32; ++e) h[e] = *cv_at(Q.a3, ro + 32u * e);
        // ---- out = sum_k wn * a3:  dz3 = dout wn leaky'(z3),  dq3 = dout a3 [wn > 0],  dt2 = Wc^T dq3 -------------------------
        float t2[8];
        WnBlock wk = wn_block(P.wn, 0, hh, col);
        // the dout row of a point is the same for its 16 neighbours: each of them loads two of its 32 slots, the products take the
        // owner's copy through DPP (mul_row_bcast)
        const float *dor = Q.dout + i * Q.dout_pitch + 4 * hh;
        const f4 dq0 = ldc4(dor + 8 * j), dq1 = ldc4(dor + 8 * (16 + j));
        wn_hidden(P.wn, dx, dy, dz, t2);
        f16v dt2;
#pragma unroll
        for (int e = 0; e < 16; ++e) dt2[e] = 0.f;
        f16v wpre = wn_pre(wk, hh, t2);
        wk = wn_block(P.wn, 1, hh, col);
        auto block_a = [&](auto vc) {
            constexpr int v = decltype(vc)::value;
            const f16v w = wpre;                                     // block v + 1's four MFMAs and block v + 2's operands travel under block v
            if constexpr (v + 1 < SPLIT_VB) wpre = wn_pre(wk, hh, t2);
            if constexpr (v + 2 < SPLIT_VB) wk = wn_block(P.wn, v + 2, hh, col);
            f4 zs[4];
            auto slot = [&](auto qc) {
                constexpr int q = decltype(qc)::value, e = 4 * v + q;
                const f4 a = h[e];
                f4 wv;
#pragma unroll
                for (int r = 0; r < 4; ++r) wv[r] = relu1(w[4 * q + r], kinf);
                const f4 da = mul_row_bcast<e & 15>(e < 16 ? dq0 : dq1, a);       // d * a
                const f4 t = mul_row_bcast<e & 15>(e < 16 ? dq0 : dq1, wv);       // d * relu(w)   (== d * w wherever w > 0, +-0 elsewhere)
                f4 qq, z;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    qq[r] = wv[r] > 0.f ? da[r] : 0.f;
                    z[r] = a[r] > 0.f ? t[r] : 0.1f * t[r];
                    dt2 = mfma_f32x2(s_wct[(16 * v + 4 * q + r) * 64 + lane], qq[r], dt2);
                }
                h[e] = z;
                zs[q] = z;
                if (valid) *cv_at(Q.dq3, ro + 32u * e) = qq;       // (dz3 goes out during the product that consumes it)
            };
            slot(std::integral_constant<int, 0>{}); slot(std::integral_constant<int, 1>{});
            slot(std::integral_constant<int, 2>{}); slot(std::integral_constant<int, 3>{});
            if (Q.dbrows) {                          // (uniform) per-query neighbour sums of the block's four slots by ONE transposing reduction
                const f4 r = row_sum16_transpose4(zs[0], zs[1], zs[2], zs[3]);      // (fused_common.h: lane j ends with slot jq; the host's bias sum shrinks 16x)
                if (valid && (j & 3) == 0) *reinterpret_cast<f4 *>(Q.dbrows + i * 512 + 32 * v + 8 * jq + 4 * hh) = r;
            }
        };
        block_a(std::integral_constant<int, 0>{}); block_a(std::integral_constant<int, 1>{}); block_a(std::integral_constant<int, 2>{});
        block_a(std::integral_constant<int, 3>{}); block_a(std::integral_constant<int, 4>{}); block_a(std::integral_constant<int, 5>{});
        block_a(std::integral_constant<int, 6>{}); block_a(std::integral_constant<int, 7>{});
        if (valid) *reinterpret_cast<f4 *>(Q.dt2 + pos * 8 + 4 * hh) = (f4){dt2[0], dt2[1], dt2[2], dt2[3]};      // rows 4 hh .. + 3 of the 8 hidden units
        // ---- da2 = W3^T dz3;  dz2 = da2 leaky'(z2) ----------------------------------------------------------------------------
        f16v acc[SPLIT_VB];
        LaneScale sc = lane_scale32(h);          // gradients span many orders of magnitude: the position's own scale is what keeps 22 bits
        if (P.amax) tensor_amax_update(P.amax, sc.mb);
        split_layer<0>(ws, h, sc.s, acc, StoreRowsSide{Q.dz3, ro, valid});
        float c = sc.inv * wi3;
#pragma unroll
        for (int v = 0; v < SPLIT_VB; ++v) {
#pragma unroll
            for (int q = 0; q < 4; ++q) h[4 * v + q] = leaky_grad_bits4(acc_slot(acc, 4 * v + q) * c, split_mask_bits(m2, v, q));
            if (Q.dbrows) {
                const f4 r = row_sum16_transpose4(h[4 * v], h[4 * v + 1], h[4 * v + 2], h[4 * v + 3]);
                if (valid && (j & 3) == 0) *reinterpret_cast<f4 *>(Q.dbrows + i * 512 + 256 + 32 * v + 8 * jq + 4 * hh) = r;
            }
        }
        // ---- da1 = W2^T dz2;  dz1 = da1 leaky'(z1);  dp1 = sum over the 16 neighbours; per-query partials of dWd = dz1^T d -----
        sc = lane_scale32(h);
        if (P.amax) tensor_amax_update(P.amax + 1, sc.mb);
        split_layer<SPLIT_NF>(ws, h, sc.s, acc, StoreRowsSide{Q.dz2, ro, valid});
        ws.sync();                                                   // wrap the stream to chunk 0
        c = sc.inv * wi2;
        // ---- next tile: neighbour index now, direction half way through the epilogue ---------------------------------------
        const int Gn = G + nbx;
        const bool more = Gn < groups;
        const int ptn = Gn * PPW + 2 * wave + pp;
        const bool validn = ptn < P.n1;
        const long in_ = (long)b * P.n1 + (validn ? ptn : P.n1 - 1);
        long knn_next = 0;
        if (more) knn_next = (long)P.knn[in_ * 16 + j];
        float dxn = 0.f, dyn = 0.f, dzn = 0.f;
        float *dpr = Q.dp1 + i * 256 + 4 * hh;
        float *dpd = Q.dpd + i * 768 + 4 * hh;
#pragma unroll
        for (int v = 0; v < SPLIT_VB; ++v) {
            if (v == 2 && more) {
                const long nbn = (long)b * P.n2 + knn_next;
                dxn = __fsub_rn(P.xyz2[nbn * 3], P.xyz1[in_ * 3]); dyn = __fsub_rn(P.xyz2[nbn * 3 + 1], P.xyz1[in_ * 3 + 1]);
                dzn = __fsub_rn(P.xyz2[nbn * 3 + 2], P.xyz1[in_ * 3 + 2]);
            }
            // the block's four slots: dz1 out, then the neighbour sums of dz1 and of dz1 (x) d, each by ONE transposing reduction of the
            // four slots (32 cross-lane operations instead of 64; lane j ends with slot jq and the first lane of each quad stores)
            f4 r[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                r[q] = leaky_grad_bits4(acc_slot(acc, 4 * v + q) * c, split_mask_bits(m1, v, q));
                if (valid) *cv_at(Q.dz1, ro + 32u * (4 * v + q)) = r[q];
            }
            const f4 s0 = row_sum16_transpose4(r[0], r[1], r[2], r[3]);
            const f4 sx = row_sum16_transpose4(r[0] * dx, r[1] * dx, r[2] * dx, r[3] * dx);
            const f4 sy = row_sum16_transpose4(r[0] * dy, r[1] * dy, r[2] * dy, r[3] * dy);
            const f4 sz = row_sum16_transpose4(r[0] * dz, r[1] * dz, r[2] * dz, r[3] * dz);
            if (valid && (j & 3) == 0) {
                *reinterpret_cast<f4 *>(dpr + 32 * v + 8 * jq) = s0;
                *reinterpret_cast<f4 *>(dpd + 32 * v + 8 * jq) = sx;
                *reinterpret_cast<f4 *>(dpd + 256 + 32 * v + 8 * jq) = sy;
                *reinterpret_cast<f4 *>(dpd + 512 + 32 * v + 8 * jq) = sz;
            }
        }
        pt = ptn; valid = validn; i = in_; dx = dxn; dy = dyn; dz = dzn;
    }
    ws.finish();
}

// ---- rtk_sa_scale on the split path: the two-layer scales with 64 output channels (sa2 scale 1, sa3 scales 0 and 1) -------------
// A wave owns 32 (centroid, neighbour) positions: one centroid of 32 neighbours or two of 16.  The offset layer (K = 4 with the
// bias column) runs on the fp32-input MFMA, the C1 -> 64 layer split with its image RESIDENT in LDS (6 C1 x 64 bytes: no stream,
// no barriers after the fill), the max over the neighbours in registers.
struct SaSplitParams {
    int samples, n, npoint, gx;
    const float *xyz, *new_xyz;
    const int *idx;
    const float *q;
    int q_pitch;
    const float *w1;        // offset layer image [C1 / 16][64] of [Wx | b1]
    const f4 *image;        // split image of the C1 -> 64 layer ...
    const float *wsc;       // ... and the inverse of its power-of-two weight scale
    const float *bias2;
    float *out;
    int out_pitch, out_offset;
    const int *src_nuniq, *dst_nuniq;
};

// Register budget: asked for five waves per SIMD the three shapes allocate 86 / 84 / 86 registers without a spill (136 / 132 / 120
// unconstrained) and fit on a SIMD next to another batch's forward cost volume (404 of 512): +2-3 % frame-pairs/s in the
// pipelined forward.  (The first capped build computed wrong maxima: see the epilogue.)
#ifndef SA_SPLIT_WAVES
#define SA_SPLIT_WAVES 5
#endif
#ifndef SA_WGS_TARGET
#define SA_WGS_TARGET 1024
#endif
template <int NS, int C1>
__global__ __launch_bounds__(256, SA_SPLIT_WAVES) void sa_scale_split_kernel(const SaSplitParams P) {
    constexpr int KS = C1 / 16, VB1 = C1 / 32, NFR = KS * 2 * 2, CPT = 32 / NS;      // k-steps, 32-blocks of layer 1, fragments, centroids per tile
    const float kinf = rtk_hidden_inf();
    __shared__ __attribute__((aligned(16))) f4 s_img[NFR * 64];
    const int lane = threadIdx.x & 63, hh = lane >> 5, col = lane & 31, pp = col / NS, slot = col % NS;
    int b, bx, nbx;
    rtk_decode_block(P.gx, b, bx, nbx);
    const int dst_e = P.dst_nuniq ? __builtin_amdgcn_readfirstlane(P.dst_nuniq[b]) : P.npoint;
    const int live_c = min(dst_e, P.npoint);                      // centroids >= live_c are duplicates of centroid 0: neither computed nor written
    const int live_units = (live_c + CPT - 1) / CPT;
    if (bx * 4 >= live_units) return;
    for (int i = threadIdx.x; i < NFR * 64; i += blockDim.x) s_img[i] = P.image[i];
    __syncthreads();
    const int src_e = P.src_nuniq ? P.src_nuniq[b] : 0x7fffffff;
    const float winv = ldc(P.wsc);
    for (int unit = bx * 4 + (threadIdx.x >> 6); unit < live_units; unit += nbx * 4) {
        int cl = unit * CPT + pp;                                  // centroid of this lane within the sample
        const bool valid = cl < live_c;
        if (!valid) cl = live_c - 1;
        const int c = b * P.npoint + cl;
        const int id = P.idx[(long)c * NS + slot];
        const long src = (long)b * P.n + id;
        const float dx = __fsub_rn(P.xyz[src * 3], P.new_xyz[(long)c * 3]), dy = __fsub_rn(P.xyz[src * 3 + 1], P.new_xyz[(long)c * 3 + 1]),
                    dz = __fsub_rn(P.xyz[src * 3 + 2], P.new_xyz[(long)c * 3 + 2]);
        // layer 1: relu(q[id] + Wx.d + b1)      (duplicate source rows alias row 0)
        const float *qrow = P.q + ((long)b * P.n + (id < src_e ? id : 0)) * P.q_pitch + 4 * hh;
        const float b0 = hh ? dy : dx, b1 = hh ? 1.0f : dz;
        f4 h[4 * VB1];
#pragma unroll
        for (int v = 0; v < VB1; ++v) {
            f16v a;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f4 t = *reinterpret_cast<const f4 *>(qrow + 32 * v + 8 * q);
                a[4 * q] = t.x; a[4 * q + 1] = t.y; a[4 * q + 2] = t.z; a[4 * q + 3] = t.w;
            }
            const int ch = 32 * v + col;
            const float *wr = P.w1 + (ch >> 4) * 64 + (ch & 15);
            a = mfma_f32x2(wr[16 * hh], b0, a);
            a = mfma_f32x2(wr[16 * (2 + hh)], b1, a);
#pragma unroll
            for (int q = 0; q < 4; ++q) h[4 * v + q] = relu_med4((f4){a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]}, kinf);
        }
        // layer 2 (C1 -> 64) on the split path, the position's activations scaled by its own power of two; ReLU after the max
        const LaneScale sc = lane_scale32(h);
        const float cs = sc.inv * winv;
        f16v acc[2];
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[v][e] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            u4v bp[2];
            split2(h[2 * s], h[2 * s + 1], sc.s, bp);
            u4v fr[2][2];
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int p = 0; p < 2; ++p) fr[v][p] = __builtin_bit_cast(u4v, s_img[((s * 2 + v) * 2 + p) * 64 + lane]);
#define RTK_SA_MM(pa, pb) acc[0] = mfma_h(fr[0][pa], bp[pb], acc[0]); acc[1] = mfma_h(fr[1][pa], bp[pb], acc[1]);
            RTK_SA_MM(1, 0) RTK_SA_MM(0, 1) RTK_SA_MM(0, 0)
#undef RTK_SA_MM
        }
        // The maximum over the neighbours by a transposing reduction (fused_common.h): the eight f4 of a lane (v, q) end as ONE (32
        // neighbours: the two v through v_permlane16_swap) or two (16 neighbours); lane (hh, col) then holds q = 2 (bit 3 of col) +
        // (bit 2 of col) and -- 32 neighbours -- v = bit 4 of col, and the first lane of each quad stores.  Bias and scale BEFORE the
        // maximum (an fma the compiler knows: it places the wait states an MFMA result needs; the position's scale differs lane by lane).
        f4 m[2][4];
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                m[v][q] = scale_bias4((f4){acc[v][4 * q], acc[v][4 * q + 1], acc[v][4 * q + 2], acc[v][4 * q + 3]}, cs,
                                      *reinterpret_cast<const f4 *>(P.bias2 + 32 * v + 8 * q + 4 * hh));
        const int qo = 2 * ((col >> 3) & 1) + ((col >> 2) & 1);
        float *o = P.out + (long)c * P.out_pitch + P.out_offset + 4 * hh + 8 * qo;
        if constexpr (NS == 32) {
            f4 t[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) t[q] = wave_max_pair16(m[0][q], m[1][q], kinf);
            const f4 r = relu_med4(row_max16_transpose4(t[0], t[1], t[2], t[3]), kinf);
            if (valid && (col & 3) == 0) *reinterpret_cast<f4 *>(o + 32 * ((col >> 4) & 1)) = r;
        } else {
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const f4 r = relu_med4(row_max16_transpose4(m[v][0], m[v][1], m[v][2], m[v][3]), kinf);
                if (valid && (col & 3) == 0) *reinterpret_cast<f4 *>(o + 32 * v) = r;
            }
        }
    }
}

}  // namespace

extern "C" int rtk_pack_split_layer(int cout, int cin, const float *w, int transposed, void *image, float *inv_scale, rtk_stream_t stream) {
    RTK_REQUIRE(cout > 0 && cin > 0 && cout % 32 == 0 && cin % 32 == 0 && w && image && inv_scale, "pack_split_layer: bad arguments");
    RTK_REQUIRE((long)cout * cin <= (1L << 22), "pack_split_layer: more than 2^22 weights (every workgroup scans the matrix for its scale)");
    const int slots = (cin / 16) * (cout / 32) * 64;
    pack_split_kernel<<<(slots + 255) / 256, 256, 0, (hipStream_t)stream>>>(cout, cin, w, transposed, (u4v *)image, inv_scale);
    RTK_CHECK_LAUNCH("pack_split_layer");
    return RTK_OK;
}

extern "C" int rtk_split_mlp2(int positions, const float *x, const void *images, const float *image_scales, const float *bias1,
                              const float *bias2, float *y, rtk_stream_t stream) {
    RTK_REQUIRE(positions > 0 && x && images && image_scales && bias1 && bias2 && y, "split_mlp2: bad arguments");
    const int tiles = (positions + 32 * SP_NW - 1) / (32 * SP_NW);
    split_mlp2_kernel<<<tiles < 256 ? tiles : 256, 64 * SP_NW, 0, (hipStream_t)stream>>>(positions, x, (const f4 *)images, image_scales, bias1, bias2, y);
    RTK_CHECK_LAUNCH("split_mlp2");
    return RTK_OK;
}

static int cv_split_fill(const char *who, CvSplitParams &P, int samples, int n1, int n2, const float *xyz1, const float *xyz2,
                         const int64_t *knn_idx, const void *split_images, const float *image_scales, const rtk_layer_t *wn, dim3 &grid) {
    RTK_REQUIRE(samples > 0 && samples <= 65535 && n1 > 0 && n2 >= 16 && xyz1 && xyz2 && knn_idx && split_images && image_scales,
                "%s: bad arguments", who);
    RTK_REQUIRE(wn && wn[0].w_packed && wn[1].w_packed && wn[2].w_packed && wn[1].bias && wn[2].bias && wn[1].cin16 == 1 &&
                wn[1].cout16 == 1 && wn[2].cin16 == 1 && wn[2].cout16 == 16, "%s: bad WeightNet layers", who);
    P.samples = samples; P.n1 = n1; P.n2 = n2;
    P.xyz1 = xyz1; P.xyz2 = xyz2; P.knn = knn_idx;
    P.blob = reinterpret_cast<const f4 *>(split_images);
    P.wsc = image_scales;
    P.wn.wa = wn[0].w_packed; P.wn.wb = wn[1].w_packed; P.wn.wc = wn[2].w_packed; P.wn.bb = wn[1].bias; P.wn.bc = wn[2].bias;
    P.p1 = P.p2 = P.wd = P.bias2 = P.bias3 = nullptr;
    P.out = nullptr; P.out_pitch = 0; P.sv1 = P.sv2 = P.sv3 = nullptr; P.mk1 = P.mk2 = nullptr; P.amax = nullptr;
    const int groups = (n1 + 2 * SP_NW - 1) / (2 * SP_NW);
    // one workgroup per CU (the tile keeps the whole register file); the rest is looped.  (Measured and rejected: 2 or 4 queued
    // workgroups per CU to shorten the tail when other kernels of the pipelined batches hold CUs -- 1 to 1.5 % slower.)
    int gx = 256 / samples;
    if (gx < 1) gx = 1;
    if (gx > groups) gx = groups;
    P.gx = samples % 8 == 0 ? gx : 0;
    grid = P.gx ? dim3(gx * samples) : dim3(gx, samples);
    return RTK_OK;
}

// compute units of the current device (one workgroup of the cost-volume kernels per CU); 256 if the runtime does not say
static int cu_count() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) return 256;
    return n;
}

static int cv_split_forward(const char *who, int samples, int n1, int n2, const float *xyz1, const float *xyz2, const int64_t *knn_idx,
                            const float *p1, const float *p2, const float *wd_packed, const void *split_images, const float *image_scales,
                            const float *bias2, const float *bias3, const rtk_layer_t *wn, float *out, int out_pitch, float *a1, float *a2,
                            float *a3, void *mask1, void *mask2, float *amax, int workgroups, rtk_stream_t stream) {
    CvSplitParams P;
    dim3 grid;
    if (cv_split_fill(who, P, samples, n1, n2, xyz1, xyz2, knn_idx, split_images, image_scales, wn, grid) != RTK_OK) return RTK_ERR_INVALID;
    RTK_REQUIRE(p1 && p2 && wd_packed && bias2 && bias3 && out, "%s: bad arguments", who);
    RTK_REQUIRE((double)samples * n2 <= 4194304.0, "%s: more than 2^22 rows in p2 (32-bit byte offsets of the row requests): split the batch", who);
    RTK_REQUIRE(out_pitch % 4 == 0 && out_pitch >= 256, "%s: bad out_pitch", who);
    const bool save = a1 != nullptr;
    RTK_REQUIRE(!save || ((double)samples * n1 * 16.0 * 1024.0 < 4294967296.0), "%s: more than 4 GiB per saved activation (32-bit row "
                "offsets): split the batch", who);
    P.p1 = p1; P.p2 = p2; P.wd = wd_packed; P.bias2 = bias2; P.bias3 = bias3; P.out = out; P.out_pitch = out_pitch;
    P.sv1 = a1; P.sv2 = a2; P.sv3 = a3; P.mk1 = (uint2 *)mask1; P.mk2 = (uint2 *)mask2; P.amax = amax;
    if (samples % 8 == 0) {      // flattened tiles (see the kernel): `workgroups` of them, a multiple of 8, at most one per tile
        const int tiles_x = (samples / 8) * ((n1 + 2 * SP_NW - 1) / (2 * SP_NW));
        int per_xcd = (workgroups > 0 ? workgroups : cu_count()) / 8;
        if (per_xcd < 1) per_xcd = 1;
        if (per_xcd > tiles_x) per_xcd = tiles_x;
        P.gx = 8 * per_xcd;
        grid = dim3(P.gx);
    }
    if (save) cost_volume_split_kernel<true><<<grid, 64 * SP_NW, 0, (hipStream_t)stream>>>(P);
    else cost_volume_split_kernel<false><<<grid, 64 * SP_NW, 0, (hipStream_t)stream>>>(P);
    RTK_CHECK_LAUNCH(who);
    return RTK_OK;
}

extern "C" int rtk_cost_volume_split(int samples, int n1, int n2, const float *xyz1, const float *xyz2, const int64_t *knn_idx,
                                     const float *p1, const float *p2, const float *wd_packed, const void *split_images,
                                     const float *image_scales, const float *bias2, const float *bias3, const rtk_layer_t *wn, float *out,
                                     int out_pitch, rtk_stream_t stream) {
    return cv_split_forward("cost_volume_split", samples, n1, n2, xyz1, xyz2, knn_idx, p1, p2, wd_packed, split_images, image_scales, bias2, bias3,
                            wn, out, out_pitch, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, stream);
}

extern "C" int rtk_cost_volume_split_shared(int samples, int n1, int n2, const float *xyz1, const float *xyz2, const int64_t *knn_idx,
                                            const float *p1, const float *p2, const float *wd_packed, const void *split_images,
                                            const float *image_scales, const float *bias2, const float *bias3, const rtk_layer_t *wn,
                                            float *out, int out_pitch, int workgroups, rtk_stream_t stream) {
    RTK_REQUIRE(workgroups >= 0, "cost_volume_split_shared: workgroups = %d", workgroups);
    return cv_split_forward("cost_volume_split_shared", samples, n1, n2, xyz1, xyz2, knn_idx, p1, p2, wd_packed, split_images, image_scales,
                            bias2, bias3, wn, out, out_pitch, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, workgroups, stream);
}

extern "C" int rtk_cost_volume_split_train(int samples, int n1, int n2, const float *xyz1, const float *xyz2, const int64_t *knn_idx,
                                           const float *p1, const float *p2, const float *wd_packed, const void *split_images,
                                           const float *image_scales, const float *bias2, const float *bias3, const rtk_layer_t *wn,
                                           float *out, int out_pitch, float *a1, float *a2, float *a3, void *mask1, void *mask2,
                                           float *act_amax, rtk_stream_t stream) {
    RTK_REQUIRE(a1 && a2 && a3 && mask1 && mask2, "cost_volume_split_train: null activation buffer");
    return cv_split_forward("cost_volume_split_train", samples, n1, n2, xyz1, xyz2, knn_idx, p1, p2, wd_packed, split_images, image_scales,
                            bias2, bias3, wn, out, out_pitch, a1, a2, a3, mask1, mask2, act_amax, 0, stream);
}

extern "C" int rtk_cost_volume_bwd_split(int samples, int n1, int n2, const float *xyz1, const float *xyz2, const int64_t *knn_idx,
                                         const void *split_images_t, const float *image_scales_t, const rtk_layer_t *wn, const float *dout,
                                         int dout_pitch,
                                         const float *a3, const void *mask1, const void *mask2, float *dz1, float *dz2, float *dz3,
                                         float *dq3, float *d4, float *dp1, float *dpd, float *dt2, float *dbias_rows,
                                         float *dz_amax, rtk_stream_t stream) {
    CvSplitBwdParams Q;
    dim3 grid;
    if (cv_split_fill("cost_volume_bwd_split", Q.f, samples, n1, n2, xyz1, xyz2, knn_idx, split_images_t, image_scales_t, wn, grid) != RTK_OK)
        return RTK_ERR_INVALID;
    RTK_REQUIRE(dout && mask1 && mask2 && a3 && dz1 && dz2 && dz3 && dq3 && d4 && dp1 && dpd && dt2, "cost_volume_bwd_split: bad arguments");
    RTK_REQUIRE((double)samples * n1 * 16.0 * 1024.0 < 4294967296.0, "cost_volume_bwd_split: more than 4 GiB per (position, 256) tensor "
                "(32-bit row offsets): split the batch");
    RTK_REQUIRE(dout_pitch % 4 == 0 && dout_pitch >= 256, "cost_volume_bwd_split: bad dout_pitch");
    Q.dout = dout; Q.dout_pitch = dout_pitch; Q.a3 = a3; Q.mk1 = (const uint2 *)mask1; Q.mk2 = (const uint2 *)mask2;
    Q.dz1 = dz1; Q.dz2 = dz2; Q.dz3 = dz3; Q.dq3 = dq3; Q.d4 = d4; Q.dp1 = dp1; Q.dpd = dpd; Q.dt2 = dt2; Q.dbrows = dbias_rows;
    Q.f.amax = dz_amax;
    cost_volume_bwd_split_kernel<<<grid, 64 * SP_NW, 0, (hipStream_t)stream>>>(Q);
    RTK_CHECK_LAUNCH("cost_volume_bwd_split");
    return RTK_OK;
}

extern "C" int rtk_sa_scale_split(int samples, int n, int npoint, int nsample, const float *xyz, const float *new_xyz, const int *idx,
                                  const float *q, int q_pitch, int c1, const float *w1xyz_packed, const void *split_image,
                                  const float *image_scale, const float *bias2, float *out, int out_pitch, int out_offset,
                                  const int *src_nuniq, const int *dst_nuniq, rtk_stream_t stream) {
    RTK_REQUIRE(samples > 0 && n > 0 && npoint > 0 && xyz && new_xyz && idx && q && w1xyz_packed && split_image && image_scale && bias2 && out,
                "sa_scale_split: bad arguments");
    RTK_REQUIRE(q_pitch % 4 == 0 && out_pitch % 4 == 0 && out_offset % 4 == 0, "sa_scale_split: pitches/offset must be multiples of 4");
    RTK_REQUIRE((long)samples * npoint * nsample < 0x7fffffffL && (long)samples * n < 0x7fffffffL && samples <= 65535,
                "sa_scale_split: problem too large for 32-bit indexing");
    SaSplitParams P;
    P.samples = samples; P.n = n; P.npoint = npoint;
    P.xyz = xyz; P.new_xyz = new_xyz; P.idx = idx; P.q = q; P.q_pitch = q_pitch; P.w1 = w1xyz_packed;
    P.image = reinterpret_cast<const f4 *>(split_image); P.wsc = image_scale; P.bias2 = bias2;
    P.out = out; P.out_pitch = out_pitch; P.out_offset = out_offset; P.src_nuniq = src_nuniq; P.dst_nuniq = dst_nuniq;
    const int units = (npoint + 32 / nsample - 1) / (32 / nsample);
    int bx = (units + 3) / 4;
    while ((long)bx * samples > SA_WGS_TARGET && bx > 1) bx = (bx + 1) / 2;      // few, fat workgroups: the LDS image fill is paid per workgroup
    P.gx = samples % 8 == 0 ? bx : 0;
    const dim3 blocks = P.gx ? dim3(bx * samples) : dim3(bx, samples);
    hipStream_t s = (hipStream_t)stream;
    if (nsample == 32 && c1 == 64) sa_scale_split_kernel<32, 64><<<blocks, 256, 0, s>>>(P);
    else if (nsample == 16 && c1 == 64) sa_scale_split_kernel<16, 64><<<blocks, 256, 0, s>>>(P);
    else if (nsample == 16 && c1 == 32) sa_scale_split_kernel<16, 32><<<blocks, 256, 0, s>>>(P);
    else {
        rtk_set_error("sa_scale_split: no kernel instance for nsample=%d, %d -> 64 channels", nsample, c1);
        return RTK_ERR_UNSUPPORTED;
    }
    RTK_CHECK_LAUNCH("sa_scale_split");
    return RTK_OK;
}
