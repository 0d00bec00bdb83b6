// split_mfma.h -- fp32 products on the bf16 matrix pipe: every fp32 operand is the EXACT sum of three bf16 pieces
// (8 + 8 + 8 significant bits, split by truncation), and a product a.b is taken as the six partial products
// a_i.b_j with i + j <= 2, accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  Each bf16 x bf16 product is exact in fp32
// (16-bit significand); the three dropped terms are below 2^-24 |a||b| -- the size of one fp32 rounding -- so the result
// carries the error of an fp32 fmaf chain (tools/experiments/exp_split_error.py: 5.0e-7 of max|y| on a 256-deep product against 6.4e-7
// for an fp32 GEMM, both measured against float64).
//
// Why: gfx950 has no xf32 and its fp32-input MFMA runs at the VECTOR rate (157 TFLOP/s, 1/16 of bf16).  The 256 x 256
// layers of the cost volume are bound by exactly that (round 2: 0.73 of the fp32 MFMA peak standalone).  Six bf16
// MFMAs do the work of sixteen fp32 ones: 6/16 of the matrix time for the same bits.
//
// Tile: a wave owns 32 POSITIONS (two query points x 16 neighbours).  Activations stay in registers in the C/D layout
// of the 32x32 MFMA, which again is a legal B layout for the next layer:
//
//     lane = 32 hh + col   (hh = 0..1, col = 0..31)        H[ch = 32 v + 8 q + 4 hh + r][position col] = h[4 v + q][r]
//
// A k-step (16 input channels) takes q in {q0, q0 + 1} of one 32-channel block: lane supplies B[k = 8 hh + t][col],
// t = 4 (q - q0) + r -- eight values it already holds.  The weights are packed with the same permutation
// (rtk_pack_split_layer on the device; pack_layer_split() in ratrack_amd/fused.py is its host restatement), one 1 KiB fragment (64 lanes x 8 bf16) per (k-step, 32-row block, piece):
//
//     frag[s][v][p][lane = 32 hh + i][t] = piece_p( W[32 v + i][32 (s / 2) + 16 (s % 2) + 8 (t / 4) + 4 hh + t % 4] )
#pragma once
#include "fused_common.h"

typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f16v mfma_bf(u4v a, u4v b, f16v c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8v, a), __builtin_bit_cast(bf8v, b), c, 0, 0, 0);
}

// ... one register (two activations) of each piece at a time: the k-step's splitting is spread over its four group steps
template <int W>
__device__ __forceinline__ void split3_word(const f4 x0, const f4 x1, u4v (&b)[3]) {
    const float xa = W < 2 ? x0[2 * W] : x1[2 * W - 4], xb = W < 2 ? x0[2 * W + 1] : x1[2 * W - 3];
    const float ra = __fsub_rn(xa, trunc_bf16(xa)), rb = __fsub_rn(xb, trunc_bf16(xb));
    const float sa = __fsub_rn(ra, trunc_bf16(ra)), sb = __fsub_rn(rb, trunc_bf16(rb));
    b[0][W] = pack_hi16(xa, xb);
    b[1][W] = pack_hi16(ra, rb);
    b[2][W] = pack_hi16(sa, sb);
}

// A chunk of F fragments is F / 6 group steps; the requests for the chunk after it go out during the first split_issue_groups(F)
// of them (the last ones must be old enough at the chunk's closing vmcnt(0) to have made their L2 round trip): three of eight
// (F = 48), two of four (F = 24).
constexpr int split_issue_groups(int F) { return F / 6 > 4 ? 3 : 2; }
constexpr int SPLIT_KS = 16;      // k-steps of a 256-channel contraction
constexpr int SPLIT_VB = 8;       // 32-row output blocks of a 256-channel layer
constexpr int SPLIT_NF = SPLIT_KS * SPLIT_VB * 3;      // fragments (KiB) of one 256 x 256 layer

// The weight stream with LDS reads the compiler does not track.  With LDS-DMA (global_load_lds) in flight hipcc turns every
// wait for an LDS read into s_waitcnt lgkmcnt(0) -- the reads issued a group ahead are waited for at once and every group
// pays the LDS latency (one wave per SIMD: nothing else hides it; measured 0.6 of the MFMA issue rate).  Here the reads are
// inline asm, invisible to the compiler's waitcnt pass, and the group step that uses them waits itself: lgkmcnt(0) at its top
// (lds_wait below) -- the reads were issued a whole group step earlier and have landed, and none of the next group's is
// outstanding yet, so the wait is free and cannot be too short.
template <int NW, int F, int NF>
struct WStreamA : WStream<NW, F, NF> {
    using Base = WStream<NW, F, NF>;
    unsigned rd;      // LDS byte address of this lane's slot in fragment 0 of the resident half
    unsigned next_off;      // byte offset in the blob of the chunk after the resident one (cyclic), computed once per chunk
    __device__ __forceinline__ void set_rd() {
        rd = (unsigned)(size_t)(__attribute__((address_space(3))) void *)(this->lds) + (unsigned)((this->buf * F) * 64 + this->lane) * 16u;
        const int nxt = this->cur + 1 == Base::NCHUNKS ? 0 : this->cur + 1;
        next_off = (unsigned)nxt * (unsigned)(F * 1024);
    }
    __device__ __forceinline__ void start(const f4 *blob_, f4 *lds_, int wave_, int lane_) { Base::start(blob_, lds_, wave_, lane_); set_rd(); }
    __device__ __forceinline__ void next() { Base::next(); set_rd(); }
    // next() in two halves for a workgroup of one wave per SIMD, where nothing hides the ~100 instructions of a chunk's DMA issue
    // if they come in one piece: sync() = the chunk requested during the previous chunk has landed everywhere, flip;
    // issue_part<K, PARTS>() = this wave's K-th share of the requests for the chunk after, one share per group step.
    __device__ __forceinline__ void sync() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        this->buf ^= 1;
        this->cur = this->cur + 1 == Base::NCHUNKS ? 0 : this->cur + 1;
        asm volatile("" : "+s"(this->cur));
        set_rd();
    }
    template <int K, int PARTS>
    __device__ __forceinline__ void issue_part() {
        static_assert(NF % F == 0, "whole chunks only");
        const char *base = this->blob + next_off;
#pragma unroll
        for (int i = K; i < (F + NW - 1) / NW; i += PARTS) {
            const int f = this->wave + i * NW;
            if (F % NW == 0 || f < F)      // (F % NW == 0 folds the test away: a branch per request would cut the group step into blocks)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + (size_t)i * NW * 1024 + this->lane_off),
                                                 (__attribute__((address_space(3))) void *)(this->lds + ((this->buf ^ 1) * F + f) * 64), 16, 0, 0);
        }
    }
    // first chunk resident, nothing else requested yet (the group steps request chunk 1)
    __device__ __forceinline__ void start_parts(const f4 *blob_, f4 *lds_, int wave_, int lane_) {
        this->lds = lds_; this->wave = wave_; this->lane = lane_;
        this->blob = reinterpret_cast<const char *>(blob_);
        this->lane_off = (unsigned)(wave_ * 64 + lane_) * 16u;
        this->cur = 0; this->buf = 0;
        Base::issue(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        set_rd();
    }
    template <int FI>
    __device__ __forceinline__ f4 frag_async() const {
        f4 r;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(rd), "n"(FI * 1024));
        return r;
    }
};

template <int OUTSTANDING>
__device__ __forceinline__ void lds_wait(f4 (&c)[6]) {      // the six fragments are outputs: their uses cannot move above the wait
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]) : "n"(OUTSTANDING));
}

// One group step: the six fragments of two output blocks (three pieces each) against the three B pieces of the k-step: twelve
// MFMAs, the two blocks interleaved so that consecutive MFMAs never wait for each other's accumulator; small terms first.
// The fragments of the next group (and, once per k-step, the B pieces of the next k-step) are fetched/split meanwhile.
// Side job of a layer: called once per group step with the layer's INPUT activations.  The kernels that must also write those
// activations to memory (saved activations of the training forward, dz of the backward) store one 16-byte slot every other group
// step instead of 32 in a burst before the layer.  A CU's store path drains ~7 bytes per cycle; a burst of 32 KiB per wave blocks
// the wave -- the only one on its SIMD -- for as long as that takes, while 16 bytes per lane every other group step stay under
// the drain rate and ride along with the MFMAs.  (Not a vmcnt effect: with every wait of the stream removed the stores of the
// backward cost the same 0.28 ms; and issuing a chunk's share right after its boundary instead changes nothing.)
struct NoSide {
    template <int GI>
    __device__ __forceinline__ void at(const f4 (&)[32]) const {}
};
struct StoreRowsSide {            // h[e] -> 16 bytes at base + ro + 32 e  (this lane's slots of a (position, 256) row)
    float *base;
    unsigned ro;
    bool valid;
    template <int GI>
    __device__ __forceinline__ void at(const f4 (&h)[32]) const {
        if constexpr ((GI & 1) == 0) {
            if (valid) *reinterpret_cast<f4 *>(reinterpret_cast<char *>(base) + (ro + 32u * (GI / 2))) = h[GI / 2];
        }
    }
};

template <int FBASE, int GI, class WS, int F>
struct SplitStep {
    static constexpr int NG = SPLIT_KS * (SPLIT_VB / 2);
    // the six reads of group GJ, two at a time (PART = 0..2), so that they can be placed between the MFMAs of the group before
    template <int GJ, int PART>
    static __device__ __forceinline__ void load_part(WS &ws, f4 (&dst)[6]) {
        constexpr int f0 = FBASE + GJ * 6;      // F % 6 == 0: a group never straddles two chunks
        if constexpr (PART == 0) {
            if constexpr (f0 % F == 0 && f0 != 0) ws.sync();
            if constexpr ((f0 % F) / 6 < split_issue_groups(F)) ws.template issue_part<(f0 % F) / 6, split_issue_groups(F)>();
        }
        dst[2 * PART] = ws.template frag_async<(f0 + 2 * PART) % F>();
        dst[2 * PART + 1] = ws.template frag_async<(f0 + 2 * PART + 1) % F>();
    }
    template <int GJ>
    static __device__ __forceinline__ void load(WS &ws, f4 (&dst)[6]) {
        load_part<GJ, 0>(ws, dst);
        load_part<GJ, 1>(ws, dst);
        load_part<GJ, 2>(ws, dst);
    }
    // Schedule of a group step, pinned: hipcc left alone issues [6 reads, wait, all VALU, 12 MFMAs back to back] -- the matrix pipe
    // idles while the ~25 other instructions issue (one wave per SIMD: nobody else feeds it).  Here the reads of the next group and
    // the splitting of the next k-step's activations sit BETWEEN the MFMAs, whose 32-cycle issue slots hide about five
    // single-issue instructions each.  The group's own fragments were requested a whole group step ago: lgkmcnt(0) up front costs
    // nothing, and no read of the next group is outstanding yet when it is taken.
    template <class Side>
    static __device__ __forceinline__ void run(WS &ws, const f4 (&h)[32], f16v (&acc)[SPLIT_VB], f4 (&a)[2][6], u4v (&b)[2][3], const Side &side) {
        constexpr int s = GI / 4, v0 = (GI % 4) * 2;
        static_assert(F % 6 == 0, "a group step reads six consecutive fragments of one chunk");
        lds_wait<0>(a[GI & 1]);
        const f4(&c)[6] = a[GI & 1];
        const u4v(&B)[3] = b[s & 1];
#define RTK_SPLIT_MM(pa, pb)                                                                          \
        acc[v0] = mfma_bf(__builtin_bit_cast(u4v, c[pa]), B[pb], acc[v0]);                            \
        acc[v0 + 1] = mfma_bf(__builtin_bit_cast(u4v, c[3 + pa]), B[pb], acc[v0 + 1]);
        RTK_SPLIT_MM(2, 0)
        if constexpr (GI + 1 < NG) load_part<GI + 1, 0>(ws, a[(GI + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        RTK_SPLIT_MM(0, 2)
        if constexpr (GI + 1 < NG) load_part<GI + 1, 1>(ws, a[(GI + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        RTK_SPLIT_MM(1, 1)
        if constexpr (GI + 1 < NG) load_part<GI + 1, 2>(ws, a[(GI + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (s + 1 < SPLIT_KS) split3_word<GI % 4>(h[2 * (s + 1)], h[2 * (s + 1) + 1], b[(s + 1) & 1]);
        RTK_SPLIT_MM(1, 0) RTK_SPLIT_MM(0, 1) RTK_SPLIT_MM(0, 0)
        side.template at<GI>(h);
#pragma unroll
        for (int k = 0; k < 6; ++k) {            // one MFMA, then up to three of the splitting's VALU instructions, six times
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
#undef RTK_SPLIT_MM
        __builtin_amdgcn_sched_barrier(0);
    }
};

template <int FBASE, class WS, int F, class Side, int... GI>
__device__ __forceinline__ void split_layer_impl(WS &ws, const f4 (&h)[32], f16v (&acc)[SPLIT_VB], const Side &side, std::integer_sequence<int, GI...>) {
    f4 a[2][6];
    u4v b[2][3];
    SplitStep<FBASE, 0, WS, F>::template load<0>(ws, a[0]);
    split3(h[0], h[1], b[0]);
    __builtin_amdgcn_sched_barrier(0);
    (SplitStep<FBASE, GI, WS, F>::run(ws, h, acc, a, b, side), ...);
}

// acc += W . h for a 256 x 256 layer whose split image starts at fragment FBASE of the stream.  All waves of the workgroup call
// this together (the stream has barriers).
template <int FBASE, int NW, int F, int NF, class Side = NoSide>
__device__ __forceinline__ void split_layer(WStreamA<NW, F, NF> &ws, const f4 (&h)[32], f16v (&acc)[SPLIT_VB], const Side &side = Side()) {
    split_layer_impl<FBASE, WStreamA<NW, F, NF>, F, Side>(ws, h, acc, side, std::make_integer_sequence<int, SPLIT_KS * (SPLIT_VB / 2)>{});
}

// Loads of kernel-lifetime constants (weights, biases) through the constant address space: the compiler may move them over the
// kernel's stores (a plain global load stays behind every store it might alias -- in the cost volume's epilogue that put one exposed
// L2 round trip in front of each of the eight output blocks) and turns the wave-uniform ones into scalar loads.
__device__ __forceinline__ float ldc(const float *p) { return *(const __attribute__((address_space(4))) float *)p; }
__device__ __forceinline__ f4 ldc4(const float *p) { return *(const __attribute__((address_space(4))) f4 *)p; }

// this lane's bias for output block v: channels 32 v + 8 q + 4 hh + r
__device__ __forceinline__ f16v split_bias(const float *__restrict__ bias, int v, int hh) {
    f16v o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f4 t = ldc4(bias + 32 * v + 8 * q + 4 * hh);
        o[4 * q] = t.x; o[4 * q + 1] = t.y; o[4 * q + 2] = t.z; o[4 * q + 3] = t.w;
    }
    return o;
}
