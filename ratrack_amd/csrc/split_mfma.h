// split_mfma.h -- fp32 products on the fp16 matrix pipe (round 5; rounds 2-4 took three bf16 pieces and six products).
//
// Every fp32 operand, scaled by an exact power of two, is the sum of TWO fp16 pieces up to 2^-23 of itself:
//     x s = h + l + e,   h = fp16(x s),   l = fp16(x s - h)   (round to nearest; x s - h is exact in fp32)
// A fp16 x fp16 product is exact in fp32 (11 + 11 significant bits), so a product a.b is taken as the THREE partial products
// a_l.b_h + a_h.b_l + a_h.b_h accumulated in fp32 by v_mfma_f32_32x32x16_f16 (same rate as the bf16 instruction); the dropped
// a_l.b_l term is below 2^-22 |a||b|.  Measured against float64 on 256-deep products (tools/experiments/exp_split_f16.py, and on
// the device tests/test_fused_gpu.py::test_split_layers_carry_fp32_accuracy): 2.7e-7 of max|y| -- an fp32 GEMM has 6.4e-7, the six
// bf16 products had 3.9e-7, three bf16 products 2.8e-5.  Half the matrix instructions, 2 instead of 5.5 VALU operations per split
// value and 4 instead of 6 bytes per streamed weight for the same accuracy.
//
// Range.  fp16 has five exponent bits, so the SCALES are what makes this an fp32 path and not an fp16 one:
//   * weights: one exact power of two per matrix, chosen by the packer so that max|W| 2^k lies in [2^14, 2^15) (rtk_pack_split_layer
//     returns 2^-k next to the image);
//   * activations: one exact power of two per POSITION (per column of the B operand), computed by the kernel from the position's
//     own largest activation (lane_scale32 below: 64 v_max3_f32 + one v_permlane32_swap per layer and lane) -- whatever the
//     magnitude of a layer's input, its largest element sits at 2^14..2^15 and an element 2^-16 below it still has all 23 bits.
//     Nothing can overflow, there is no subnormal cliff to fall off (smaller elements keep an ABSOLUTE precision of 2^-39 of the
//     position's maximum: below one fp32 rounding of the dot product they enter), and there is no range flag because there is no
//     range to leave: tests/test_fused_gpu.py::test_split_layers_are_scale_invariant runs the layers at 2^-60 .. 2^60.
//   The accumulator holds y 2^(kw + kx); the epilogue multiplies by the lane's 2^-(kw + kx) inside the bias fma -- exact.
//
// Why a matrix-pipe detour at all: gfx950 has no xf32 and its fp32-input MFMA runs at the VECTOR rate (157 TFLOP/s, 1/16 of
// fp16 / bf16).  Three fp16 MFMAs do the work of sixteen fp32 ones for the same bits.
//
// Tile: a wave owns 32 POSITIONS (two query points x 16 neighbours).  Activations stay in registers in the C/D layout of the
// 32x32 MFMA, which again is a legal B layout for the next layer:
//
//     lane = 32 hh + col   (hh = 0..1, col = 0..31)        H[ch = 32 v + 8 q + 4 hh + r][position col] = h[4 v + q][r]
//
// A k-step (16 input channels) takes q in {q0, q0 + 1} of one 32-channel block: lane supplies B[k = 8 hh + t][col],
// t = 4 (q - q0) + r -- eight values it already holds.  The weights are packed with the same permutation (rtk_pack_split_layer
// on the device; pack_layer_split() in ratrack_amd/fused.py is its host restatement), one 1 KiB fragment (64 lanes x 8 fp16) per
// (k-step, 32-row block, piece):
//
//     frag[s][v][p][lane = 32 hh + i][t] = piece_p( 2^k W[32 v + i][32 (s / 2) + 16 (s % 2) + 8 (t / 4) + 4 hh + t % 4] ),  p = 0: h, 1: l
#pragma once
#include "fused_common.h"

typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f16v mfma_h(u4v a, u4v b, f16v c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8v, a), __builtin_bit_cast(h8v, b), c, 0, 0, 0);
}

// (the three-piece bf16 form of rounds 2-4: the training kernels' contractions over POSITIONS still take it -- train_gemm.hip)
__device__ __forceinline__ f16v mfma_bf(u4v a, u4v b, f16v c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8v, a), __builtin_bit_cast(bf8v, b), c, 0, 0, 0);
}

// The activation scale of this lane's position: the two lanes of a position (hh = 0, 1) hold half of its channels each.
// v_permlane32_swap exchanges the upper half of its first operand with the lower half of its second: with both = m, lane l and lane
// l + 32 between them see (m[l], m[l + 32]) in the two results.
__device__ __forceinline__ LaneScale lane_scale32_of(float m) {
    const unsigned mb = __float_as_uint(m);
    const auto r = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
    return lane_scale_of(r[0] > r[1] ? r[0] : r[1]);      // m >= 0: the order of the bit patterns is the order of the values
}
template <int N>
__device__ __forceinline__ LaneScale lane_scale32(const f4 (&h)[N]) { return lane_scale32_of(abs_max_f4(h)); }

// ... one register (two activations) of each piece at a time: the k-step's splitting is spread over its four group steps
template <int W>
__device__ __forceinline__ void split2_word_of(const f4 x0, const f4 x1, float s, u4v (&b)[2]) {
    const float xa = W < 2 ? x0[2 * W] : x1[2 * W - 4], xb = W < 2 ? x0[2 * W + 1] : x1[2 * W - 3];
    const SplitWord w = split2_word(xa, xb, s);
    b[0][W] = w.h; b[1][W] = w.l;
}

// A chunk of F fragments is F / 4 group steps; the requests for the chunk after it go out during the first split_issue_groups(F)
// of them, four per wave and group step (the last ones must be old enough at the chunk's closing vmcnt(0) to have made their L2
// round trip): two of eight (F = 32), four of sixteen (F = 64).
constexpr int SPLIT_GF = 4;       // fragments of a group step: two 32-row output blocks x two pieces
constexpr int split_issue_groups(int F) { return F >= 16 ? F / 16 : 1; }
constexpr int SPLIT_KS = 16;      // k-steps of a 256-channel contraction
constexpr int SPLIT_VB = 8;       // 32-row output blocks of a 256-channel layer
constexpr int SPLIT_NF = SPLIT_KS * SPLIT_VB * 2;      // fragments (KiB) of one 256 x 256 layer

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
__device__ __forceinline__ void lds_wait(f4 (&c)[SPLIT_GF]) {      // the four fragments are outputs: their uses cannot move above the wait
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]) : "n"(OUTSTANDING));
}

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
// Loads of kernel-lifetime constants (weights, biases) through the constant address space: the compiler may move them over the
// kernel's stores (a plain global load stays behind every store it might alias -- in the cost volume's epilogue that put one exposed
// L2 round trip in front of each of the eight output blocks) and turns the wave-uniform ones into scalar loads.
__device__ __forceinline__ float ldc(const float *p) { return *(const __attribute__((address_space(4))) float *)p; }
__device__ __forceinline__ f4 ldc4(const float *p) { return *(const __attribute__((address_space(4))) f4 *)p; }

// The layer's bias in the tile layout (slot e = 4 v + q: channels 32 v + 8 q + 4 hh + r) arrives in four QUARTERS of eight slots, each
// requested while the quarter before it is being worked on (1 KiB per layer, the same for every tile: first-level cache hits); the
// first one during the layer's last k-step (BiasSide, two slots per group step).  (All 32 slots at once are 128 registers next to
// the 128 accumulators and their copies on the way to the vector pipe: the kernel then needs all 512 and spills.)
__device__ __forceinline__ f4 bias_slot(const float *bias4hh, int e) {      // (plain loads: through the constant address space they are
    return *reinterpret_cast<const f4 *>(bias4hh + 32 * (e >> 2) + 8 * (e & 3));      // hoisted out of the tile loop -- 128 registers for good)
}
struct BiasSide {
    const float *bias;            // + 4 hh
    f4 (&bq)[2][8];
    template <int GI>
    __device__ __forceinline__ void at(const f4 (&)[32]) const {
        constexpr int NG = SPLIT_KS * (SPLIT_VB / 2);
        if constexpr (GI >= NG - 4) {
            bq[0][2 * (GI - (NG - 4))] = bias_slot(bias, 2 * (GI - (NG - 4)));
            bq[0][2 * (GI - (NG - 4)) + 1] = bias_slot(bias, 2 * (GI - (NG - 4)) + 1);
        }
    }
};
// emit(e, acc slot e, bias slot e) for e = 0..31, quarter by quarter
template <class Emit>
__device__ __forceinline__ void split_epilogue(const f16v (&acc)[SPLIT_VB], const float *bias4hh, f4 (&bq)[2][8], Emit &&emit) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < 3) {
#pragma unroll
            for (int e = 0; e < 8; ++e) bq[(k + 1) & 1][e] = bias_slot(bias4hh, 8 * (k + 1) + e);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int v = (8 * k + e) >> 2, q = e & 3;
            emit(8 * k + e, (f4){acc[v][4 * q], acc[v][4 * q + 1], acc[v][4 * q + 2], acc[v][4 * q + 3]}, bq[k & 1][e]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <class A, class B>
struct SidePair {
    A a;
    B b;
    template <int GI>
    __device__ __forceinline__ void at(const f4 (&h)[32]) const { a.template at<GI>(h); b.template at<GI>(h); }
};

// One group step: the four fragments of two output blocks (two pieces each) against the two B pieces of the k-step: six MFMAs,
// the two blocks interleaved so that consecutive MFMAs never wait for each other's accumulator; small terms first.  The fragments
// of the next group (and, once per k-step, the B pieces of the next k-step) are fetched / split meanwhile.
template <int FBASE, int GI, class WS, int F, bool ZERO>
struct SplitStep {
    static constexpr int NG = SPLIT_KS * (SPLIT_VB / 2);
    // the four reads of group GJ, two at a time (PART = 0, 1), so that they can be placed between the MFMAs of the group before
    template <int GJ, int PART>
    static __device__ __forceinline__ void load_part(WS &ws, f4 (&dst)[SPLIT_GF]) {
        constexpr int f0 = FBASE + GJ * SPLIT_GF;      // F % 4 == 0: a group never straddles two chunks
        if constexpr (PART == 0) {
            if constexpr (f0 % F == 0 && f0 != 0) ws.sync();
            if constexpr ((f0 % F) / SPLIT_GF < split_issue_groups(F)) ws.template issue_part<(f0 % F) / SPLIT_GF, split_issue_groups(F)>();
        }
        dst[2 * PART] = ws.template frag_async<(f0 + 2 * PART) % F>();
        dst[2 * PART + 1] = ws.template frag_async<(f0 + 2 * PART + 1) % F>();
    }
    template <int GJ>
    static __device__ __forceinline__ void load(WS &ws, f4 (&dst)[SPLIT_GF]) {
        load_part<GJ, 0>(ws, dst);
        load_part<GJ, 1>(ws, dst);
    }
    // Schedule of a group step, pinned: hipcc left alone issues [reads, wait, all VALU, the MFMAs back to back] -- the matrix pipe
    // idles while the other instructions issue (one wave per SIMD: nobody else feeds it).  Here the reads of the next group and
    // the splitting of the next k-step's activations sit BETWEEN the MFMAs, whose 32-cycle issue slots hide about five
    // single-issue instructions each.  The group's own fragments were requested a whole group step ago: lgkmcnt(0) up front costs
    // nothing, and no read of the next group is outstanding yet when it is taken.
    template <class Side>
    static __device__ __forceinline__ void run(WS &ws, const f4 (&h)[32], float scale, f16v (&acc)[SPLIT_VB], f4 (&a)[2][SPLIT_GF], u4v (&b)[2][2],
                                               const Side &side) {
        constexpr int s = GI / 4, v0 = (GI % 4) * 2;
        static_assert(F % SPLIT_GF == 0, "a group step reads four consecutive fragments of one chunk");
        lds_wait<0>(a[GI & 1]);
        const f4(&c)[SPLIT_GF] = a[GI & 1];
        const u4v(&B)[2] = b[s & 1];
        // (the first k-step of a ZERO layer starts its accumulators from the instruction's inline zero: no 128 register writes)
        f16v z0, z1;
        if constexpr (ZERO && s == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) z0[e] = z1[e] = 0.f;
        } else {
            z0 = acc[v0]; z1 = acc[v0 + 1];
        }
        acc[v0] = mfma_h(__builtin_bit_cast(u4v, c[1]), B[0], z0);                    // l . h
        acc[v0 + 1] = mfma_h(__builtin_bit_cast(u4v, c[3]), B[0], z1);
        if constexpr (GI + 1 < NG) load_part<GI + 1, 0>(ws, a[(GI + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        acc[v0] = mfma_h(__builtin_bit_cast(u4v, c[0]), B[1], acc[v0]);               // h . l
        acc[v0 + 1] = mfma_h(__builtin_bit_cast(u4v, c[2]), B[1], acc[v0 + 1]);
        if constexpr (GI + 1 < NG) load_part<GI + 1, 1>(ws, a[(GI + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (s + 1 < SPLIT_KS) split2_word_of<GI % 4>(h[2 * (s + 1)], h[2 * (s + 1) + 1], scale, b[(s + 1) & 1]);
        acc[v0] = mfma_h(__builtin_bit_cast(u4v, c[0]), B[0], acc[v0]);               // h . h
        acc[v0 + 1] = mfma_h(__builtin_bit_cast(u4v, c[2]), B[0], acc[v0 + 1]);
        side.template at<GI>(h);
#pragma unroll
        for (int k = 0; k < 2; ++k) {            // one MFMA, then up to three of the splitting's VALU instructions, twice
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
};

template <int FBASE, class WS, int F, bool ZERO, class Side, int... GI>
__device__ __forceinline__ void split_layer_impl(WS &ws, const f4 (&h)[32], float scale, f16v (&acc)[SPLIT_VB], const Side &side,
                                                 std::integer_sequence<int, GI...>) {
    f4 a[2][SPLIT_GF];
    u4v b[2][2];
    SplitStep<FBASE, 0, WS, F, ZERO>::template load<0>(ws, a[0]);
    split2(h[0], h[1], scale, b[0]);
    __builtin_amdgcn_sched_barrier(0);
    (SplitStep<FBASE, GI, WS, F, ZERO>::run(ws, h, scale, acc, a, b, side), ...);
}

// acc = 2^(kw + kx) W . h for a 256 x 256 layer whose split image starts at fragment FBASE of the stream (acc += with ZERO = false);
// scale = this lane's 2^kx (lane_scale32).  All waves of the workgroup call this together (the stream has barriers).
template <int FBASE, bool ZERO = true, int NW, int F, int NF, class Side = NoSide>
__device__ __forceinline__ void split_layer(WStreamA<NW, F, NF> &ws, const f4 (&h)[32], float scale, f16v (&acc)[SPLIT_VB], const Side &side = Side()) {
    split_layer_impl<FBASE, WStreamA<NW, F, NF>, F, ZERO, Side>(ws, h, scale, acc, side, std::make_integer_sequence<int, SPLIT_KS * (SPLIT_VB / 2)>{});
}

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
// slot e = 4 v + q of an accumulator set as an f4
__device__ __forceinline__ f4 acc_slot(const f16v (&acc)[SPLIT_VB], int e) {
    const int v = e >> 2, q = e & 3;
    return (f4){acc[v][4 * q], acc[v][4 * q + 1], acc[v][4 * q + 2], acc[v][4 * q + 3]};
}
