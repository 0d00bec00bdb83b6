// fused_common.h -- the "activation-stationary" fp32 MFMA micro-kernel every fused stage is built on.
//
// A wave owns a tile of 16 POSITIONS (points, or the 16 neighbours of one point).  Activations live
// in registers in the C/D layout of v_mfma_f32_16x16x4_f32 and never leave them between layers:
//
//     lane = 16*g + j   (g = 0..3, j = 0..15)          h[u][r] = H[channel 16u + 4g + r][position j]
//
// A layer Y = W.X is computed transposed, Y^T tiles = W (A operand) x H (B operand):
//     A[i][k=g] = W[16v + i][16u + 4g + r],   B[k=g][j] = h[u][r],   D -> acc[v][r'] = Y[16v + 4g + r'][j]
// The contraction index inside one MFMA is the lane group g, so the k-order is a fixed permutation
// of the channel order -- legal because the weights are packed on the host with the same permutation
// (pack_layer() in ratrack_amd/fused.py: packed[u][v][lane][r] = W[16v + (lane&15)][16u + 4(lane>>4) + r]).
// The D layout of layer L is exactly the B layout of layer L+1: bias + activation are applied in
// place and the next layer starts, with zero shuffles, zero LDS traffic for activations.
//
// fp32-input MFMA is bit-for-bit an fmaf chain (MI355X guide), so the only difference to the
// reference's conv2d is the summation ORDER; parity is within fp32 rounding (tests: 1e-4 rel).
#pragma once
#include <hip/hip_runtime.h>

#include <utility>

typedef float f4 __attribute__((ext_vector_type(4)));

#define RTK_ACT_NONE 0
#define RTK_ACT_RELU 1
#define RTK_ACT_LEAKY 2    // LeakyReLU(0.1)
#define RTK_ACT_SIGMOID 3

__device__ __forceinline__ f4 f4_zero() { return (f4){0.f, 0.f, 0.f, 0.f}; }

__device__ __forceinline__ f4 mfma4(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

__device__ __forceinline__ float act1(float x, int act) {
    switch (act) {
        case RTK_ACT_RELU: return fmaxf(x, 0.f);
        case RTK_ACT_LEAKY: return x > 0.f ? x : 0.1f * x;
        case RTK_ACT_SIGMOID: return 1.f / (1.f + __expf(-x));
        default: return x;
    }
}

__device__ __forceinline__ f4 act4(f4 x, int act) {
    return (f4){act1(x.x, act), act1(x.y, act), act1(x.z, act), act1(x.w, act)};
}

// Activation of a whole accumulator set.  `act` is wave-uniform (a kernel argument): branch once per
// layer instead of letting the compiler evaluate every variant per element and select.
template <int V>
__device__ __forceinline__ void apply_act(f4 (&acc)[V], int act) {
    if (act == RTK_ACT_RELU) {
#pragma unroll
        for (int v = 0; v < V; ++v) acc[v] = (f4){fmaxf(acc[v].x, 0.f), fmaxf(acc[v].y, 0.f), fmaxf(acc[v].z, 0.f), fmaxf(acc[v].w, 0.f)};
    } else if (act == RTK_ACT_LEAKY) {
#pragma unroll
        for (int v = 0; v < V; ++v)
            acc[v] = (f4){fmaxf(acc[v].x, 0.1f * acc[v].x), fmaxf(acc[v].y, 0.1f * acc[v].y), fmaxf(acc[v].z, 0.1f * acc[v].z),
                          fmaxf(acc[v].w, 0.1f * acc[v].w)};
    } else if (act == RTK_ACT_SIGMOID) {
#pragma unroll
        for (int v = 0; v < V; ++v) acc[v] = act4(acc[v], RTK_ACT_SIGMOID);
    }
}

// acc[v] += W[16v..16v+15][:] . h     W fragments read as one 16-byte load per (u, v) from `w`
// (global or LDS pointer to the packed image [U][V][64] f4).  V independent accumulators are
// interleaved in groups of 4 so that back-to-back MFMAs never depend on each other.
template <int U, int V>
__device__ __forceinline__ void mlp_layer(const f4 *__restrict__ w, int lane, const f4 (&h)[U], f4 (&acc)[V]) {
    constexpr int G = V >= 4 ? 4 : V;
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int v0 = 0; v0 < V; v0 += G) {
            f4 a[G];
#pragma unroll
            for (int q = 0; q < G; ++q)
                if (v0 + q < V) a[q] = w[(u * V + v0 + q) * 64 + lane];
#pragma unroll
            for (int q = 0; q < G; ++q) if (v0 + q < V) acc[v0 + q] = mfma4(a[q].x, h[u].x, acc[v0 + q]);
#pragma unroll
            for (int q = 0; q < G; ++q) if (v0 + q < V) acc[v0 + q] = mfma4(a[q].y, h[u].y, acc[v0 + q]);
#pragma unroll
            for (int q = 0; q < G; ++q) if (v0 + q < V) acc[v0 + q] = mfma4(a[q].z, h[u].z, acc[v0 + q]);
#pragma unroll
            for (int q = 0; q < G; ++q) if (v0 + q < V) acc[v0 + q] = mfma4(a[q].w, h[u].w, acc[v0 + q]);
        }
    }
}

// ---- workgroup -> (sample, slice) decode, XCD-aware ---------------------------------------------------------------------
// Consecutive workgroup ids are dispatched round-robin over the 8 XCDs (observed: block b runs on XCD b % 8) and each XCD
// has its own L2.  With a plain 2-D grid (x = slice, y = sample) the slices of one sample land on different XCDs and every
// one of those L2s fetches the sample's gathered rows again (measured on the cost volume: 2.9x the algorithmic bytes, on the
// patch aggregation 6.6x).  When samples % 8 == 0 the launchers use a 1-D grid of nbx * samples workgroups instead, decoded
// so that ALL workgroups of sample s run on XCD s % 8:  id -> (xcd = id % 8, slot = id / 8), s = 8 (slot / nbx) + xcd.
// gx <= 0 selects the plain 2-D grid.
__device__ __forceinline__ void rtk_decode_block(int gx, int &b, int &bx, int &nbx) {
    if (gx > 0) {
        const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
        b = (slot / gx) * 8 + xcd;
        bx = slot - (slot / gx) * gx;
        nbx = gx;
    } else {
        b = blockIdx.y;
        bx = blockIdx.x;
        nbx = gridDim.x;
    }
}

// ---- LDS weight stream ----------------------------------------------------------------------------
// The packed weights of a whole layer chain form one blob of NF fragments (1 fragment = one (u, v)
// A-operand image = 64 lanes x 16 B = 1 KiB).  The workgroup streams the blob cyclically through a
// double buffer of F fragments per half with global_load_lds (HBM/L2 -> LDS, no VGPR round trip):
// while the waves run MFMAs on chunk k, chunk k+1 lands in the other half.  One barrier per chunk.
// Fragment indices are compile-time constants (index_sequence expansion), so every `fi % F == 0`
// test and every bounds check folds away.
template <int NW, int F, int NF>
struct WStream {
    static constexpr int NCHUNKS = (NF + F - 1) / F;
    const char *blob;  // wave-uniform base of the packed weights
    f4 *lds;           // 2 * F * 64 f4
    unsigned lane_off; // byte offset of this lane's 16-byte slot inside fragment `wave`
    int cur, buf, wave, lane;

    // chunk index is wave-uniform; `into` selects the half of the double buffer.  The source address is
    // written as (uniform 64-bit base) + (32-bit per-lane offset) so that it selects the SGPR-base form of
    // global_load_lds and there is no per-call 64-bit VGPR address for the compiler to hoist and spill.
    __device__ __forceinline__ void issue(int chunk, int into) {
        const char *base = blob + (size_t)chunk * F * 1024;
#pragma unroll
        for (int i = 0; i < (F + NW - 1) / NW; ++i) {
            const int f = wave + i * NW;
            bool ok = f < F;
            if (NF % F != 0) ok = ok && (chunk * F + f < NF);   // only the last chunk of a ragged blob is partial
            if (ok) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + (size_t)i * NW * 1024 + lane_off),
                                                 (__attribute__((address_space(3))) void *)(lds + (into * F + f) * 64), 16, 0, 0);
            }
        }
    }
    __device__ __forceinline__ void start(const f4 *blob_, f4 *lds_, int wave_, int lane_) {
        lds = lds_; wave = wave_; lane = lane_;
        blob = reinterpret_cast<const char *>(blob_);
        lane_off = (unsigned)(wave * 64 + lane) * 16u;
        cur = 0; buf = 0;
        issue(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (NCHUNKS > 1) issue(1, 1);
    }
    // start() without its wait: chunk 0 is requested and the state is that of a stream whose previous pass has just ended (last
    // chunk current, chunk 0 on its way into the other half), so that the caller's first next() -- placed AFTER it has issued its own
    // first loads -- does the waiting: one round trip for the weights and the tile's inputs together instead of two in a row.
    __device__ __forceinline__ void start_deferred(const f4 *blob_, f4 *lds_, int wave_, int lane_) {
        static_assert(NCHUNKS > 1, "a blob of one chunk stays resident: start()");
        lds = lds_; wave = wave_; lane = lane_;
        blob = reinterpret_cast<const char *>(blob_);
        lane_off = (unsigned)(wave * 64 + lane) * 16u;
        cur = NCHUNKS - 1; buf = 1;
        issue(0, 0);
    }
    // the first next() of a tile after start_deferred() (a one-chunk blob was started with start() and stays resident)
    __device__ __forceinline__ void next_if_deferred() { if constexpr (NCHUNKS > 1) next(); }
    // move from the resident chunk to the next one (cyclic)
    __device__ __forceinline__ void next() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        buf ^= 1;
        cur = cur + 1 == NCHUNKS ? 0 : cur + 1;
        // `cur` is a compile-time constant at every call site; hide that, otherwise hipcc precomputes the
        // 64-bit source address of every load of every chunk outside the tile loop and spills them all.
        asm volatile("" : "+s"(cur));
        issue(cur + 1 == NCHUNKS ? 0 : cur + 1, buf ^ 1);
    }
    __device__ __forceinline__ f4 frag(int f_in_chunk) const {
        return lds[(buf * F + f_in_chunk) * 64 + lane];
    }
    __device__ __forceinline__ void finish() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
};

// One group step of a layer: read the fragments of group GI+1 from LDS (crossing into the next chunk
// of the stream if needed) while the 4*G MFMAs of group GI issue.  GI is a template constant so that
// h[u], acc[v] and the a[] ping-pong are all statically indexed (nothing can fall to scratch).
template <int U, int V, int FBASE, int GI, class WS, int F>
struct LayerStep {
    static constexpr int G = V >= 4 ? 4 : V;
    static constexpr int GV = (V + G - 1) / G;
    static constexpr int NG = U * GV;
    template <int GJ>
    static __device__ __forceinline__ void load(WS &ws, f4 (&dst)[G]) {
        constexpr int u = GJ / GV, v0 = (GJ % GV) * G;
#pragma unroll
        for (int q = 0; q < G; ++q) {
            if (v0 + q < V) {
                const int fi = FBASE + u * V + v0 + q;
                if (fi % F == 0 && fi != 0) ws.next();
                dst[q] = ws.frag(fi % F);
            }
        }
    }
    static __device__ __forceinline__ void run(WS &ws, const f4 (&h)[U], f4 (&acc)[V], f4 (&a)[2][G]) {
        constexpr int u = GI / GV, v0 = (GI % GV) * G;
        if constexpr (GI + 1 < NG) load<GI + 1>(ws, a[(GI + 1) & 1]);
        f4(&c)[G] = a[GI & 1];
#pragma unroll
        for (int q = 0; q < G; ++q) if (v0 + q < V) acc[v0 + q] = mfma4(c[q].x, h[u].x, acc[v0 + q]);
#pragma unroll
        for (int q = 0; q < G; ++q) if (v0 + q < V) acc[v0 + q] = mfma4(c[q].y, h[u].y, acc[v0 + q]);
#pragma unroll
        for (int q = 0; q < G; ++q) if (v0 + q < V) acc[v0 + q] = mfma4(c[q].z, h[u].z, acc[v0 + q]);
#pragma unroll
        for (int q = 0; q < G; ++q) if (v0 + q < V) acc[v0 + q] = mfma4(c[q].w, h[u].w, acc[v0 + q]);
        __builtin_amdgcn_sched_barrier(0);
    }
};

template <int U, int V, int FBASE, class WS, int F, int... GI>
__device__ __forceinline__ void mlp_layer_ws_impl(WS &ws, const f4 (&h)[U], f4 (&acc)[V], std::integer_sequence<int, GI...>) {
    constexpr int G = V >= 4 ? 4 : V;
    f4 a[2][G];
    LayerStep<U, V, FBASE, 0, WS, F>::template load<0>(ws, a[0]);
    (LayerStep<U, V, FBASE, GI, WS, F>::run(ws, h, acc, a), ...);
}

// acc[v] += W . h with the weights coming from the stream; FBASE = index of the layer's first fragment
// in the blob.  All waves of the workgroup must call this together (it contains barriers).
template <int U, int V, int FBASE, int NW, int F, int NF>
__device__ __forceinline__ void mlp_layer_ws(WStream<NW, F, NF> &ws, const f4 (&h)[U], f4 (&acc)[V]) {
    constexpr int G = V >= 4 ? 4 : V;
    constexpr int NG = U * ((V + G - 1) / G);
    mlp_layer_ws_impl<U, V, FBASE, WStream<NW, F, NF>, F>(ws, h, acc, std::make_integer_sequence<int, NG>{});
}


// ---- fp32 products on the 16-bit matrix pipe (the scheme is described in split_mfma.h) -----------------------------------------
// Shared pieces: the splits of eight activations (two fp16 pieces under a power-of-two scale: the layers; three exact bf16 pieces: the
// training kernels' contractions over positions), and the 16-position variant of the layer routine: the tile, register layout and
// weight stream of mlp_layer_ws, with v_mfma_f32_16x16x32_f16 (same C/D layout as the fp32-input 16x16x4) taking two 16-channel
// input blocks per k-step -- lane (g, j) supplies B[k = 8 g + t][j] = channel 16 (2 up + t / 4) + 4 g + t % 4, eight values it
// holds as h[2 up], h[2 up + 1].  Image: one 1 KiB fragment per (input block pair, output block, piece):
// frag[up][v][p][lane = 16 g + i][t] = piece_p(2^k W[16 v + i][16 (2 up + t / 4) + 4 g + t % 4]), p = 0: h, 1: l.
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_hi16(float lo, float hi) {      // the bf16 truncations of two floats in one register
    return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
__device__ __forceinline__ float trunc_bf16(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }

// the three bf16 pieces of eight activations (one k-step of this lane's B operand)
__device__ __forceinline__ void split3(const f4 x0, const f4 x1, u4v (&b)[3]) {
    float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w}, r1[8], r2[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        r1[t] = __fsub_rn(x[t], trunc_bf16(x[t]));          // exact
        r2[t] = __fsub_rn(r1[t], trunc_bf16(r1[t]));        // exact; at most 8 significant bits are left
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        b[0][w] = pack_hi16(x[2 * w], x[2 * w + 1]);
        b[1][w] = pack_hi16(r1[2 * w], r1[2 * w + 1]);
        b[2][w] = pack_hi16(r2[2 * w], r2[2 * w + 1]);
    }
}


// ---- round 5: TWO fp16 pieces, THREE products (the scheme is described in split_mfma.h) -----------------------------------------
// x s = h + l + e with h = fp16(x s), l = fp16(x s - h), both round-to-nearest: |e| <= 2^-23 |x s| while l is a normal fp16
// number and <= 2^-25 absolutely below that (subnormal pieces are KEPT by the f16 matrix instructions and by v_cvt_pk_f16_f32 on
// gfx950: tools/micro/f16_mfma_probe.hip).  s is an exact power of two chosen PER POSITION from the position's own largest
// activation, so that nothing overflows and the absolute floor sits 2^-39 below that largest value: the scheme has no range to
// leave.  Built from compiler-known instructions (v_pk_mul_f32, v_cvt_pk_f16_f32, v_fma_mixlo_f16 / v_fma_mixhi_f16 -- two VALU
// operations per value with -fno-slp-vectorize; the bf16 three-way split took 5.5).
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));

struct LaneScale {
    float s;        // 2^k: max |x s| over the position's channels lies in [2^14, 2^15)   (fp16 overflows at 65520)
    float inv;      // 2^-k
    unsigned mb;    // the bits of max |x| the scale was taken from
};
// from the bits of m = max |x| >= 0.  m = 0 or below 2^-112: k is clamped to 126 (everything scaled is then below 2^14 -- exact
// all the same); m = inf: the scaled operand overflows to inf and the product is NaN, as non-finite as the reference's.
__device__ __forceinline__ LaneScale lane_scale_of(unsigned mbits) {
    const unsigned e = mbits >> 23;
    unsigned sf = 268u - e;
    sf = sf > 253u ? 253u : sf;
    return LaneScale{__uint_as_float(sf << 23), __uint_as_float((254u - sf) << 23), mbits};
}
// The largest |element| of a whole TENSOR, for the consumers that contract over POSITIONS and therefore need ONE scale per tensor
// (train_gemm.hip): the producing kernels fold their lanes' maxima (LaneScale::mb) into a zero-initialised word -- one wave reduction
// and one atomic per tile and layer; an unsigned maximum on the bits of non-negative floats: order-independent, deterministic.
__device__ __forceinline__ void tensor_amax_update(float *amax, unsigned mb) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const unsigned o = (unsigned)__shfl_xor((int)mb, d, 64); mb = o > mb ? o : mb; }
    // (a look first: after the first tiles the word rarely grows, and tens of thousands of atomics on ONE address serialise in L2 -- a
    // stale look only costs an atomic that changes nothing)
    if ((threadIdx.x & 63) == 0 && mb > __atomic_load_n(reinterpret_cast<unsigned *>(amax), __ATOMIC_RELAXED))
        atomicMax(reinterpret_cast<unsigned *>(amax), mb);
}

// max |.| over a lane's activations, two per instruction (v_max3_f32 with |.| source modifiers; a NaN operand is ignored by the
// maximum and comes back through the product)
template <int N>
__device__ __forceinline__ float abs_max_f4(const f4 (&h)[N]) {
    float m = 0.f;
#pragma unroll
    for (int e = 0; e < N; ++e) {
        m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(h[e].x)), __builtin_fabsf(h[e].y));
        m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(h[e].z)), __builtin_fabsf(h[e].w));
    }
    return m;
}
// the two fp16 pieces of two scaled activations, one register each (low half = first value)
struct SplitWord { unsigned h, l; };
__device__ __forceinline__ SplitWord split2_word(float xa, float xb, float s) {
    const f2v xs = (f2v){xa, xb} * s;
    const h2v h = __builtin_convertvector(xs, h2v);
    const _Float16 la = (_Float16)__builtin_fmaf(xa, s, -(float)h[0]);      // x s - h is exact in fp32: ONE rounding, to fp16
    const _Float16 lb = (_Float16)__builtin_fmaf(xb, s, -(float)h[1]);
    return SplitWord{__builtin_bit_cast(unsigned, h), __builtin_bit_cast(unsigned, (h2v){la, lb})};
}
// ... of eight activations (one k-step of this lane's B operand): b[0] = h pieces, b[1] = l pieces
__device__ __forceinline__ void split2(const f4 x0, const f4 x1, float s, u4v (&b)[2]) {
    const SplitWord w0 = split2_word(x0.x, x0.y, s), w1 = split2_word(x0.z, x0.w, s), w2 = split2_word(x1.x, x1.y, s), w3 = split2_word(x1.z, x1.w, s);
    b[0] = (u4v){w0.h, w1.h, w2.h, w3.h};
    b[1] = (u4v){w0.l, w1.l, w2.l, w3.l};
}
__device__ __forceinline__ f4 mfma16_h(u4v a, u4v b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, a), __builtin_bit_cast(h8v, b), c, 0, 0, 0);
}

__device__ __forceinline__ f4 mfma16_bf(u4v a, u4v b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, a), __builtin_bit_cast(bf8v, b), c, 0, 0, 0);
}

constexpr int split16_nf(int U, int V) { return ((U + 1) / 2) * V * 2; }      // fragments of a layer's split image

// the activation scale of this lane's position in the 16-position tile: the four lanes j, 16 + j, 32 + j, 48 + j hold a quarter of
// its channels each.  v_permlane16_swap exchanges the odd rows of its first operand with the even rows of its second,
// v_permlane32_swap the upper half with the lower half: with both operands = m the two results hold a lane's and its partner's value.
template <int N>
__device__ __forceinline__ LaneScale lane_scale16(const f4 (&h)[N]) {
    unsigned mb = __float_as_uint(abs_max_f4(h));                // m >= 0: the order of the bit patterns is the order of the values
    auto r = __builtin_amdgcn_permlane16_swap(mb, mb, false, false);
    mb = r[0] > r[1] ? r[0] : r[1];
    r = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
    return lane_scale_of(r[0] > r[1] ? r[0] : r[1]);
}

// group step = one input block pair x two output blocks: four fragments (h and l piece of each block), six MFMAs (the two blocks
// interleaved, small terms first)
template <int U, int V, int FBASE, int GI, class WS, int F>
struct LayerStepS {
    static constexpr int GV = (V + 1) / 2;
    static constexpr int NG = ((U + 1) / 2) * GV;
    template <int GJ>
    static __device__ __forceinline__ void load(WS &ws, f4 (&dst)[4]) {
        constexpr int up = GJ / GV, v0 = (GJ % GV) * 2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (v0 + q / 2 < V) {
                const int fi = FBASE + (up * V + v0 + q / 2) * 2 + q % 2;
                if (fi % F == 0 && fi != 0) ws.next();
                dst[q] = ws.frag(fi % F);
            }
        }
    }
    static __device__ __forceinline__ void run(WS &ws, const f4 (&h)[U], float scale, f4 (&acc)[V], f4 (&a)[2][4], u4v (&b)[2]) {
        constexpr int up = GI / GV, v0 = (GI % GV) * 2;
        if constexpr (GI + 1 < NG) load<GI + 1>(ws, a[(GI + 1) & 1]);
        if constexpr (GI % GV == 0) split2(h[2 * up], 2 * up + 1 < U ? h[2 * up + 1] : f4_zero(), scale, b);
        const f4(&c)[4] = a[GI & 1];
#define RTK_S16_MM(pa, pb)                                                                            \
        acc[v0] = mfma16_h(__builtin_bit_cast(u4v, c[pa]), b[pb], acc[v0]);                            \
        if constexpr (v0 + 1 < V) acc[v0 + 1] = mfma16_h(__builtin_bit_cast(u4v, c[2 + pa]), b[pb], acc[v0 + 1]);
        RTK_S16_MM(1, 0) RTK_S16_MM(0, 1) RTK_S16_MM(0, 0)
#undef RTK_S16_MM
        __builtin_amdgcn_sched_barrier(0);
    }
};

template <int U, int V, int FBASE, class WS, int F, int... GI>
__device__ __forceinline__ void mlp_layer_split_impl(WS &ws, const f4 (&h)[U], float scale, f4 (&acc)[V], std::integer_sequence<int, GI...>) {
    f4 a[2][4];
    u4v b[2];
    LayerStepS<U, V, FBASE, 0, WS, F>::template load<0>(ws, a[0]);
    (LayerStepS<U, V, FBASE, GI, WS, F>::run(ws, h, scale, acc, a, b), ...);
}

// acc[v] += 2^(kw + kx) W . h on the split path (scale = this lane's 2^kx, lane_scale16); FBASE = index of the layer's first fragment
// in the (split) blob
template <int U, int V, int FBASE, int NW, int F, int NF>
__device__ __forceinline__ void mlp_layer_ws_split(WStream<NW, F, NF> &ws, const f4 (&h)[U], float scale, f4 (&acc)[V]) {
    constexpr int NG = ((U + 1) / 2) * ((V + 1) / 2);
    mlp_layer_split_impl<U, V, FBASE, WStream<NW, F, NF>, F>(ws, h, scale, acc, std::make_integer_sequence<int, NG>{});
}

// Weights resident in LDS (image [U][V][64] f4 at `w`): same pipeline without the stream.
struct WResident {
    const f4 *w;
    int lane;
    __device__ __forceinline__ void next() {}
    __device__ __forceinline__ f4 frag(int f) const { return w[f * 64 + lane]; }
};

template <int U, int V>
__device__ __forceinline__ void mlp_layer_res(const f4 *w, int lane, const f4 (&h)[U], f4 (&acc)[V]) {
    constexpr int G = V >= 4 ? 4 : V;
    constexpr int NG = U * ((V + G - 1) / G);
    WResident wr{w, lane};
    mlp_layer_ws_impl<U, V, 0, WResident, (1 << 30)>(wr, h, acc, std::make_integer_sequence<int, NG>{});
}

// bias fragment of this lane for output block v: channels 16v + 4g .. +3
__device__ __forceinline__ f4 bias_frag(const float *__restrict__ bias, int v, int g) {
    return *reinterpret_cast<const f4 *>(bias + 16 * v + 4 * g);
}

// ---- reductions over the 16 positions of a tile (the 16 lanes of a DPP row) ---------------------
// One instruction per step: v_max_f32 / v_add_f32 with a DPP-permuted first operand (row_ror rotates within a
// 16-lane row, so 4 steps leave the full-row result in every lane).  Written as inline asm on a whole f4:
// hipcc otherwise emits v_mov_b32_dpp + s_nop + canonicalising v_max per step (5x the instructions), which made
// the epilogue, not the MFMAs, the cost of the small set-abstraction scales.  Hazards: (1) a VGPR written by VALU needs
// 2 wait states before a DPP read -- the leading s_nop 1 covers the producer of v, and inside the block each register is
// re-read only after the 3 other components were issued; (2) a VGPR written by an MFMA needs up to 19 wait states before any
// VALU read, and the compiler's hazard recognizer does not look inside inline asm: an accumulator that reached the asm block
// directly was read too early (round 2: wrong maxima under a register cap).  rtk_dpp_fence makes every helper self-contained:
// each component first goes through a compiler-VISIBLE VALU instruction that cannot be folded away -- a bitwise OR with a zero
// the optimiser cannot see (an SGPR written by a volatile asm) --, so the recognizer inserts exactly the MFMA wait states the
// producer needs (none when the producer is an ordinary VALU op) in front of it; the block's own s_nop 1 then covers VALU -> DPP.
// (Beware: __builtin_bit_cast(int, v.x) on an ext_vector_type ELEMENT reads element 0 whatever the element named -- hipcc 7.2;
// __float_as_int on the element's value is fine.)
__device__ __forceinline__ void rtk_dpp_fence(f4 &v) {
    int zero;
    asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
    v.x = __int_as_float(__float_as_int(v.x) | zero);
    v.y = __int_as_float(__float_as_int(v.y) | zero);
    v.z = __int_as_float(__float_as_int(v.z) | zero);
    v.w = __int_as_float(__float_as_int(v.w) | zero);
}
#define RTK_DPP4(op, ctrl)                                                 \
    op " %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf\n"                 \
    op " %1, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf\n"                 \
    op " %2, %2, %2 " ctrl " row_mask:0xf bank_mask:0xf\n"                 \
    op " %3, %3, %3 " ctrl " row_mask:0xf bank_mask:0xf\n"

__device__ __forceinline__ void row_max16_f4(f4 &v) {
    rtk_dpp_fence(v);
    asm volatile("s_nop 1\n" RTK_DPP4("v_max_f32_dpp", "row_ror:8") RTK_DPP4("v_max_f32_dpp", "row_ror:4")
                 RTK_DPP4("v_max_f32_dpp", "row_ror:2") RTK_DPP4("v_max_f32_dpp", "row_ror:1")
                 : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
}

__device__ __forceinline__ void row_sum16_f4(f4 &v) {
    rtk_dpp_fence(v);
    asm volatile("s_nop 1\n" RTK_DPP4("v_add_f32_dpp", "row_ror:8") RTK_DPP4("v_add_f32_dpp", "row_ror:4")
                 RTK_DPP4("v_add_f32_dpp", "row_ror:2") RTK_DPP4("v_add_f32_dpp", "row_ror:1")
                 : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
}

// ... for an argument every component of which is the result of a VALU instruction the compiler sees (a product, a sum): it has
// already placed the wait states a matrix-core producer further up needs, and the block's own s_nop 1 covers VALU -> DPP; the
// fence's four ORs are saved (the cost volume's epilogue calls this 32 times per tile).
__device__ __forceinline__ void row_sum16_valu_f4(f4 &v) {
    asm volatile("s_nop 1\n" RTK_DPP4("v_add_f32_dpp", "row_ror:8") RTK_DPP4("v_add_f32_dpp", "row_ror:4")
                 RTK_DPP4("v_add_f32_dpp", "row_ror:2") RTK_DPP4("v_add_f32_dpp", "row_ror:1")
                 : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
}

// ... four of them at once, step by step over all sixteen registers: a rotation step's result is the next step's operand, and with
// one f4 at a time the dependent DPP instructions are four apart -- closer than the cross-lane path's latency.
#define RTK_DPP16(op, ctrl)                                                                                                          \
    op " %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf\n" op " %1, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf\n"                   \
    op " %2, %2, %2 " ctrl " row_mask:0xf bank_mask:0xf\n" op " %3, %3, %3 " ctrl " row_mask:0xf bank_mask:0xf\n"                   \
    op " %4, %4, %4 " ctrl " row_mask:0xf bank_mask:0xf\n" op " %5, %5, %5 " ctrl " row_mask:0xf bank_mask:0xf\n"                   \
    op " %6, %6, %6 " ctrl " row_mask:0xf bank_mask:0xf\n" op " %7, %7, %7 " ctrl " row_mask:0xf bank_mask:0xf\n"                   \
    op " %8, %8, %8 " ctrl " row_mask:0xf bank_mask:0xf\n" op " %9, %9, %9 " ctrl " row_mask:0xf bank_mask:0xf\n"                   \
    op " %10, %10, %10 " ctrl " row_mask:0xf bank_mask:0xf\n" op " %11, %11, %11 " ctrl " row_mask:0xf bank_mask:0xf\n"             \
    op " %12, %12, %12 " ctrl " row_mask:0xf bank_mask:0xf\n" op " %13, %13, %13 " ctrl " row_mask:0xf bank_mask:0xf\n"             \
    op " %14, %14, %14 " ctrl " row_mask:0xf bank_mask:0xf\n" op " %15, %15, %15 " ctrl " row_mask:0xf bank_mask:0xf\n"
__device__ __forceinline__ void row_sum16_valu_f4x4(f4 (&v)[4]) {      // arguments: VALU results the compiler sees (row_sum16_valu_f4)
    asm volatile("s_nop 1\n" RTK_DPP16("v_add_f32_dpp", "row_ror:8") RTK_DPP16("v_add_f32_dpp", "row_ror:4")
                 RTK_DPP16("v_add_f32_dpp", "row_ror:2") RTK_DPP16("v_add_f32_dpp", "row_ror:1")
                 : "+v"(v[0].x), "+v"(v[0].y), "+v"(v[0].z), "+v"(v[0].w), "+v"(v[1].x), "+v"(v[1].y), "+v"(v[1].z), "+v"(v[1].w),
                   "+v"(v[2].x), "+v"(v[2].y), "+v"(v[2].z), "+v"(v[2].w), "+v"(v[3].x), "+v"(v[3].y), "+v"(v[3].z), "+v"(v[3].w));
}

// ---- transposing reductions (round 5) ------------------------------------------------------------------------------------------
// The reductions above leave the full-row result of EVERY register in EVERY lane: log2(16) steps x N registers.  When N registers are
// reduced at once, half of that is waste: a step that combines lanes l and l ^ k can at the same time HALVE the registers -- the lanes
// with bit k clear keep register a's partial result, the lanes with bit k set keep register b's, in ONE register.  The DPP write masks
// do this for free for k = 8 and k = 4 (bank_mask enables the four 4-lane banks of a row separately: a masked lane keeps the old
// destination), v_permlane16_swap for k = 16 (row 1 of a <-> row 0 of b, row 3 <-> row 2: then one max / add); the last two steps
// (k = 2, 1, inside a quad) run on the N / 4 (N / 8) registers that are left.  16 registers over a row: 16 + 8 + 4 + 4 = 32 cross-lane
// operations instead of 64.  Afterwards register j of lane l holds the result of input register (j, bits 3 and 2 of l): pair8 keeps
// `a` in lanes 0-7 and `b` in lanes 8-15 of each row, pair4 keeps `a` where bit 2 of the lane is clear.
// Arguments: VALU results the compiler sees (see row_sum16_valu_f4); each block's s_nop covers VALU -> DPP.
#define RTK_DPP_PAIR(op, c0, m0, c1, m1)                                                                                     \
    asm volatile("s_nop 1\n"                                                                                                 \
                 op " %0, %0, %0 " c0 " row_mask:0xf bank_mask:" m0 "\n" op " %1, %1, %1 " c0 " row_mask:0xf bank_mask:" m0 "\n"  \
                 op " %2, %2, %2 " c0 " row_mask:0xf bank_mask:" m0 "\n" op " %3, %3, %3 " c0 " row_mask:0xf bank_mask:" m0 "\n"  \
                 op " %0, %4, %4 " c1 " row_mask:0xf bank_mask:" m1 "\n" op " %1, %5, %5 " c1 " row_mask:0xf bank_mask:" m1 "\n"  \
                 op " %2, %6, %6 " c1 " row_mask:0xf bank_mask:" m1 "\n" op " %3, %7, %7 " c1 " row_mask:0xf bank_mask:" m1 "\n"  \
                 : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w) : "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w))
__device__ __forceinline__ void row_max_pair8(f4 &a, const f4 b) { RTK_DPP_PAIR("v_max_f32_dpp", "row_ror:8", "0x3", "row_ror:8", "0xc"); }
__device__ __forceinline__ void row_max_pair4(f4 &a, const f4 b) { RTK_DPP_PAIR("v_max_f32_dpp", "row_shl:4", "0x5", "row_shr:4", "0xa"); }
__device__ __forceinline__ void row_sum_pair8(f4 &a, const f4 b) { RTK_DPP_PAIR("v_add_f32_dpp", "row_ror:8", "0x3", "row_ror:8", "0xc"); }
__device__ __forceinline__ void row_sum_pair4(f4 &a, const f4 b) { RTK_DPP_PAIR("v_add_f32_dpp", "row_shl:4", "0x5", "row_shr:4", "0xa"); }
#undef RTK_DPP_PAIR
// the last two steps, inside each quad (every lane of a quad ends with the quad's result)
__device__ __forceinline__ void quad_max_valu_f4(f4 &v) {
    asm volatile("s_nop 1\n" RTK_DPP4("v_max_f32_dpp", "quad_perm:[1,0,3,2]") RTK_DPP4("v_max_f32_dpp", "quad_perm:[2,3,0,1]")
                 : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
}
__device__ __forceinline__ void quad_sum_valu_f4(f4 &v) {
    asm volatile("s_nop 1\n" RTK_DPP4("v_add_f32_dpp", "quad_perm:[1,0,3,2]") RTK_DPP4("v_add_f32_dpp", "quad_perm:[2,3,0,1]")
                 : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
}
// rows 0 | 1 and 2 | 3 of the wave (lanes l and l ^ 16): a's combination in the even rows, b's in the odd rows
// (inf: +infinity the optimiser cannot see -- max(x, y) = med3(x, y, inf) is ONE instruction, fmaxf under IEEE mode up to three)
__device__ __forceinline__ f4 wave_max_pair16(const f4 a, const f4 b, float inf) {
    f4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const auto t = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[r]), __float_as_uint(b[r]), false, false);
        o[r] = __builtin_amdgcn_fmed3f(__uint_as_float(t[0]), __uint_as_float(t[1]), inf);
    }
    return o;
}
// Four f4 registers (index q) -> one: the maximum over the 16 lanes of each row; lane l ends with q = 2 (bit 3 of l) + (bit 2 of l)
__device__ __forceinline__ f4 row_max16_transpose4(f4 q0, f4 q1, const f4 q2, const f4 q3) {
    row_max_pair8(q0, q2);
    row_max_pair8(q1, q3);
    row_max_pair4(q0, q1);
    quad_max_valu_f4(q0);
    return q0;
}
__device__ __forceinline__ f4 row_sum16_transpose4(f4 q0, f4 q1, const f4 q2, const f4 q3) {
    row_sum_pair8(q0, q2);
    row_sum_pair8(q1, q3);
    row_sum_pair4(q0, q1);
    quad_sum_valu_f4(q0);
    return q0;
}

// max over aligned sub-groups of GROUP (4, 8 or 16) lanes within the row
template <int GROUP>
__device__ __forceinline__ void row_max_group_f4(f4 &v) {
    if constexpr (GROUP >= 16) {
        row_max16_f4(v);
    } else if constexpr (GROUP == 8) {
        rtk_dpp_fence(v);
        // quad xor-1, quad xor-2, then row_half_mirror (lane i <-> 7-i inside each 8-lane half reaches the other quad)
        asm volatile("s_nop 1\n" RTK_DPP4("v_max_f32_dpp", "quad_perm:[1,0,3,2]") RTK_DPP4("v_max_f32_dpp", "quad_perm:[2,3,0,1]")
                     RTK_DPP4("v_max_f32_dpp", "row_half_mirror")
                     : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
    } else {
        rtk_dpp_fence(v);
        asm volatile("s_nop 1\n" RTK_DPP4("v_max_f32_dpp", "quad_perm:[1,0,3,2]") RTK_DPP4("v_max_f32_dpp", "quad_perm:[2,3,0,1]")
                     : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
    }
}
