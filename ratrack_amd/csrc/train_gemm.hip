// train_gemm.hip -- weight gradients that are contractions over POSITIONS, on the split matrix path (split_mfma.h; round 5: two fp16
// pieces per operand and three products, rounds 3-4: three bf16 pieces and six):
//
//     out[i][j] = sum_{r < m} x[r][i] * y[r][j]        x, y (m, 256) fp32 row-major, m in the hundred thousands
//
// (the cost volume's dW2 = dz2^T a1 and dW3 = dz3^T a2: utils/model_utils/model_utils.py:177-183,226-231 in the backward; 34 GFLOP
// each at B = 64).  The fp32-input MFMA runs at the vector rate (157 TFLOP/s) and a library GEMM sat at 0.89 of that; here every
// fp32 product is three fp16 MFMA products of two operand pieces each, fp32 accumulate -- the error of an fp32 fmaf chain.
// Scales: the contraction index is the ROW, so an operand can only take ONE power of two for the whole tensor (a per-position scale,
// what the layers use, would have to multiply the accumulators): 2^k with max|x| 2^k in [2^14, 2^15), from the tensor's largest
// |element| which the producing kernels hand over (rtk_cost_volume_split_train / rtk_cost_volume_bwd_split fold it into a word per
// tensor; rtk_absmax for anything else).  A row far below the tensor's largest keeps an ABSOLUTE precision of 2^-39 of that largest --
// relative to the sums it enters (dominated by the large rows) below one fp32 rounding.
//
// Both operands are contiguous along the CHANNEL, the contraction runs along the row index: the MFMA wants, per lane, eight
// consecutive k (= rows) of one channel.  The transposition happens on the way into LDS and costs nothing extra because the operands
// have to be split into their bf16 pieces there anyway:
//
//   * a workgroup (4 waves, one per SIMD) owns a SLAB of rows and the full 256 x 256 output: every element of x and y is read from
//     HBM exactly once by the whole grid;
//   * a k-step is 16 rows.  Thread (wave kq, lane cq) loads rows 4 kq .. 4 kq + 3 of channels 4 cq .. 4 cq + 3 as four float4 (a wave
//     reads whole 1 KiB rows), and holds -- per channel -- four consecutive k: split into three pieces, packed, one 8-byte LDS write
//     per (channel, piece).  LDS stage: [operand][piece][k half][slot][8 bf16], slot = 64 (channel % 4) + channel / 4: lanes write
//     consecutive slots, and a lane's MFMA operand (row `slot`, k half) is ONE conflict-free 16-byte read.  The output comes out in
//     slot order and is un-permuted by the reduction;
//   * wave (wa, wb) accumulates the 128 x 128 block of slots in 256 accumulator registers: per k-step 24 fragment reads feed 96
//     MFMAs; the global loads of step t + 2 are in flight while step t multiplies and step t + 1 is split and written to the other
//     LDS stage (one barrier per step);
//   * slabs are reduced by a second kernel (deterministic; no float atomics).
#include "rtk_common.h"
#include "split_mfma.h"

namespace {

constexpr int TN_T = 256;
constexpr int TN_NP = 2;                             // pieces per operand (h, l)
constexpr int TN_STAGE = 2 * TN_NP * 2 * 256 * 16;   // bytes: [operand 2][piece 2][k half 2][slot 256][16]
constexpr int TN_MAX_JOBS = 4;

struct TnParams {
    const float *x[TN_MAX_JOBS], *y[TN_MAX_JOBS];
    float *out[TN_MAX_JOBS];
    int out_pitch[TN_MAX_JOBS];
    const float *xmax[TN_MAX_JOBS], *ymax[TN_MAX_JOBS];      // the operands' largest |element| (device words)
    float *partial;      // (njobs, nslabs, 256 slots, 256 slots)
    long m;
    int steps_per_slab, nslabs;
};

// one (operand, channel) unit of a k-step: channel c of the four rows a thread holds -> the two fp16 pieces of its four scaled k,
// 8 bytes each (split2_word: v_pk_mul_f32, v_cvt_pk_f16_f32, v_fma_mixlo_f16 / v_fma_mixhi_f16)
template <int C>
__device__ __forceinline__ void tn_split_store(char *stage, int operand, const f4 (&v)[4], float scale, int cq, int kq) {
    const SplitWord w0 = split2_word(v[0][C], v[1][C], scale), w1 = split2_word(v[2][C], v[3][C], scale);
    const unsigned slot = (unsigned)(C * 64 + cq);
    char *dst = stage + ((((unsigned)operand * TN_NP) * 2u + (unsigned)(kq >> 1)) * 256u + slot) * 16u + (unsigned)(kq & 1) * 8u;
    *reinterpret_cast<uint2 *>(dst) = make_uint2(w0.h, w1.h);
    *reinterpret_cast<uint2 *>(dst + 2 * 256 * 16) = make_uint2(w0.l, w1.l);
}

__global__ __launch_bounds__(TN_T) void tn_gemm256_split_kernel(const TnParams Q) {
    extern __shared__ __attribute__((aligned(16))) char s_stage[];      // two stages
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wa = wave >> 1, wb = wave & 1, i = lane & 31, h = lane >> 5;
    const int cq = lane, kq = __builtin_amdgcn_readfirstlane(wave);      // the rows a wave loads are uniform: scalar address arithmetic
    const int job = blockIdx.y, slab = blockIdx.x;
    const float *X = Q.x[job], *Y = Q.y[job];
    const float sx = lane_scale_of(__float_as_uint(ldc(Q.xmax[job]))).s, sy = lane_scale_of(__float_as_uint(ldc(Q.ymax[job]))).s;
    const long m = Q.m;
    // the slabs are INTERLEAVED: workgroup `slab` takes the 16-row steps slab, slab + nslabs, slab + 2 nslabs, ...  The grid walks
    // its slabs in lockstep, so at any moment the 256 CUs stream one contiguous stretch of x and of y -- with contiguous slabs they
    // would be 2 MiB apart and on the same HBM channels (measured: 3 TB/s)
    const long total_steps = (m + 15) / 16;
    const int nsteps = (int)(total_steps > slab ? (total_steps - slab + Q.nslabs - 1) / Q.nslabs : 0);

    f16v acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // rows of step `step` this thread holds that exist (0..4); loads of the others are clamped to row 0 and ignored by the split
    // Rows through BUFFER loads: a row at or past m (the tail of the last step, the steps that round a slab up to a multiple of four)
    // is out of the resource's range and reads as zero -- no mask, no clamp, no branch in the pinned schedule of a k-step.
    // The row offset is wave-uniform (kq) and goes into the instruction's scalar offset.
    const __amdgpu_buffer_rsrc_t RX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(X), 0, (int)(unsigned)(m * 1024), 0x00020000);
    const __amdgpu_buffer_rsrc_t RY = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Y), 0, (int)(unsigned)(m * 1024), 0x00020000);      // (m < 2^22: the byte count fits 32 bits)
    auto gload = [&](const __amdgpu_buffer_rsrc_t &R, int step, f4 (&v)[4]) {
        const unsigned r0 = (unsigned)((step * Q.nslabs + slab) * 16 + 4 * kq);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned r = r0 + e < (unsigned)m ? r0 + e : (unsigned)m;      // (scalar; keeps the 32-bit byte offset from wrapping)
            v[e] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(R, cq * 16, (int)(r * 1024u), 0));
        }
    };
    // this wave's fragments of a stage: A = x slots 128 wa + 32 ib + i, B = y slots 128 wb + 32 jb + i; k half h
    const unsigned a_off = (unsigned)(h * 256 + 128 * wa + i) * 16u;
    const unsigned b_off = (unsigned)((TN_NP * 2 + h) * 256 + 128 * wb + i) * 16u;
    auto frag = [&](const char *stage, unsigned off, int piece, int blk) {
        return *reinterpret_cast<const u4v *>(stage + off + (unsigned)(piece * 2 * 256 + 32 * blk) * 16u);
    };
    // One k-step: 4 row blocks x 3 products x 4 column blocks = 12 groups of four independent MFMAs (small terms first; a group keeps
    // the matrix pipe busy for 128 cycles).  Between the groups, in their shadow: the eight split-and-store units of the NEXT step
    // (held in nx, ny since the step before) into the other stage, and -- as soon as an operand's four units are done -- the loads
    // of the step after that into the same registers.
    auto step = [&](const char *cur, char *nxt, f4 (&nx)[4], f4 (&ny)[4], int t) {
        u4v bf[4][TN_NP], af[2][TN_NP];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int p = 0; p < TN_NP; ++p) bf[jb][p] = frag(cur, b_off, p, jb);
#pragma unroll
        for (int p = 0; p < TN_NP; ++p) af[0][p] = frag(cur, a_off, p, 0);
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
            // one row block: 24 MFMAs with, between them, two split-and-store units, the next row block's fragments and (odd blocks)
            // the loads that refill the operand whose four units are done -- pinned one MFMA : three VALU (hipcc left alone issues the
            // MFMAs back to back and the wave waits at each for the pipe; everything else then runs behind them, unhidden)
            const u4v(&A)[TN_NP] = af[ib & 1];
            if (ib < 3) {
#pragma unroll
                for (int p = 0; p < TN_NP; ++p) af[(ib + 1) & 1][p] = frag(cur, a_off, p, ib + 1);
            }
#define TN_PRODUCT(pa, pb) _Pragma("unroll") for (int jb = 0; jb < 4; ++jb) acc[ib][jb] = mfma_h(A[pa], bf[jb][pb], acc[ib][jb]);
            TN_PRODUCT(1, 0) TN_PRODUCT(0, 1) TN_PRODUCT(0, 0)
#undef TN_PRODUCT
            if (ib == 0) { tn_split_store<0>(nxt, 0, nx, sx, cq, kq); tn_split_store<1>(nxt, 0, nx, sx, cq, kq); }
            if (ib == 1) { tn_split_store<2>(nxt, 0, nx, sx, cq, kq); tn_split_store<3>(nxt, 0, nx, sx, cq, kq); gload(RX, t + 5, nx); }
            if (ib == 2) { tn_split_store<0>(nxt, 1, ny, sy, cq, kq); tn_split_store<1>(nxt, 1, ny, sy, cq, kq); }
            if (ib == 3) { tn_split_store<2>(nxt, 1, ny, sy, cq, kq); tn_split_store<3>(nxt, 1, ny, sy, cq, kq); gload(RY, t + 5, ny); }
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                if (k < TN_NP) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                // next block's fragments
                if (k % 3 == 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);               // a piece leaves
                if ((ib & 1) && k >= 8) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);       // a row is requested
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };

    // four register sets: the rows of steps t + 1 .. t + 4 are in flight or waiting while step t multiplies -- 96-128 KiB per CU, what it
    // takes to cover the HBM latency at 1 KiB per wave and load (with two sets the kernel ran at 2.5 TB/s: Little's law, not the matrix pipe)
    f4 x0[4], y0[4], x1[4], y1[4], x2[4], y2[4], x3[4], y3[4];
    gload(RX, 0, x0); gload(RY, 0, y0);
    gload(RX, 1, x1); gload(RY, 1, y1);
    gload(RX, 2, x2); gload(RY, 2, y2);
    gload(RX, 3, x3); gload(RY, 3, y3);
    {
        tn_split_store<0>(s_stage, 0, x0, sx, cq, kq); tn_split_store<1>(s_stage, 0, x0, sx, cq, kq);
        tn_split_store<2>(s_stage, 0, x0, sx, cq, kq); tn_split_store<3>(s_stage, 0, x0, sx, cq, kq);
        tn_split_store<0>(s_stage, 1, y0, sy, cq, kq); tn_split_store<1>(s_stage, 1, y0, sy, cq, kq);
        tn_split_store<2>(s_stage, 1, y0, sy, cq, kq); tn_split_store<3>(s_stage, 1, y0, sy, cq, kq);
    }
    gload(RX, 4, x0); gload(RY, 4, y0);
    __syncthreads();
    for (int t = 0; t < nsteps; t += 4) {      // (steps past nsteps multiply zeros: masked rows)
        step(s_stage, s_stage + TN_STAGE, x1, y1, t);
        step(s_stage + TN_STAGE, s_stage, x2, y2, t + 1);
        step(s_stage, s_stage + TN_STAGE, x3, y3, t + 2);
        step(s_stage + TN_STAGE, s_stage, x0, y0, t + 3);
    }
    // ---- partial block in slot order: D[row 8 (r / 4) + 4 h + r % 4][col i] of every 32 x 32 tile -------------------------------
    float *P = Q.partial + ((size_t)job * Q.nslabs + slab) * 65536;
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int sa = 128 * wa + 32 * ib + 8 * (r >> 2) + 4 * h + (r & 3), sb = 128 * wb + 32 * jb + i;
                P[(size_t)sa * 256 + sb] = acc[ib][jb][r];
            }
}

// out[channel(sa)][channel(sb)] = sum over the slabs, channel(slot) = 4 (slot % 64) + slot / 64
__global__ __launch_bounds__(256) void tn_gemm256_reduce_kernel(const TnParams Q) {
    const int job = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;      // sa * 256 + sb
    const float *P = Q.partial + (size_t)job * Q.nslabs * 65536 + e;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 4 <= Q.nslabs; s += 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] += P[(size_t)(s + q) * 65536];
    }
    for (; s < Q.nslabs; ++s) a[0] += P[(size_t)s * 65536];
    const int sa = e >> 8, sb = e & 255;
    const int ca = 4 * (sa & 63) + (sa >> 6), cb = 4 * (sb & 63) + (sb >> 6);
    const float c = lane_scale_of(__float_as_uint(ldc(Q.xmax[job]))).inv * lane_scale_of(__float_as_uint(ldc(Q.ymax[job]))).inv;      // exact: powers of two
    Q.out[job][(size_t)ca * Q.out_pitch[job] + cb] = ((a[0] + a[1]) + (a[2] + a[3])) * c;
}

__global__ __launch_bounds__(256) void absmax_kernel(const float *__restrict__ x, long n, float *__restrict__ amax) {
    float m = 0.f;
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f4 t = reinterpret_cast<const f4 *>(x)[i];
        m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(t.x)), __builtin_fabsf(t.y));
        m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(t.z)), __builtin_fabsf(t.w));
    }
    if (blockIdx.x == 0 && threadIdx.x < n - 4 * n4) m = __builtin_fmaxf(m, __builtin_fabsf(x[4 * n4 + threadIdx.x]));
    tensor_amax_update(amax, __float_as_uint(m));
}

}  // namespace

extern "C" int rtk_tn_gemm256_split(int njobs, const rtk_tn_job_t *jobs, long m, float *workspace, long workspace_floats, rtk_stream_t stream) {
    RTK_REQUIRE(njobs >= 1 && njobs <= TN_MAX_JOBS && jobs && m > 0 && m < (1L << 22) && workspace, "rtk_tn_gemm256_split: bad arguments (%d jobs, m = %ld)", njobs, m);
    TnParams Q = {};
    for (int k = 0; k < njobs; ++k) {
        RTK_REQUIRE(jobs[k].x && jobs[k].y && jobs[k].out && jobs[k].out_pitch >= 256, "rtk_tn_gemm256_split: job %d incomplete", k);
        RTK_REQUIRE(((size_t)jobs[k].x & 15) == 0 && ((size_t)jobs[k].y & 15) == 0, "rtk_tn_gemm256_split: operands must be 16-byte aligned");
        RTK_REQUIRE(jobs[k].x_amax && jobs[k].y_amax, "rtk_tn_gemm256_split: job %d without the operands' largest |element| (rtk_absmax)", k);
        Q.x[k] = jobs[k].x; Q.y[k] = jobs[k].y; Q.out[k] = jobs[k].out; Q.out_pitch[k] = jobs[k].out_pitch;
        Q.xmax[k] = jobs[k].x_amax; Q.ymax[k] = jobs[k].y_amax;
    }
    // about one workgroup per CU, slabs of a multiple of four (>= 8) 16-row steps
    const long total = (m + 15) / 16;
    long per = (total + 256 / njobs - 1) / (256 / njobs);
    per = per < 8 ? 8 : per;
    per = (per + 3) / 4 * 4;
    long nslabs = (total + per - 1) / per;
    while (nslabs * njobs * 65536 > workspace_floats && nslabs > 1) {      // a small workspace: fewer, longer slabs
        per *= 2;
        nslabs = (total + per - 1) / per;
    }
    RTK_REQUIRE(nslabs * njobs * 65536 <= workspace_floats, "rtk_tn_gemm256_split: workspace of %ld floats < %ld", workspace_floats,
                (long)njobs * 65536);
    Q.partial = workspace; Q.m = m; Q.steps_per_slab = (int)per; Q.nslabs = (int)nslabs;
    hipStream_t s = (hipStream_t)stream;
    (void)hipFuncSetAttribute((const void *)tn_gemm256_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TN_STAGE);
    tn_gemm256_split_kernel<<<dim3((unsigned)nslabs, njobs), TN_T, 2 * TN_STAGE, s>>>(Q);
    RTK_CHECK_LAUNCH("rtk_tn_gemm256_split");
    tn_gemm256_reduce_kernel<<<dim3(256, njobs), 256, 0, s>>>(Q);
    RTK_CHECK_LAUNCH("rtk_tn_gemm256_split");
    return RTK_OK;
}

extern "C" int rtk_absmax(const float *x, long n, float *amax, rtk_stream_t stream) {
    RTK_REQUIRE(x && n > 0 && amax && ((size_t)x & 15) == 0, "rtk_absmax: bad arguments");
    long blocks = (n / 4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    absmax_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(x, n, amax);
    RTK_CHECK_LAUNCH("rtk_absmax");
    return RTK_OK;
}
