// ops_pointnet2.hip -- the ten pointnet2 ops + knn_point as gfx950 HIP kernels (C ABI in
// include/rtk_pointnet2.h).  Written for CDNA4: 64-lane wavefronts, DPP wave reductions, source
// clouds staged in LDS as SoA, one wave per serial problem instead of one thread.
//
// Bit-exactness contract: every distance goes through rtk_sqdist() (explicit __fmaf_rn chain) and
// the file is compiled with -ffp-contract=off, so nothing else is fused; results equal
// oracle/pointnet2_ref.c bit for bit on indices and on three_nn / three_interpolate floats.
#include <math.h>
#include <stdarg.h>
#include <string.h>

#include "rtk_common.h"
#include "rtk_fused.h"

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void rtk_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *rtk_last_error(void) { return g_err; }
extern "C" int rtk_version(void) { return (0 << 16) | 3; }

// ------------------------------------------------------------------------------------------------
// wave64 reductions on DPP (no LDS, no barriers).  row_shr:1,2,4,8 leave each row's maximum in its
// lane 15; row_bcast:15 / row_bcast:31 carry it across rows so lane 63 holds the wave result.
// Lanes without a valid DPP source read 0 (bound_ctrl) / keep old = 0, the identity of unsigned max,
// so each step is one v_max_u32 with a DPP operand.
// ------------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_max_u32(unsigned v) {
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true);
    return v > o ? v : o;
}

__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = dpp_max_u32<0x111, 0xf>(v);  // row_shr:1
    v = dpp_max_u32<0x112, 0xf>(v);  // row_shr:2
    v = dpp_max_u32<0x114, 0xf>(v);  // row_shr:4
    v = dpp_max_u32<0x118, 0xf>(v);  // row_shr:8
    v = dpp_max_u32<0x142, 0xa>(v);  // row_bcast:15 -> rows 1,3
    v = dpp_max_u32<0x143, 0xc>(v);  // row_bcast:31 -> rows 2,3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// 64-bit key variant (used by the n > 2048 fallback)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void key_max_step(unsigned &hi, unsigned &lo) {
    const unsigned h2 = (unsigned)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, ROW_MASK, 0xf, false);
    const unsigned l2 = (unsigned)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, ROW_MASK, 0xf, false);
    const bool take = (h2 > hi) || (h2 == hi && l2 > lo);
    hi = take ? h2 : hi;
    lo = take ? l2 : lo;
}

__device__ __forceinline__ void wave_key_max(unsigned &hi, unsigned &lo) {
    key_max_step<0x111, 0xf>(hi, lo);
    key_max_step<0x112, 0xf>(hi, lo);
    key_max_step<0x114, 0xf>(hi, lo);
    key_max_step<0x118, 0xf>(hi, lo);
    key_max_step<0x142, 0xa>(hi, lo);
    key_max_step<0x143, 0xc>(hi, lo);
    hi = (unsigned)__builtin_amdgcn_readlane((int)hi, 63);
    lo = (unsigned)__builtin_amdgcn_readlane((int)lo, 63);
}

// ------------------------------------------------------------------------------------------------
// furthest point sampling   (reference: sampling_gpu.cu:94-209)
//
// The reference runs one CUDA block per sample with block = 2^floor(log2 n) threads and 511 block-wide
// shared-memory tree reductions.  Its selection rule, which everything below reproduces bit for bit:
//   * thread tid scans k = tid, tid+block, ... and keeps the FIRST maximum (strict >, :136-137);
//   * the halving tree (:143-203) merges slot t+s into slot t for s = block/2 ... 1 and keeps slot t's entry
//     on equal values (__update, :86-91), so among threads holding the maximum the one with the smallest
//     BIT-REVERSED tid wins (tids 1 and 2 tied -> tid 2).
// Hence: winner = maximum of the running min-distance, ties broken by the key (bitrev(k mod block), k div block).
//
// Here one WAVE owns one sample and the 511-round dependent chain contains no barrier:
//   * the n <= 64*PPL points and their running min-distances live in registers, laid out in the
//     reference's TIE ORDER: position p enumerates the points by ascending (bitrev(k mod block), k div block),
//     so "ties -> reference winner" becomes "ties -> smallest position";
//   * the round's maximum is a 6-instruction DPP max of the distance bits (non-negative floats order
//     like unsigned ints); the winner is the first set bit of ballot(t == max) -- no index travels through
//     the reduction;
//   * the winner's (x, y, z, k) comes back with one uniform-address LDS read;
//   * once the maximum is 0 every remaining pick is point 0 (all distances stay 0, position 0 = point 0 wins
//     every tie), so over-sampled levels (n < npoint, or duplicated points) stop early -- exact;
//   * `tie` reports whether any round had more than one position at a non-zero maximum: without such a tie the
//     selection is independent of the tie rule, which is what allows the level-2/3 shortcut of
//     rtk_fps_relevel below.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int fps_brev(int r, int bits) { return bits ? (int)(__brev((unsigned)r) >> (32 - bits)) : 0; }

// Number of preference ranks i' < i whose residue bitrev(i') owns an extra point, i.e. bitrev(i') < rem.
// Digit walk over the set bits of i: the ranks that share i's bits above t and have bit t clear are the numbers
// bitrev = F * 2^(bits-t) + L with t free high bits F and the fixed low part L.
__device__ __forceinline__ int fps_extra_before(int i, int bits, int rem) {
    int cnt = 0;
    for (int t = bits - 1; t >= 0; --t) {
        if (!((i >> t) & 1)) continue;
        const int hi = (i >> (t + 1)) << (t + 1);
        const int L = fps_brev(hi, bits);
        int c = rem > L ? (rem - L + (1 << (bits - t)) - 1) >> (bits - t) : 0;
        c = c < (1 << t) ? c : (1 << t);
        cnt += c;
    }
    return cnt;
}

// position of point k in the tie order: ranks (= bit-reversed residues) ascending, q or q+1 points per residue
__device__ __forceinline__ int fps_index_to_pos(int k, int block, int bits, int q, int rem) {
    const int i = fps_brev(k & (block - 1), bits);
    return i * q + fps_extra_before(i, bits, rem) + (k >> bits);
}

typedef float f2 __attribute__((ext_vector_type(2)));

// One FPS run by one wave.  s_pt: n float4 of LDS.  Returns the number of picks made before the cloud was exhausted.
template <int PPL>
__device__ __forceinline__ int fps_wave_body(int n, int m, int block, const float *__restrict__ xyz, float *__restrict__ temp,
                                             int *__restrict__ idxs, float *__restrict__ new_xyz, float4 *s_pt, int lane,
                                             int &tie_out, float *__restrict__ snap = nullptr, int *__restrict__ first_tie = nullptr,
                                             int settle_from = -1, int nu_prev = 0, int j_start = 1,
                                             const float *__restrict__ st_snap = nullptr, const int *__restrict__ st_i1 = nullptr,
                                             const int *__restrict__ st_i2 = nullptr, int ps = 3, int cs = 1, int *first_tie_val = nullptr) {
    // (ps, cs): component c of point k sits at xyz[k * ps + c * cs] -- (3, 1) for point-major clouds, (1, pitch) for the API's
    // channel-major (3, n) tensors (rtk_geometry_front reads those directly; level 1 only: the copies of the resumed / settled paths
    // below are point-major).  first_tie_val (optional): the first tied round as a value (0: none), for a caller that goes on in-kernel.
    // Re-levelling (rtk_fps_relevel): settle_from = the previous level's last tied round; j_start > 1 = resume: rounds < j_start are
    // the identity (every level repeats level 1 exactly before level 1's FIRST tied round) and the min-distance state of round
    // j_start is level 1's, saved by ORIGINAL point index in st_snap -- a function of the point, carried to this level's cloud
    // through the index lists: cloud index k -> st_i2[k] (level-2 index, if this is level 3) -> st_i1[...] (original point).
    // snap / first_tie (optional outputs): at the FIRST round with a tie, that round's number and the min-distance state it started
    // from, by cloud index.
    const int bits = 31 - __builtin_clz(block);
    const int q = n >> bits, rem = n & (block - 1);
    for (int k = lane; k < n; k += 64)      // coalesced read of the cloud, scattered into tie order
        s_pt[fps_index_to_pos(k, block, bits, q, rem)] = make_float4(xyz[k * ps], xyz[k * ps + cs], xyz[k * ps + 2 * cs], __int_as_float(k));
    // s_pt belongs to THIS wave (the body is run by exactly one wave, also inside rtk_geometry_front's 256-thread workgroups, whose other
    // waves have left by then): a wave's LDS operations execute in order, so what the scattered writes need before the reads below is
    // that they have been issued and counted -- not a workgroup barrier
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // LANE-MAJOR layout: lane l owns positions PPL*l .. PPL*l + PPL-1, so "smallest position among the maxima" =
    // lowest lane with the maximum, then its lowest slot.  Slots are processed in pairs with packed fp32 math
    // (v_pk_add/mul/fma: same roundings as the scalar chain).
    constexpr int H = PPL / 2;
    f2 x[H], y[H], z[H];
    unsigned t[PPL];   // min-distance bits; padding slots hold 0 and coordinates of point 0 (distance stays 0)
    const float4 p0 = s_pt[0];   // position 0 is always point 0
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const int p = lane * PPL + i;
        const bool ok = p < n;
        const float4 v = ok ? s_pt[p] : p0;
        x[i / 2][i % 2] = v.x; y[i / 2][i % 2] = v.y; z[i / 2][i % 2] = v.z;
        float t0 = 1e10f;
        if (temp) t0 = temp[__float_as_int(v.w)];
        if (j_start > 1 && ok) {
            int k = __float_as_int(v.w);
            if (st_i2) k = st_i2[k];
            t0 = st_snap[st_i1[k]];
        }
        t[i] = ok ? __float_as_uint(t0) : 0u;
    }
    float4 o = p0;
    if (j_start > 1) {          // rounds < j_start: the identity; the last of them picked cloud point j_start - 1
        for (int jj = lane; jj < j_start; jj += 64) idxs[jj] = jj;
        for (int jj = lane; jj < 3 * j_start; jj += 64) new_xyz[jj] = xyz[jj];
        o = s_pt[fps_index_to_pos(j_start - 1, block, bits, q, rem)];
    } else if (lane == 0) {
        idxs[0] = 0;
        if (new_xyz) { new_xyz[0] = o.x; new_xyz[1] = o.y; new_xyz[2] = o.z; }
    }
    int tie = 0;            // last round whose maximum was attained by more than one position (0: none)
    int kmax = j_start - 1; // largest index picked so far (re-levelling: settle once the picked set is the prefix {0..j})
    bool settled = false;
    int j = j_start;
    for (; j < m; ++j) {
        // The winner's coordinates as REAL register pairs.  Left to itself hipcc broadcasts a component that sits in the odd half of
        // a pair -- o.y straight out of the ds_read_b128 destination -- through the packed instruction's op_sel modifier
        // (v_pk_add_f32 ..., v[14:15] op_sel:[0,1]), and on this hardware that form read a wrong value about once in 10^4 rounds
        // whenever waves of another batch's split-bf16 kernels shared the SIMD: a different -- valid -- point was picked, run to run
        // (round 4; every irreproducible build of this loop had the op_sel:[0,1] form, no reproducible one did).
        // The empty asm pins each broadcast in a pair of its own; tests/test_isa_cpu.py keeps the op_sel form out of this file.
        f2 ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z};
        asm volatile("" : "+v"(ox), "+v"(oy), "+v"(oz));
        unsigned mloc = 0u;
#pragma unroll
        for (int hh = 0; hh < H; ++hh) {
            const f2 dx = x[hh] - ox, dy = y[hh] - oy, dz = z[hh] - oz;
            f2 d = dx * dx;                                    // fp-contract is off: mul, then two explicit fmas
            d = __builtin_elementwise_fma(dy, dy, d);
            d = __builtin_elementwise_fma(dz, dz, d);
            // d >= +0 (or NaN, whose bits exceed every finite value), so the unsigned minimum of the bit patterns is
            // fminf(d, t) without the canonicalising v_max; padding slots hold t = 0 and stay 0
            const unsigned d0 = __float_as_uint(d[0]), d1 = __float_as_uint(d[1]);
            t[2 * hh] = d0 < t[2 * hh] ? d0 : t[2 * hh];
            t[2 * hh + 1] = d1 < t[2 * hh + 1] ? d1 : t[2 * hh + 1];
            const unsigned mm = t[2 * hh] > t[2 * hh + 1] ? t[2 * hh] : t[2 * hh + 1];
            mloc = mm > mloc ? mm : mloc;
        }
        // lowest (bits 0-7) and highest (bits 8-15) slot holding the lane's own maximum: independent of the wave
        // reduction, fills its DPP wait states
        int sl = 0, sh = 0;
#pragma unroll
        for (int i = PPL - 1; i >= 0; --i) sl = t[i] == mloc ? i : sl;
#pragma unroll
        for (int i = 0; i < PPL; ++i) sh = t[i] == mloc ? i : sh;
        int slh = sl | (sh << 8);
        asm volatile("" : "+v"(slh));             // keep it above the reduction (the compiler would sink it below)
        const unsigned M = wave_max_u32(mloc);
        if (M == 0u) break;                       // exhausted: every remaining pick is index 0
        const unsigned long long mask = __ballot(mloc == M);
        const int wl = __builtin_ctzll(mask);     // lowest lane holding the maximum
        const int w = __builtin_amdgcn_readlane(slh, wl);
        const bool tied_now = ((mask & (mask - 1)) != 0ull) | ((w >> 8) != (w & 0xff));
        if (tied_now && tie == 0 && snap) {       // wave-uniform, at most once per run: save the state this round started from
#pragma unroll
            for (int i = 0; i < PPL; ++i) {
                const int p = lane * PPL + i;
                if (p < n) snap[__float_as_int(s_pt[p].w)] = __uint_as_float(t[i]);
            }
            if (lane == 0) *first_tie = j;
            if (first_tie_val) *first_tie_val = j;
        }
        tie = tied_now ? j : tie;
        const int pos = wl * PPL + (w & 0xff);
        o = s_pt[pos];                            // one uniform-address b128 read; lane 0 stores from registers
        asm volatile("" : "+v"(o.x), "+v"(o.y), "+v"(o.z), "+v"(o.w));
        if (lane == 0) {
            idxs[j] = __float_as_int(o.w);
            if (new_xyz) { new_xyz[j * 3 + 0] = o.x; new_xyz[j * 3 + 1] = o.y; new_xyz[j * 3 + 2] = o.z; }
        }
        if (settle_from >= 0) {
            const int k = __builtin_amdgcn_readfirstlane(__float_as_int(o.w));
            kmax = k > kmax ? k : kmax;
            if (j >= settle_from && kmax <= j) { settled = true; ++j; break; }
        }
    }
    if (settled) {          // past the previous level's last tie with the prefix {0..j-1} picked: the rest is the identity
        for (int jj = j + lane; jj < m; jj += 64) idxs[jj] = jj < nu_prev ? jj : 0;
        for (int jj = 3 * j + lane; jj < 3 * m; jj += 64) new_xyz[jj] = xyz[jj];
        tie_out = tie;
        return nu_prev;
    }
    for (int jj = j + lane; jj < m; jj += 64) {
        idxs[jj] = 0;
        if (new_xyz) { new_xyz[jj * 3 + 0] = p0.x; new_xyz[jj * 3 + 1] = p0.y; new_xyz[jj * 3 + 2] = p0.z; }
    }
    if (temp) {
#pragma unroll
        for (int i = 0; i < PPL; ++i) {
            const int p = lane * PPL + i;
            if (p < n) temp[__float_as_int(s_pt[p].w)] = __uint_as_float(t[i]);
        }
    }
    tie_out = tie;
    return j;
}

// One level of one cloud by one wave; every pointer is the cloud's own.  tie_prev < 0: a first level; tie_prev == 0: re-levelling of a
// cloud whose previous level had no tie (the identity, copied); tie_prev > 0: re-levelling that runs the selection (resumed at
// j_start, settling past round tie_prev).  nvalid >= 0: padded cloud.  Returns the picks made before the cloud was exhausted.
// (Shared by fps_wave_kernel and geometry_front_kernel.)
template <int PPL>
__device__ __forceinline__ int fps_cloud_level(int n, int m, int block, const float *__restrict__ xyz, int ps, int cs, float *__restrict__ temp,
                                               int *__restrict__ idxs, float *__restrict__ new_xyz, float4 *s_pt, int lane, int &tie_out,
                                               int nvalid, float *__restrict__ snap, int *__restrict__ first_tie, int tie_prev, int nu_prev,
                                               int j_start, const float *__restrict__ snap1, const int *__restrict__ idx1,
                                               const int *__restrict__ idx2, int *first_tie_val = nullptr) {
    if (tie_prev == 0) {
        // Re-levelling (rtk_fps_relevel; n == m): the previous level's selection of this cloud had no tie, so this level is the
        // identity on the coordinates (proof at rtk_fps_relevel) -- idx = (0 .. U-1, 0, 0, ...), new_xyz = xyz, no tie.
        for (int jj = lane; jj < m; jj += 64) idxs[jj] = jj < nu_prev ? jj : 0;
        for (int jj = lane; jj < 3 * m; jj += 64) new_xyz[jj] = xyz[jj];
        tie_out = 0;
        return nu_prev;
    }
    if (nvalid >= 0) {     // padded batch: this sample's cloud is its first nvalid points; the tie rule follows ITS size
        n = nvalid < n ? nvalid : n;
        n = n < 1 ? 1 : n;
        block = 1 << (31 - __builtin_clz(n));      // == cuda_utils.h:10-14 for every n < 2^21 (checked exhaustively)
        block = block > 1024 ? 1024 : block;
    }
    return fps_wave_body<PPL>(n, m, block, xyz, temp, idxs, new_xyz, s_pt, lane, tie_out, snap, first_tie, tie_prev, tie_prev >= 0 ? nu_prev : 0,
                              j_start, snap1, idx1, idx2, ps, cs, first_tie_val);
}

template <int PPL>
__global__ __launch_bounds__(64) void fps_wave_kernel(int n, int m, int block, const float *__restrict__ xyz,
                                                      float *__restrict__ temp, int *__restrict__ idxs,
                                                      float *__restrict__ new_xyz, int *__restrict__ nuniq,
                                                      int *__restrict__ tie, const int *__restrict__ nvalid,
                                                      float *__restrict__ snap, int *__restrict__ first_tie,
                                                      const int *__restrict__ tie_prev, const int *__restrict__ nuniq_prev,
                                                      const int *__restrict__ first_tie1, const float *__restrict__ snap1, int snap_pitch,
                                                      const int *__restrict__ idx1, const int *__restrict__ idx2) {
    extern __shared__ __attribute__((aligned(16))) float4 s_pt[];   // (x, y, z, bits(k)) by position
    const int b = blockIdx.x;
    int tied;
    const int pitch = n;
    const int j = fps_cloud_level<PPL>(n, m, block, xyz + (size_t)b * pitch * 3, 3, 1, temp ? temp + (size_t)b * pitch : nullptr,
                                       idxs + (size_t)b * m, new_xyz ? new_xyz + (size_t)b * m * 3 : nullptr, s_pt, (int)threadIdx.x, tied,
                                       nvalid ? nvalid[b] : -1, snap ? snap + (size_t)b * pitch : nullptr, first_tie ? first_tie + b : nullptr,
                                       tie_prev ? tie_prev[b] : -1, tie_prev ? nuniq_prev[b] : 0,
                                       (first_tie1 && first_tie1[b] > 1 && first_tie1[b] < m) ? first_tie1[b] : 1,
                                       snap1 ? snap1 + (size_t)b * snap_pitch : nullptr, idx1 ? idx1 + (size_t)b * m : nullptr,
                                       idx2 ? idx2 + (size_t)b * m : nullptr);
    if (threadIdx.x == 0) {
        if (nuniq) nuniq[b] = j;
        if (tie) tie[b] = tied;
    }
}

// Levels 2.. of a PNHead: furthest point sampling of npoint out of the npoint centroids of the previous level
// (model_utils.py:415-417) -- rtk_fps_relevel, one launch of fps_wave_kernel per level.  Let P[0..U) be the previous level's
// selection (then the cloud is exhausted and index 0 repeats) and T the last round of that selection whose maximum was attained by
// more than one position (0 = none).
//   * The running min-distances of a selection depend only on the selected SET.  T = 0: every round of the previous level had a
//     UNIQUE maximum, the selection did not depend on the tie rule, and the same rounds on the same coordinates pick P[1], P[2], ...
//     again -- the level is the identity on the coordinates and is only copied (the kernel's prologue):
//     idx = (0 .. U-1, 0, 0, ...), new_xyz = xyz.
//   * T > 0: the reference re-breaks the previous level's ties by the POSITION in the new cloud (bit-reversed), which may differ
//     from the previous choice.  The level then runs the selection, (i) RESUMED at level 1's first tied round T0 -- before it every
//     level repeats level 1 exactly -- from the min-distance state level 1 saved there (snap, by original point index: a function
//     of the point, carried to the level's own cloud through the index lists), and (ii) STOPPED at the first round r >= T at which
//     the picked indices are exactly {0..r} (largest picked index <= r; typically T or T+1: two tied points picked in the other
//     order): from there on it is the identity again.  The level reports its own T for the next one.
// (Rounds 2-3 did both levels in one launch, fps_relevel_kernel; round 4 found that launch irreproducible next to other kernels and
// traced it to the op_sel form of the packed distance arithmetic -- see the round loop of fps_wave_body -- which the level-1
// kernel's code generation happened to avoid.  One kernel now serves every level.)

// General fallback for n > 2048: one 256-thread workgroup per sample, min-distances in global memory.
__global__ __launch_bounds__(256) void fps_block_kernel(int n, int m, int block, const float *__restrict__ xyz,
                                                        float *__restrict__ temp, int *__restrict__ idxs) {
    __shared__ unsigned s_hi[4], s_lo[4];
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int bits = 31 - __builtin_clz(block);
    xyz += (size_t)b * n * 3;
    temp += (size_t)b * n;
    idxs += (size_t)b * m;
    if (tid == 0) idxs[0] = 0;
    int old = 0;
    for (int j = 1; j < m; ++j) {
        const float ox = xyz[old * 3 + 0], oy = xyz[old * 3 + 1], oz = xyz[old * 3 + 2];
        unsigned hi = 0u, lo = 0u;
        for (int k = tid; k < n; k += 256) {
            const float d = rtk_sqdist(xyz[k * 3 + 0], xyz[k * 3 + 1], xyz[k * 3 + 2], ox, oy, oz);
            const float d2 = fminf(d, temp[k]);
            temp[k] = d2;
            // tie key of the reference: (bitrev(k mod block), k div block) ascending = complement descending
            const unsigned nr = ~(((unsigned)fps_brev(k & (block - 1), bits) << 16) | (unsigned)(k >> bits));
            const unsigned h = __float_as_uint(d2);
            const bool take = (h > hi) || (h == hi && nr > lo);
            hi = take ? h : hi;
            lo = take ? nr : lo;
        }
        wave_key_max(hi, lo);
        if (lane == 0) { s_hi[wave] = hi; s_lo[wave] = lo; }
        __syncthreads();
        hi = s_hi[0]; lo = s_lo[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const bool take = (s_hi[w] > hi) || (s_hi[w] == hi && s_lo[w] > lo);
            hi = take ? s_hi[w] : hi;
            lo = take ? s_lo[w] : lo;
        }
        const unsigned rank = ~lo;
        old = (int)((rank & 0xffffu) * (unsigned)block) + fps_brev((int)(rank >> 16), bits);
        if (tid == 0) idxs[j] = old;
        __syncthreads();
    }
}

static int fps_block_size(int n) {  // cuda_utils.h:10-14 (host code in the reference as well)
    const int pow_2 = (int)(log((double)n) / log(2.0));
    int t = 1 << pow_2;
    if (t > 1024) t = 1024;
    if (t < 1) t = 1;
    return t;
}

static int fps_launch(int b, int n, int npoint, const float *xyz, float *temp, int *idx, float *new_xyz, int *nuniq,
                      int *tie, const int *nvalid, float *snap, int *first_tie, hipStream_t s, const int *tie_prev = nullptr,
                      const int *nuniq_prev = nullptr, const int *first_tie1 = nullptr, const float *snap1 = nullptr, int snap_pitch = 0,
                      const int *idx1 = nullptr, const int *idx2 = nullptr) {
    const int block = fps_block_size(n);
    RTK_REQUIRE(n / block < 65536, "furthest_point_sampling: n=%d too large", n);
    const size_t lds = (size_t)n * sizeof(float4);
    if (n <= 64 * 4) fps_wave_kernel<4><<<b, 64, lds, s>>>(n, npoint, block, xyz, temp, idx, new_xyz, nuniq, tie, nvalid, snap, first_tie, tie_prev, nuniq_prev,
                                              first_tie1, snap1, snap_pitch, idx1, idx2);
    else if (n <= 64 * 8) fps_wave_kernel<8><<<b, 64, lds, s>>>(n, npoint, block, xyz, temp, idx, new_xyz, nuniq, tie, nvalid, snap, first_tie, tie_prev, nuniq_prev,
                                              first_tie1, snap1, snap_pitch, idx1, idx2);
    else if (n <= 64 * 16) fps_wave_kernel<16><<<b, 64, lds, s>>>(n, npoint, block, xyz, temp, idx, new_xyz, nuniq, tie, nvalid, snap, first_tie, tie_prev, nuniq_prev,
                                              first_tie1, snap1, snap_pitch, idx1, idx2);
    else if (n <= 64 * 32) fps_wave_kernel<32><<<b, 64, lds, s>>>(n, npoint, block, xyz, temp, idx, new_xyz, nuniq, tie, nvalid, snap, first_tie, tie_prev, nuniq_prev,
                                              first_tie1, snap1, snap_pitch, idx1, idx2);
    else return 1;   // caller falls back to the block kernel
    return 0;
}

extern "C" int rtk_furthest_point_sampling(int b, int n, int npoint, const float *xyz, float *temp, int *idx,
                                           rtk_stream_t stream) {
    RTK_REQUIRE(b > 0 && n > 0 && xyz && temp && idx, "furthest_point_sampling: bad arguments (b=%d n=%d)", b, n);
    if (npoint <= 0) return RTK_OK;
    hipStream_t s = (hipStream_t)stream;
    const int rc = fps_launch(b, n, npoint, xyz, temp, idx, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, s);
    if (rc < 0) return rc;
    if (rc == 1) fps_block_kernel<<<b, 256, 0, s>>>(n, npoint, fps_block_size(n), xyz, temp, idx);
    RTK_CHECK_LAUNCH("furthest_point_sampling");
    return RTK_OK;
}

extern "C" int rtk_fps_centroids(int b, int n, int npoint, const float *xyz, int *idx, float *new_xyz, int *nuniq,
                                 int *tie, const int *n_valid, float *snap, int *first_tie, rtk_stream_t stream) {
    RTK_REQUIRE(b > 0 && n > 0 && npoint > 0 && xyz && idx && new_xyz, "fps_centroids: bad arguments (b=%d n=%d)", b, n);
    RTK_REQUIRE(n <= 2048, "fps_centroids: n=%d > 2048 (use rtk_furthest_point_sampling + rtk_gather_points)", n);
    RTK_REQUIRE((snap == nullptr) == (first_tie == nullptr), "fps_centroids: snap and first_tie go together");
    const int rc = fps_launch(b, n, npoint, xyz, nullptr, idx, new_xyz, nuniq, tie, n_valid, snap, first_tie, (hipStream_t)stream);
    if (rc < 0) return rc;
    RTK_CHECK_LAUNCH("fps_centroids");
    return RTK_OK;
}

extern "C" int rtk_fps_relevel(int b, int npoint, int levels, const float *xyz1, const int *nuniq1, const int *tie, int *idx,
                               float *new_xyz, int *nuniq, int *tie_work, const int *idx1, const float *snap1, int snap_pitch,
                               const int *first_tie1, rtk_stream_t stream) {
    RTK_REQUIRE(b > 0 && npoint > 0 && levels > 0 && xyz1 && nuniq1 && tie && idx && new_xyz && nuniq,
                "fps_relevel: bad arguments (b=%d npoint=%d levels=%d)", b, npoint, levels);
    RTK_REQUIRE(npoint <= 2048, "fps_relevel: npoint=%d > 2048", npoint);
    RTK_REQUIRE(levels == 1 || tie_work, "fps_relevel: more than one level needs the (levels, b) tie workspace");
    RTK_REQUIRE((!idx1 && !snap1 && !first_tie1) || (idx1 && snap1 && first_tie1 && snap_pitch > 0),
                "fps_relevel: idx1, snap1 and first_tie1 go together");
    RTK_REQUIRE(!idx1 || levels <= 2, "fps_relevel: the saved state is carried through at most two levels");
    hipStream_t s = (hipStream_t)stream;
    const float *src = xyz1;
    const int *tie_prev = tie, *nuniq_prev = nuniq1;
    for (int l = 0; l < levels; ++l) {      // one launch per level: untied clouds copy, tied clouds select (resume + settle)
        int *tie_out = tie_work ? tie_work + (size_t)l * b : nullptr;
        float *xo = new_xyz + (size_t)l * b * npoint * 3;
        const int rc = fps_launch(b, npoint, npoint, src, nullptr, idx + (size_t)l * b * npoint, xo, nuniq + (size_t)l * b, tie_out, nullptr, nullptr,
                                  nullptr, s, tie_prev, nuniq_prev, first_tie1, snap1, snap_pitch, idx1, l > 0 ? idx + (size_t)(l - 1) * b * npoint : nullptr);
        RTK_REQUIRE(rc == 0, "fps_relevel: no kernel instance for npoint=%d", npoint);
        RTK_CHECK_LAUNCH("fps_relevel");
        src = xo; tie_prev = tie_out; nuniq_prev = nuniq + (size_t)l * b;
    }
    return RTK_OK;
}

// ------------------------------------------------------------------------------------------------
// gather_points / grad   (sampling_gpu.cu:8-24, 46-63)
// ------------------------------------------------------------------------------------------------
__global__ void gather_points_kernel(int c, int n, int m, const float *__restrict__ points,
                                     const int *__restrict__ idx, float *__restrict__ out) {
    const int bs = blockIdx.z, ci = blockIdx.y, pt = blockIdx.x * blockDim.x + threadIdx.x;
    if (pt >= m) return;
    out[((size_t)bs * c + ci) * m + pt] = points[((size_t)bs * c + ci) * n + idx[(size_t)bs * m + pt]];
}

__global__ void gather_points_grad_kernel(int c, int n, int m, const float *__restrict__ grad_out,
                                          const int *__restrict__ idx, float *__restrict__ grad_points) {
    const int bs = blockIdx.z, ci = blockIdx.y, pt = blockIdx.x * blockDim.x + threadIdx.x;
    if (pt >= m) return;
    atomicAdd(grad_points + ((size_t)bs * c + ci) * n + idx[(size_t)bs * m + pt], grad_out[((size_t)bs * c + ci) * m + pt]);
}

extern "C" int rtk_gather_points(int b, int c, int n, int npoint, const float *points, const int *idx, float *out,
                                 rtk_stream_t stream) {
    RTK_REQUIRE(b > 0 && c > 0 && n > 0 && npoint > 0 && points && idx && out, "gather_points: bad arguments");
    RTK_REQUIRE(c <= 65535 && b <= 65535, "gather_points: c/b exceed grid limits");
    gather_points_kernel<<<dim3(rtk_divup(npoint, 256), c, b), 256, 0, (hipStream_t)stream>>>(c, n, npoint, points, idx, out);
    RTK_CHECK_LAUNCH("gather_points");
    return RTK_OK;
}

extern "C" int rtk_gather_points_grad(int b, int c, int n, int npoint, const float *grad_out, const int *idx,
                                      float *grad_points, rtk_stream_t stream) {
    RTK_REQUIRE(b > 0 && c > 0 && n > 0 && npoint > 0 && grad_out && idx && grad_points, "gather_points_grad: bad arguments");
    RTK_REQUIRE(c <= 65535 && b <= 65535, "gather_points_grad: c/b exceed grid limits");
    gather_points_grad_kernel<<<dim3(rtk_divup(npoint, 256), c, b), 256, 0, (hipStream_t)stream>>>(c, n, npoint, grad_out, idx, grad_points);
    RTK_CHECK_LAUNCH("gather_points_grad");
    return RTK_OK;
}

// ------------------------------------------------------------------------------------------------
// ball query   (ball_query_gpu.cu:9-45)
//
// Reference: one thread per centroid, serial O(n) scan over AoS points.  Here: the sample's cloud is
// staged once per workgroup in LDS as SoA; a wave takes one centroid at a time, its 64 lanes test
// 64 consecutive points, __ballot gives the hit mask and popcount-below gives each hit its rank in
// index order -- the "first nsample in index order" semantics without any serial scan -- and the
// scan stops (wave-uniformly) once nsample hits are found.
// ------------------------------------------------------------------------------------------------
#ifndef BQ_WAVES
#define BQ_WAVES 4
#endif
#ifndef BQ_CENTROIDS_PER_WAVE
#define BQ_CENTROIDS_PER_WAVE 2      // (8: 24-33 us per launch, 2: 14-18 -- the per-centroid scan is a serial chain, more waves hide it)
#endif

template <bool USE_LDS>
__global__ __launch_bounds__(64 * BQ_WAVES) void ball_query_kernel(int n, int m, float radius2, int nsample,
                                                                    const float *__restrict__ new_xyz,
                                                                    const float *__restrict__ xyz,
                                                                    int *__restrict__ idx) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int bs = blockIdx.y;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    xyz += (size_t)bs * n * 3;
    float *sx = smem, *sy = smem + n, *sz = smem + 2 * n;
    if (USE_LDS) {
        for (int k = tid; k < n; k += 64 * BQ_WAVES) {
            sx[k] = xyz[k * 3 + 0];
            sy[k] = xyz[k * 3 + 1];
            sz[k] = xyz[k * 3 + 2];
        }
        __syncthreads();
    }
    const int c0 = (blockIdx.x * BQ_WAVES + wave) * BQ_CENTROIDS_PER_WAVE;
    for (int ci = 0; ci < BQ_CENTROIDS_PER_WAVE; ++ci) {
        const int pt = c0 + ci;
        if (pt >= m) break;
        const float *q = new_xyz + ((size_t)bs * m + pt) * 3;
        const float qx = q[0], qy = q[1], qz = q[2];
        int *out = idx + ((size_t)bs * m + pt) * nsample;
        int cnt = 0, first = -1;
        for (int base = 0; base < n && cnt < nsample; base += 64) {
            const int k = base + lane;
            bool hit = false;
            if (k < n) {
                const float px = USE_LDS ? sx[k] : xyz[k * 3 + 0];
                const float py = USE_LDS ? sy[k] : xyz[k * 3 + 1];
                const float pz = USE_LDS ? sz[k] : xyz[k * 3 + 2];
                hit = rtk_sqdist(qx, qy, qz, px, py, pz) < radius2;
            }
            const unsigned long long mask = __ballot(hit);
            if (mask) {
                if (first < 0) first = base + __builtin_ctzll(mask);
                const int rank = cnt + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
                if (hit && rank < nsample) out[rank] = k;
                cnt += __builtin_popcountll(mask);
            }
        }
        if (first >= 0 && cnt < nsample) {
            for (int l = cnt + lane; l < nsample; l += 64) out[l] = first;  // slots after the hits repeat the first hit
        }
    }
}

extern "C" int rtk_ball_query(int b, int n, int npoint, float radius, int nsample, const float *new_xyz,
                              const float *xyz, int *idx, rtk_stream_t stream) {
    RTK_REQUIRE(b > 0 && n > 0 && npoint > 0 && nsample > 0 && new_xyz && xyz && idx, "ball_query: bad arguments");
    RTK_REQUIRE(b <= 65535, "ball_query: b exceeds grid limits");
    const float r2 = radius * radius;  // fp32 product, as ball_query_gpu.cu:23
    dim3 grid(rtk_divup(npoint, BQ_WAVES * BQ_CENTROIDS_PER_WAVE), b);
    const size_t lds = (size_t)n * 3 * sizeof(float);
    if (lds <= 64 * 1024)
        ball_query_kernel<true><<<grid, 64 * BQ_WAVES, lds, (hipStream_t)stream>>>(n, npoint, r2, nsample, new_xyz, xyz, idx);
    else
        ball_query_kernel<false><<<grid, 64 * BQ_WAVES, 0, (hipStream_t)stream>>>(n, npoint, r2, nsample, new_xyz, xyz, idx);
    RTK_CHECK_LAUNCH("ball_query");
    return RTK_OK;
}

// Both scales of an MSG level in one scan (radius1 <= radius2, so every scale-1 hit is a scale-2 hit).
// (body: workgroup `chunk` of sample `bs`; smem = 3 n floats.  Shared by ball_query_pair_kernel and geometry_tables_kernel.)
__device__ __forceinline__ void ball_query_pair_body(int bs, int chunk0, int chunk_stride, int n, int m, float r2a, int nsa, float r2b, int nsb,
                                                     const float *__restrict__ new_xyz, const float *__restrict__ xyz,
                                                     int *__restrict__ idxa, int *__restrict__ idxb, const int *__restrict__ nuniq,
                                                     float *smem, const int *__restrict__ src_nuniq = nullptr) {
    // src_nuniq (optional): source rows >= src_nuniq[bs] are copies of source row 0 (the centroids a level picked after its cloud was
    // exhausted).  They are hits exactly when row 0 is one, and they follow every unique hit in index order: only the unique prefix
    // is staged and scanned, and a ball that contains row 0 and is not full yet takes the indices src_nuniq[bs], src_nuniq[bs] + 1, ...
    // -- the table the full scan writes, at half the scan for the levels above an over-sampled one.
    // The workgroup takes chunks chunk0, chunk0 + chunk_stride, ... of its sample (a chunk = BQ_WAVES * BQ_CENTROIDS_PER_WAVE centroids):
    // the cloud is staged once, and a launch of a few thousand fat workgroups is not bound by the rate workgroups can be dispatched at
    // (rtk_geometry_tables: 35 k one-chunk workgroups took 90 us, the sum of the launches they came from).
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    constexpr int CPC = BQ_WAVES * BQ_CENTROIDS_PER_WAVE;
    xyz += (size_t)bs * n * 3;
    float *sx = smem, *sy = smem + n, *sz = smem + 2 * n;
    const int climit = nuniq ? min(m, nuniq[bs]) : m;
    if (chunk0 * CPC >= climit) return;      // whole workgroup beyond the unique centroids
    const int n_full = n;
    if (src_nuniq) n = max(1, min(n, src_nuniq[bs]));
    if (chunk0 == 0) {
        // rows [climit, next multiple of 32) read as zeros: a consumer's last tile of centroids may load the index rows of the (skipped)
        // duplicates next to the live ones before it masks them, and the tables need not be zero-filled by the caller for that
        const int z1 = min(m, (climit + 31) & ~31);
        int *za = idxa + ((size_t)bs * m + climit) * nsa, *zb = idxb + ((size_t)bs * m + climit) * nsb;
        for (int l = tid; l < (z1 - climit) * nsa; l += 64 * BQ_WAVES) za[l] = 0;
        for (int l = tid; l < (z1 - climit) * nsb; l += 64 * BQ_WAVES) zb[l] = 0;
    }
    for (int k = tid; k < n; k += 64 * BQ_WAVES) {
        sx[k] = xyz[k * 3 + 0];
        sy[k] = xyz[k * 3 + 1];
        sz[k] = xyz[k * 3 + 2];
    }
    // the coordinates of all of this wave's centroids in ONE round trip (lane l: component l % 3 of centroid l / 3), handed out by
    // shuffles: fetched at the top of each iteration they were eight dependent global-load latencies per wave -- most of the kernel;
    // the next chunk's are requested before this chunk's scan
    auto fetch = [&](int chunk) {
        const int c0 = (chunk * BQ_WAVES + wave) * BQ_CENTROIDS_PER_WAVE;
        float v = 0.f;
        if (lane < 3 * BQ_CENTROIDS_PER_WAVE && c0 + lane / 3 < climit) v = new_xyz[((size_t)bs * m + c0 + lane / 3) * 3 + lane % 3];
        return v;
    };
    float cq = fetch(chunk0);
    __syncthreads();
    for (int chunk = chunk0; chunk * CPC < climit; chunk += chunk_stride) {
        const int c0 = (chunk * BQ_WAVES + wave) * BQ_CENTROIDS_PER_WAVE;
        const float cq_next = (chunk_stride > 0 && (long)(chunk + chunk_stride) * CPC < climit) ? fetch(chunk + chunk_stride) : 0.f;
        for (int ci = 0; ci < BQ_CENTROIDS_PER_WAVE; ++ci) {
            const int pt = c0 + ci;
            if (pt >= climit) break;
            const float qx = __shfl(cq, 3 * ci, 64), qy = __shfl(cq, 3 * ci + 1, 64), qz = __shfl(cq, 3 * ci + 2, 64);
            int *oa = idxa + ((size_t)bs * m + pt) * nsa;
            int *ob = idxb + ((size_t)bs * m + pt) * nsb;
            int ca = 0, cb = 0, fa = -1, fb = -1;
            bool a0 = false, b0 = false;      // source row 0 inside the ball
            for (int base = 0; base < n && (ca < nsa || cb < nsb); base += 64) {
                const int k = base + lane;
                float d2 = INFINITY;
                if (k < n) d2 = rtk_sqdist(qx, qy, qz, sx[k], sy[k], sz[k]);
                const bool hb = d2 < r2b, ha = d2 < r2a;
                const unsigned long long mb = __ballot(hb);
                if (mb) {
                    const unsigned long long below = (1ull << lane) - 1ull;
                    if (base == 0) b0 = (mb & 1ull) != 0ull;
                    if (cb < nsb) {
                        if (fb < 0) fb = base + __builtin_ctzll(mb);
                        const int rank = cb + __builtin_popcountll(mb & below);
                        if (hb && rank < nsb) ob[rank] = k;
                        cb += __builtin_popcountll(mb);
                    }
                    const unsigned long long ma = __ballot(ha);
                    if (base == 0) a0 = (ma & 1ull) != 0ull;
                    if (ma && ca < nsa) {
                        if (fa < 0) fa = base + __builtin_ctzll(ma);
                        const int rank = ca + __builtin_popcountll(ma & below);
                        if (ha && rank < nsa) oa[rank] = k;
                        ca += __builtin_popcountll(ma);
                    }
                }
            }
            if (n < n_full) {                 // the copies of row 0 beyond the unique prefix, in index order
                if (a0 && ca < nsa) {
                    const int cnt = min(nsa - ca, n_full - n);
                    for (int l = lane; l < cnt; l += 64) oa[ca + l] = n + l;
                    ca += cnt;
                }
                if (b0 && cb < nsb) {
                    const int cnt = min(nsb - cb, n_full - n);
                    for (int l = lane; l < cnt; l += 64) ob[cb + l] = n + l;
                    cb += cnt;
                }
            }
            if (fa >= 0 && ca < nsa) for (int l = ca + lane; l < nsa; l += 64) oa[l] = fa;
            if (fb >= 0 && cb < nsb) for (int l = cb + lane; l < nsb; l += 64) ob[l] = fb;
            // an empty ball reads as zeros (what the caller's zero-initialisation leaves there in the reference, lib/pointnet2_utils.py:246):
            // written here, so that the fused geometry needs no fill of its 9 MB of tables
            if (fa < 0) for (int l = lane; l < nsa; l += 64) oa[l] = 0;
            if (fb < 0) for (int l = lane; l < nsb; l += 64) ob[l] = 0;
        }
        if (chunk_stride <= 0) break;
        cq = cq_next;
    }
}

__global__ __launch_bounds__(64 * BQ_WAVES) void ball_query_pair_kernel(int n, int m, float r2a, int nsa, float r2b, int nsb,
                                                                         const float *__restrict__ new_xyz,
                                                                         const float *__restrict__ xyz, int *__restrict__ idxa,
                                                                         int *__restrict__ idxb, const int *__restrict__ nuniq) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    ball_query_pair_body(blockIdx.y, blockIdx.x, 0, n, m, r2a, nsa, r2b, nsb, new_xyz, xyz, idxa, idxb, nuniq, smem);
}

extern "C" int rtk_ball_query_pair(int b, int n, int npoint, float radius1, int nsample1, float radius2, int nsample2,
                                   const float *new_xyz, const float *xyz, int *idx1, int *idx2, const int *nuniq,
                                   rtk_stream_t stream) {
    RTK_REQUIRE(b > 0 && n > 0 && npoint > 0 && nsample1 > 0 && nsample2 > 0 && new_xyz && xyz && idx1 && idx2, "ball_query_pair: bad arguments");
    RTK_REQUIRE(radius1 <= radius2, "ball_query_pair: radius1 (%g) must not exceed radius2 (%g)", radius1, radius2);
    RTK_REQUIRE(b <= 65535 && (size_t)n * 12 <= 64 * 1024, "ball_query_pair: b or n too large (n=%d)", n);
    dim3 grid(rtk_divup(npoint, BQ_WAVES * BQ_CENTROIDS_PER_WAVE), b);
    ball_query_pair_kernel<<<grid, 64 * BQ_WAVES, (size_t)n * 12, (hipStream_t)stream>>>(n, npoint, radius1 * radius1, nsample1,
                                                                                         radius2 * radius2, nsample2, new_xyz, xyz,
                                                                                         idx1, idx2, nuniq);
    RTK_CHECK_LAUNCH("ball_query_pair");
    return RTK_OK;
}

// ------------------------------------------------------------------------------------------------
// group_points / grad   (group_points_gpu.cu:47-66, 8-25)
// ------------------------------------------------------------------------------------------------
__global__ void group_points_kernel(int c, int n, int sn, const float *__restrict__ points,
                                    const int *__restrict__ idx, float *__restrict__ out) {
    const int bs = blockIdx.z, ci = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= sn) return;
    out[((size_t)bs * c + ci) * sn + t] = points[((size_t)bs * c + ci) * n + idx[(size_t)bs * sn + t]];
}

__global__ void group_points_grad_kernel(int c, int n, int sn, const float *__restrict__ grad_out,
                                         const int *__restrict__ idx, float *__restrict__ grad_points) {
    const int bs = blockIdx.z, ci = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= sn) return;
    atomicAdd(grad_points + ((size_t)bs * c + ci) * n + idx[(size_t)bs * sn + t], grad_out[((size_t)bs * c + ci) * sn + t]);
}

extern "C" int rtk_group_points(int b, int c, int n, int npoint, int nsample, const float *points, const int *idx,
                                float *out, rtk_stream_t stream) {
    RTK_REQUIRE(b > 0 && c > 0 && n > 0 && npoint > 0 && nsample > 0 && points && idx && out, "group_points: bad arguments");
    RTK_REQUIRE(c <= 65535 && b <= 65535, "group_points: c/b exceed grid limits");
    const int sn = npoint * nsample;
    group_points_kernel<<<dim3(rtk_divup(sn, 256), c, b), 256, 0, (hipStream_t)stream>>>(c, n, sn, points, idx, out);
    RTK_CHECK_LAUNCH("group_points");
    return RTK_OK;
}

// Scatter-add without global atomics: one workgroup owns one (batch, channel) row of grad_points, accumulates the
// npoint*nsample contributions in an LDS copy of the row (ds_add_f32: same-address collisions cost an LDS cycle each,
// not an L2 round trip) and writes the row back with plain stores.  The reference's per-element atomicAdd
// (group_points_gpu.cu:24) collapses when many neighbourhoods share points -- e.g. the 256 duplicate centroids of an
// over-sampled level hit the same 4-32 addresses 256 times per channel: 21 ms of a 92 ms train step at B=64.
template <bool SET, int CPB>      // CPB channels per workgroup: the index table is read once per CPB gradient planes
__global__ __launch_bounds__(256) void group_points_grad_lds_kernel(int c, int n, int npoint, int nsample, const float *__restrict__ grad_out,
                                                                    const int *__restrict__ idx, float *__restrict__ grad_points) {
    extern __shared__ float s_acc[];                               // [CPB][n]
    const int bs = blockIdx.y, c0 = blockIdx.x * CPB, tid = threadIdx.x;
    const int sn = npoint * nsample;
    for (int k = tid; k < CPB * n; k += 256) s_acc[k] = 0.f;
    __syncthreads();
    const float *go = grad_out + ((size_t)bs * c + c0) * sn;
    const int *id = idx + (size_t)bs * sn;
    if ((sn & 3) == 0) {
        // Coalesced: a thread takes one float4 of every plane (consecutive threads, consecutive 16 bytes), four chunks in flight.
        // A ball query pads a short neighbourhood with copies of its first hit, so equal neighbours inside a float4 are the
        // rule: they are summed in registers and hit the LDS once.
        const int4 *ip = reinterpret_cast<const int4 *>(id);
        const int n4 = sn >> 2;
        for (int base = 0; base < n4; base += 256 * 4) {
            int4 t[4];
            float4 g[4][CPB];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e4 = base + u * 256 + tid;
                if (e4 < n4) {
                    t[u] = ip[e4];
#pragma unroll
                    for (int q = 0; q < CPB; ++q) g[u][q] = *reinterpret_cast<const float4 *>(go + (size_t)q * sn + 4 * (size_t)e4);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (base + u * 256 + tid >= n4) continue;
                const int tt[4] = {t[u].x, t[u].y, t[u].z, t[u].w};
#pragma unroll
                for (int q = 0; q < CPB; ++q) {
                    const float gg[4] = {g[u][q].x, g[u][q].y, g[u][q].z, g[u][q].w};
                    float acc = gg[0];
#pragma unroll
                    for (int e = 1; e < 4; ++e) {
                        if (tt[e] == tt[e - 1]) {
                            acc += gg[e];
                        } else {
                            atomicAdd(&s_acc[q * n + tt[e - 1]], acc);
                            acc = gg[e];
                        }
                    }
                    atomicAdd(&s_acc[q * n + tt[3]], acc);
                }
            }
        }
    } else {
        for (int t = tid; t < sn; t += 256) {
            const int k = id[t];
#pragma unroll
            for (int q = 0; q < CPB; ++q) atomicAdd(&s_acc[q * n + k], go[(size_t)q * sn + t]);
        }
    }
    __syncthreads();
    float *gp = grad_points + ((size_t)bs * c + c0) * n;
    for (int k = tid; k < CPB * n; k += 256) gp[k] = SET ? s_acc[k] : gp[k] + s_acc[k];
}

static int group_points_grad_impl(bool set, int b, int c, int n, int npoint, int nsample, const float *grad_out, const int *idx,
                                  float *grad_points, rtk_stream_t stream) {
    RTK_REQUIRE(b > 0 && c > 0 && n > 0 && npoint > 0 && nsample > 0 && grad_out && idx && grad_points,
                "group_points_grad: bad arguments");
    RTK_REQUIRE(c <= 65535 && b <= 65535, "group_points_grad: c/b exceed grid limits");
    const int sn = npoint * nsample;
    hipStream_t s = (hipStream_t)stream;
    if ((size_t)n * sizeof(float) <= 64 * 1024) {
        const bool four = (c % 4 == 0) && (size_t)n * 4 * sizeof(float) <= 64 * 1024;
        const size_t lds = (size_t)n * sizeof(float) * (four ? 4 : 1);
        const dim3 grid(four ? c / 4 : c, b);
        if (set && four) group_points_grad_lds_kernel<true, 4><<<grid, 256, lds, s>>>(c, n, npoint, nsample, grad_out, idx, grad_points);
        else if (set) group_points_grad_lds_kernel<true, 1><<<grid, 256, lds, s>>>(c, n, npoint, nsample, grad_out, idx, grad_points);
        else if (four) group_points_grad_lds_kernel<false, 4><<<grid, 256, lds, s>>>(c, n, npoint, nsample, grad_out, idx, grad_points);
        else group_points_grad_lds_kernel<false, 1><<<grid, 256, lds, s>>>(c, n, npoint, nsample, grad_out, idx, grad_points);
    } else {
        if (set) (void)hipMemsetAsync(grad_points, 0, (size_t)b * c * n * sizeof(float), s);
        group_points_grad_kernel<<<dim3(rtk_divup(sn, 256), c, b), 256, 0, s>>>(c, n, sn, grad_out, idx, grad_points);
    }
    RTK_CHECK_LAUNCH("group_points_grad");
    return RTK_OK;
}

extern "C" int rtk_group_points_grad(int b, int c, int n, int npoint, int nsample, const float *grad_out,
                                     const int *idx, float *grad_points, rtk_stream_t stream) {
    return group_points_grad_impl(false, b, c, n, npoint, nsample, grad_out, idx, grad_points, stream);
}

extern "C" int rtk_group_points_grad_set(int b, int c, int n, int npoint, int nsample, const float *grad_out,
                                         const int *idx, float *grad_points, rtk_stream_t stream) {
    return group_points_grad_impl(true, b, c, n, npoint, nsample, grad_out, idx, grad_points, stream);
}

// ------------------------------------------------------------------------------------------------
// K-smallest selection as branch-free sorting networks over KV_LPQ lanes per query (4; 16 until round 6).
//
// The reference scans all candidates serially per query thread (three_nn, knn) or materialises the
// (B,S,N) distance matrix and calls torch.topk (knn_point).  Here KV_LPQ neighbouring lanes of a DPP row share a
// query: each lane sorts / merges its share of the candidates in registers with compare-exchange
// networks on 64-bit keys (distance bits : index) -- non-negative floats order like unsigned ints, so
// one v_cmp_lt_u64 implements "smaller distance, ties -> smaller index", exactly the order a serial
// strict-< scan produces -- and 4 DPP row_shl steps merge the 16 partial lists into lane 0.
// No divergence, no scratch (all register indices are static after unrolling).
// ------------------------------------------------------------------------------------------------
#define KEY_INF_D 0x7f800000u
#define KEY_INF_I 0x7fffffff

__device__ __forceinline__ void kv_cex(unsigned &da, int &ia, unsigned &db, int &ib) {   // (a, b) -> (min, max)
    const unsigned long long ka = ((unsigned long long)da << 32) | (unsigned)ia;
    const unsigned long long kb = ((unsigned long long)db << 32) | (unsigned)ib;
    const bool sw = kb < ka;
    const unsigned td = da; const int ti = ia;
    da = sw ? db : da; ia = sw ? ib : ia;
    db = sw ? td : db; ib = sw ? ti : ib;
}

template <int K>
__device__ __forceinline__ void kv_bitonic_sort(unsigned (&d)[K], int (&i)[K]) {   // ascending
#pragma unroll
    for (int k = 2; k <= K; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int a = 0; a < K; ++a) {
                const int l = a ^ j;
                if (l > a) {
                    if ((a & k) == 0) kv_cex(d[a], i[a], d[l], i[l]);
                    else kv_cex(d[l], i[l], d[a], i[a]);
                }
            }
        }
    }
}

// best <- the K smallest of (best U other), both ascending on entry, ascending on exit
template <int K>
__device__ __forceinline__ void kv_merge_low(unsigned (&d)[K], int (&i)[K], const unsigned (&od)[K], const int (&oi)[K]) {
#pragma unroll
    for (int a = 0; a < K; ++a) {   // half-cleaner against the reversed other list: keeps the K smallest (bitonic)
        const unsigned long long ka = ((unsigned long long)d[a] << 32) | (unsigned)i[a];
        const unsigned long long kb = ((unsigned long long)od[K - 1 - a] << 32) | (unsigned)oi[K - 1 - a];
        const bool sw = kb < ka;
        d[a] = sw ? od[K - 1 - a] : d[a];
        i[a] = sw ? oi[K - 1 - a] : i[a];
    }
#pragma unroll
    for (int j = K >> 1; j > 0; j >>= 1) {
#pragma unroll
        for (int a = 0; a < K; ++a) {
            const int l = a ^ j;
            if (l > a) kv_cex(d[a], i[a], d[l], i[l]);
        }
    }
}

template <int CTRL>
__device__ __forceinline__ unsigned dpp_pull(unsigned v) {   // row_shl:n -> lane i reads lane i+n of its row
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}

// merge the lists of the LPQ (4 or 16) consecutive lanes that share a query into the first of them (the other lanes end with garbage)
template <int K, int LPQ = 16>
__device__ __forceinline__ void kv_row_merge(unsigned (&d)[K], int (&i)[K]) {
    unsigned od[K];
    int oi[K];
#pragma unroll
    for (int a = 0; a < K; ++a) { od[a] = dpp_pull<0x101>(d[a]); oi[a] = (int)dpp_pull<0x101>((unsigned)i[a]); }
    kv_merge_low<K>(d, i, od, oi);
#pragma unroll
    for (int a = 0; a < K; ++a) { od[a] = dpp_pull<0x102>(d[a]); oi[a] = (int)dpp_pull<0x102>((unsigned)i[a]); }
    kv_merge_low<K>(d, i, od, oi);
    static_assert(LPQ == 4 || LPQ == 8 || LPQ == 16, "4, 8 or 16 lanes per query");
    if constexpr (LPQ >= 8) {
#pragma unroll
        for (int a = 0; a < K; ++a) { od[a] = dpp_pull<0x104>(d[a]); oi[a] = (int)dpp_pull<0x104>((unsigned)i[a]); }
        kv_merge_low<K>(d, i, od, oi);
    }
    if constexpr (LPQ == 16) {
#pragma unroll
        for (int a = 0; a < K; ++a) { od[a] = dpp_pull<0x108>(d[a]); oi[a] = (int)dpp_pull<0x108>((unsigned)i[a]); }
        kv_merge_low<K>(d, i, od, oi);
    }
}

// Lanes per query of the selection kernels.  Round 6: 4 (rounds 1-5: 16).  With 16 lanes the four cross-lane merge steps were a quarter
// (three-NN) to a half (kNN) of a query's instructions -- the fastest way to ONE query's answer, and at B = 64 a launch of them is
// bound by VALU issue, not by latency (three-NN: ~175 instructions per query with 16 lanes, ~100 with 4; kNN: ~475 -> ~220).
#ifndef KV_LPQ
#define KV_LPQ 4
#endif
constexpr int KV_QPW = 256 / KV_LPQ;      // queries per 256-thread workgroup (= per chunk)

// ------------------------------------------------------------------------------------------------
// three_nn   (interpolate_gpu.cu:81-124)
// Output: the 3 smallest squared distances ascending + indices, ties -> earlier index; fewer than 3
// known points leave (+inf, 0) in the unfilled slots (the reference's double 1e40 narrowed to float).
// ------------------------------------------------------------------------------------------------
// (body: workgroup `chunk` -- KV_QPW queries -- of sample `bs`; smem = 3 m floats with USE_LDS.  Shared by three_nn_kernel and
// geometry_tables_kernel.)
template <bool USE_LDS>
__device__ __forceinline__ void three_nn_body(int bs, int chunk0, int chunk_stride, int n, int m, const float *__restrict__ unknown,
                                              const float *__restrict__ known, float *__restrict__ dist2,
                                              int *__restrict__ idx, const int *__restrict__ unknown_nuniq,
                                              const int *__restrict__ known_nuniq, float *smem) {
    // chunks chunk0, chunk0 + chunk_stride, ... (KV_QPW queries each, KV_LPQ lanes per query; chunk_stride = 0: one chunk), the known
    // cloud staged once
    const int tid = threadIdx.x;
    const int qlimit = unknown_nuniq ? min(n, unknown_nuniq[bs]) : n;      // queries >= unknown_nuniq[bs] are duplicate rows: not computed
    if (chunk0 * KV_QPW >= qlimit) return;
    // known rows >= known_nuniq[b] are copies of known row 0: of those only the first two (lowest indices) can reach the
    // top 3, at the distance of row 0 -- scan the unique prefix and add these two candidates: identical result, half the scan
    const int m_full = m;
    const int ke = known_nuniq ? min(m, known_nuniq[bs]) : m;
    known += (size_t)bs * m * 3;
    m = ke;
    float *sx = smem, *sy = smem + m, *sz = smem + 2 * m;
    // the query's coordinates are requested before the staging loop and its barrier: one global round trip instead of two in a row
    // (and the next chunk's before this chunk's scan)
    const int li = tid & (KV_LPQ - 1);
    auto query = [&](int chunk, float &x, float &y, float &z) {
        const int pr = chunk * KV_QPW + tid / KV_LPQ;
        const float *u = unknown + ((size_t)bs * n + (pr < n ? pr : n - 1)) * 3;
        x = u[0]; y = u[1]; z = u[2];
    };
    float ux, uy, uz;
    query(chunk0, ux, uy, uz);
    if (USE_LDS) {
        for (int k = tid; k < m; k += 256) {
            sx[k] = known[k * 3 + 0];
            sy[k] = known[k * 3 + 1];
            sz[k] = known[k * 3 + 2];
        }
        __syncthreads();
    }
    for (int chunk = chunk0; chunk * KV_QPW < qlimit; chunk += chunk_stride) {
        const int pt_raw = chunk * KV_QPW + tid / KV_LPQ;
        const int pt = pt_raw < n ? pt_raw : n - 1;
        float nx = 0.f, ny = 0.f, nz = 0.f;
        if (chunk_stride > 0 && (chunk + chunk_stride) * KV_QPW < qlimit) query(chunk + chunk_stride, nx, ny, nz);
        unsigned d[4] = {KEY_INF_D, KEY_INF_D, KEY_INF_D, KEY_INF_D};
        int i[4] = {KEY_INF_I, KEY_INF_I, KEY_INF_I, KEY_INF_I};
        for (int k = li; k < m; k += KV_LPQ) {
            const float kx = USE_LDS ? sx[k] : known[k * 3 + 0];
            const float ky = USE_LDS ? sy[k] : known[k * 3 + 1];
            const float kz = USE_LDS ? sz[k] : known[k * 3 + 2];
            const float dd = rtk_sqdist(ux, uy, uz, kx, ky, kz);
            if (dd < INFINITY) {    // the reference's `d < 1e40` predicate: inf / NaN never enter
                d[3] = __float_as_uint(dd); i[3] = k;
                kv_cex(d[2], i[2], d[3], i[3]);
                kv_cex(d[1], i[1], d[2], i[2]);
                kv_cex(d[0], i[0], d[1], i[1]);
            }
        }
        if (li >= 1 && li <= 2 && ke + li - 1 < m_full) {          // the two possible duplicate-of-row-0 candidates: indices ke, ke+1
            const float kx = USE_LDS ? sx[0] : known[0], ky = USE_LDS ? sy[0] : known[1], kz = USE_LDS ? sz[0] : known[2];
            const float dd = rtk_sqdist(ux, uy, uz, kx, ky, kz);
            if (dd < INFINITY) {
                d[3] = __float_as_uint(dd); i[3] = ke + li - 1;
                kv_cex(d[2], i[2], d[3], i[3]);
                kv_cex(d[1], i[1], d[2], i[2]);
                kv_cex(d[0], i[0], d[1], i[1]);
            }
        }
        d[3] = KEY_INF_D; i[3] = KEY_INF_I;
        kv_row_merge<4, KV_LPQ>(d, i);
        if (li == 0 && pt_raw < n) {
            float *od = dist2 + ((size_t)bs * n + pt) * 3;
            int *oi = idx + ((size_t)bs * n + pt) * 3;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                od[a] = __uint_as_float(d[a]);
                oi[a] = d[a] == KEY_INF_D ? 0 : i[a];
            }
        }
        if (chunk_stride <= 0) break;
        ux = nx; uy = ny; uz = nz;
    }
}

template <bool USE_LDS>
__global__ __launch_bounds__(256) void three_nn_kernel(int n, int m, const float *__restrict__ unknown,
                                                       const float *__restrict__ known, float *__restrict__ dist2,
                                                       int *__restrict__ idx, const int *__restrict__ unknown_nuniq,
                                                       const int *__restrict__ known_nuniq) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    three_nn_body<USE_LDS>(blockIdx.y, blockIdx.x, 0, n, m, unknown, known, dist2, idx, unknown_nuniq, known_nuniq, smem);
}

extern "C" int rtk_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx,
                            rtk_stream_t stream) {
    RTK_REQUIRE(b > 0 && n > 0 && m > 0 && unknown && known && dist2 && idx, "three_nn: bad arguments");
    RTK_REQUIRE(b <= 65535, "three_nn: b exceeds grid limits");
    dim3 grid(rtk_divup(n, KV_QPW), b);
    const size_t lds = (size_t)m * 3 * sizeof(float);
    if (lds <= 64 * 1024)
        three_nn_kernel<true><<<grid, 256, lds, (hipStream_t)stream>>>(n, m, unknown, known, dist2, idx, nullptr, nullptr);
    else
        three_nn_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(n, m, unknown, known, dist2, idx, nullptr, nullptr);
    RTK_CHECK_LAUNCH("three_nn");
    return RTK_OK;
}

extern "C" int rtk_three_nn_masked(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx,
                                   const int *unknown_nuniq, const int *known_nuniq, rtk_stream_t stream) {
    RTK_REQUIRE(b > 0 && n > 0 && m > 0 && unknown && known && dist2 && idx, "three_nn_masked: bad arguments");
    RTK_REQUIRE(b <= 65535, "three_nn_masked: b exceeds grid limits");
    dim3 grid(rtk_divup(n, KV_QPW), b);
    const size_t lds = (size_t)m * 3 * sizeof(float);
    if (lds <= 64 * 1024)
        three_nn_kernel<true><<<grid, 256, lds, (hipStream_t)stream>>>(n, m, unknown, known, dist2, idx, unknown_nuniq, known_nuniq);
    else
        three_nn_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(n, m, unknown, known, dist2, idx, unknown_nuniq, known_nuniq);
    RTK_CHECK_LAUNCH("three_nn_masked");
    return RTK_OK;
}

// ------------------------------------------------------------------------------------------------
// knn   (interpolate_gpu.cu:9-57)  -- exported by the reference, unused on its live path.
// Straight restatement: ascending insertion list per query thread (k <= 200, as the reference).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void knn_kernel(int n, int m, int k, const float *__restrict__ unknown,
                                                 const float *__restrict__ known, float *__restrict__ dist2,
                                                 int *__restrict__ idx) {
    const int bs = blockIdx.y, pt = blockIdx.x * blockDim.x + threadIdx.x;
    if (pt >= n) return;
    const float *u = unknown + ((size_t)bs * n + pt) * 3;
    known += (size_t)bs * m * 3;
    const float ux = u[0], uy = u[1], uz = u[2];
    float best[200];
    int besti[200];
    for (int i = 0; i < k; ++i) { best[i] = INFINITY; besti[i] = 0; }
    for (int i = 0; i < m; ++i) {
        const float d = rtk_sqdist(ux, uy, uz, known[i * 3 + 0], known[i * 3 + 1], known[i * 3 + 2]);
        if (!(d < best[k - 1])) continue;
        int j = k - 1;
        while (j > 0 && d < best[j - 1]) { best[j] = best[j - 1]; besti[j] = besti[j - 1]; --j; }
        best[j] = d;
        besti[j] = i;
    }
    float *od = dist2 + ((size_t)bs * n + pt) * k;
    int *oi = idx + ((size_t)bs * n + pt) * k;
    for (int i = 0; i < k; ++i) { od[i] = best[i]; oi[i] = besti[i]; }
}

extern "C" int rtk_knn(int b, int n, int m, int k, const float *unknown, const float *known, float *dist2, int *idx,
                       rtk_stream_t stream) {
    RTK_REQUIRE(b > 0 && n > 0 && m > 0 && unknown && known && dist2 && idx, "knn: bad arguments");
    RTK_REQUIRE(k >= 1 && k <= 200, "knn: k=%d outside [1,200] (the reference's fixed best[200])", k);
    RTK_REQUIRE(b <= 65535, "knn: b exceeds grid limits");
    knn_kernel<<<dim3(rtk_divup(n, 64), b), 64, 0, (hipStream_t)stream>>>(n, m, k, unknown, known, dist2, idx);
    RTK_CHECK_LAUNCH("knn");
    return RTK_OK;
}

// ------------------------------------------------------------------------------------------------
// three_interpolate / grad   (interpolate_gpu.cu:149-169, 192-214)
// ------------------------------------------------------------------------------------------------
// A thread interpolates its point for TI_CPB channels: the three indices and weights are loaded once (with one channel per thread
// the launch was 16 384 workgroups of three gathers per thread at the training shapes).
template <int TI_CPB>       // 8 for large batches; 1 where that would leave most of the chip idle
__global__ void three_interpolate_kernel(int c, int m, int n, const float *__restrict__ points,
                                         const int *__restrict__ idx, const float *__restrict__ weight,
                                         float *__restrict__ out) {
    const int bs = blockIdx.z, c0 = blockIdx.y * TI_CPB, pt = blockIdx.x * blockDim.x + threadIdx.x;
    if (pt >= n) return;
    const float *w = weight + ((size_t)bs * n + pt) * 3;
    const int *id = idx + ((size_t)bs * n + pt) * 3;
    const float w0 = w[0], w1 = w[1], w2 = w[2];
    const int i0 = id[0], i1 = id[1], i2 = id[2];
    const int nc = min(TI_CPB, c - c0);
#pragma unroll
    for (int q = 0; q < TI_CPB; ++q) {
        if (q < nc) {
            const float *p = points + ((size_t)bs * c + c0 + q) * m;
            out[((size_t)bs * c + c0 + q) * n + pt] = __fmaf_rn(w2, p[i2], __fmaf_rn(w1, p[i1], __fmul_rn(w0, p[i0])));
        }
    }
}

__global__ void three_interpolate_grad_kernel(int c, int n, int m, const float *__restrict__ grad_out,
                                              const int *__restrict__ idx, const float *__restrict__ weight,
                                              float *__restrict__ grad_points) {
    const int bs = blockIdx.z, ci = blockIdx.y, pt = blockIdx.x * blockDim.x + threadIdx.x;
    if (pt >= n) return;
    const float g = grad_out[((size_t)bs * c + ci) * n + pt];
    const float *w = weight + ((size_t)bs * n + pt) * 3;
    const int *id = idx + ((size_t)bs * n + pt) * 3;
    float *gp = grad_points + ((size_t)bs * c + ci) * m;
    atomicAdd(gp + id[0], __fmul_rn(g, w[0]));
    atomicAdd(gp + id[1], __fmul_rn(g, w[1]));
    atomicAdd(gp + id[2], __fmul_rn(g, w[2]));
}

extern "C" int rtk_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                                     const float *weight, float *out, rtk_stream_t stream) {
    RTK_REQUIRE(b > 0 && c > 0 && m > 0 && n > 0 && points && idx && weight && out, "three_interpolate: bad arguments");
    RTK_REQUIRE(c <= 65535 && b <= 65535, "three_interpolate: c/b exceed grid limits");
    if ((long)b * c * rtk_divup(n, 256) >= 4096)
        three_interpolate_kernel<8><<<dim3(rtk_divup(n, 256), rtk_divup(c, 8), b), 256, 0, (hipStream_t)stream>>>(c, m, n, points, idx, weight, out);
    else
        three_interpolate_kernel<1><<<dim3(rtk_divup(n, 256), c, b), 256, 0, (hipStream_t)stream>>>(c, m, n, points, idx, weight, out);
    RTK_CHECK_LAUNCH("three_interpolate");
    return RTK_OK;
}

// As group_points_grad_lds_kernel: LDS accumulation, no global atomics.  One workgroup per (batch, TIG_CPB channels): the three
// indices and weights of a point are the same for every channel, so a thread loads them once and scatters TIG_CPB gradients with
// them (one channel per workgroup was 16 384 workgroups of one point per thread at the train-step shapes: pure launch latency).
constexpr int TIG_CPB = 8;
template <bool SET>
__global__ __launch_bounds__(256) void three_interpolate_grad_lds_kernel(int c, int n, int m, const float *__restrict__ grad_out,
                                                                         const int *__restrict__ idx, const float *__restrict__ weight,
                                                                         float *__restrict__ grad_points) {
    extern __shared__ float s_acc[];                               // [TIG_CPB][m]
    const int bs = blockIdx.y, c0 = blockIdx.x * TIG_CPB, tid = threadIdx.x;
    const int nc = min(TIG_CPB, c - c0);
    for (int k = tid; k < nc * m; k += 256) s_acc[k] = 0.f;
    __syncthreads();
    const float *go = grad_out + ((size_t)bs * c + c0) * n;
    for (int pt = tid; pt < n; pt += 256) {
        const float *w = weight + ((size_t)bs * n + pt) * 3;
        const int *id = idx + ((size_t)bs * n + pt) * 3;
        const float w0 = w[0], w1 = w[1], w2 = w[2];
        const int i0 = id[0], i1 = id[1], i2 = id[2];
        float g[TIG_CPB];
#pragma unroll
        for (int q = 0; q < TIG_CPB; ++q) g[q] = go[(size_t)min(q, nc - 1) * n + pt];      // unconditional (clamped) loads
#pragma unroll
        for (int q = 0; q < TIG_CPB; ++q) {
            if (q < nc) {
                atomicAdd(&s_acc[q * m + i0], __fmul_rn(g[q], w0));
                atomicAdd(&s_acc[q * m + i1], __fmul_rn(g[q], w1));
                atomicAdd(&s_acc[q * m + i2], __fmul_rn(g[q], w2));
            }
        }
    }
    __syncthreads();
    float *gp = grad_points + ((size_t)bs * c + c0) * m;
    for (int k = tid; k < nc * m; k += 256) gp[k] = SET ? s_acc[k] : gp[k] + s_acc[k];
}

static int three_interpolate_grad_impl(bool set, int b, int c, int n, int m, const float *grad_out, const int *idx, const float *weight,
                                       float *grad_points, rtk_stream_t stream) {
    RTK_REQUIRE(b > 0 && c > 0 && m > 0 && n > 0 && grad_out && idx && weight && grad_points,
                "three_interpolate_grad: bad arguments");
    RTK_REQUIRE(c <= 65535 && b <= 65535, "three_interpolate_grad: c/b exceed grid limits");
    hipStream_t s = (hipStream_t)stream;
    if ((size_t)TIG_CPB * m * sizeof(float) <= 64 * 1024) {
        const dim3 grid(rtk_divup(c, TIG_CPB), b);
        const size_t lds = (size_t)TIG_CPB * m * sizeof(float);
        if (set) three_interpolate_grad_lds_kernel<true><<<grid, 256, lds, s>>>(c, n, m, grad_out, idx, weight, grad_points);
        else three_interpolate_grad_lds_kernel<false><<<grid, 256, lds, s>>>(c, n, m, grad_out, idx, weight, grad_points);
    } else {
        if (set) (void)hipMemsetAsync(grad_points, 0, (size_t)b * c * m * sizeof(float), s);
        three_interpolate_grad_kernel<<<dim3(rtk_divup(n, 256), c, b), 256, 0, s>>>(c, n, m, grad_out, idx, weight, grad_points);
    }
    RTK_CHECK_LAUNCH("three_interpolate_grad");
    return RTK_OK;
}

extern "C" int rtk_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out, const int *idx,
                                          const float *weight, float *grad_points, rtk_stream_t stream) {
    return three_interpolate_grad_impl(false, b, c, n, m, grad_out, idx, weight, grad_points, stream);
}

extern "C" int rtk_three_interpolate_grad_set(int b, int c, int n, int m, const float *grad_out, const int *idx,
                                              const float *weight, float *grad_points, rtk_stream_t stream) {
    return three_interpolate_grad_impl(true, b, c, n, m, grad_out, idx, weight, grad_points, stream);
}

// ------------------------------------------------------------------------------------------------
// knn_point   (utils/model_utils/model_utils.py:17-39,85-99: square_distance + torch.topk)
//
// The reference materialises the (B,S,N) distance matrix with a matmul and runs torch.topk on it.
// Here KV_LPQ lanes share a query: each lane takes candidates j = lane + KV_LPQ c in chunks of K, sorts the
// chunk with a bitonic network and merges it into its running K-best; kv_row_merge folds the 16 lists.
// Distances follow the expansion formula of the arithmetic contract so the neighbour SET equals the
// CPU reference's; output order is (distance, index) ascending.
// ------------------------------------------------------------------------------------------------
// (body: workgroup `chunk` -- KV_QPW queries -- of sample `bs`; smem = 4 n floats with USE_LDS.  Component c of point j of a sample sits
// at base + j * ps + c * cs: (ps, cs) = (3, 1) for point-major (n, 3) clouds, (1, pitch) for the API's channel-major (3, n) ones.
// Shared by knn_point_kernel and geometry_front_kernel.)
template <int K, bool USE_LDS>
__device__ __forceinline__ void knn_point_body(int bs, int chunk, int s, int n, int k, const float *__restrict__ query, int q_ps, int q_cs,
                                               const float *__restrict__ points, int p_ps, int p_cs, int64_t *__restrict__ idx,
                                               const int *__restrict__ nvalid, float *smem) {
    const int tid = threadIdx.x;
    points += (size_t)bs * n * 3;
    if (nvalid) {     // padded batch: only the sample's own first nvalid[bs] points are candidates (the row pitch stays n)
        const int nv = nvalid[bs];
        n = nv < n ? (nv < k ? k : nv) : n;
    }
    float *sx = smem, *sy = smem + n, *sz = smem + 2 * n, *sn = smem + 3 * n;
    if (USE_LDS) {
        for (int j = tid; j < n; j += 256) {
            const float px = points[j * p_ps], py = points[j * p_ps + p_cs], pz = points[j * p_ps + 2 * p_cs];
            sx[j] = px; sy[j] = py; sz[j] = pz;
            sn[j] = __fadd_rn(__fadd_rn(__fmul_rn(px, px), __fmul_rn(py, py)), __fmul_rn(pz, pz));
        }
        __syncthreads();
    }
    const int li = tid & (KV_LPQ - 1);
    const int qi_raw = chunk * KV_QPW + tid / KV_LPQ;
    const int qi = qi_raw < s ? qi_raw : s - 1;
    const float *q = query + (size_t)bs * s * 3 + (size_t)qi * q_ps;
    const float qx = q[0], qy = q[q_cs], qz = q[2 * q_cs];
    const float qn = __fadd_rn(__fadd_rn(__fmul_rn(qx, qx), __fmul_rn(qy, qy)), __fmul_rn(qz, qz));
    unsigned bd[K];
    int bi[K];
#pragma unroll
    for (int a = 0; a < K; ++a) { bd[a] = KEY_INF_D; bi[a] = KEY_INF_I; }
    for (int base = 0; base < n; base += KV_LPQ * K) {
        unsigned cd[K];
        int ci[K];
#pragma unroll
        for (int a = 0; a < K; ++a) {
            const int j = base + li + KV_LPQ * a;
            cd[a] = KEY_INF_D; ci[a] = KEY_INF_I;
            if (j < n) {
                float px, py, pz, pn;
                if (USE_LDS) { px = sx[j]; py = sy[j]; pz = sz[j]; pn = sn[j]; }
                else {
                    px = points[j * p_ps]; py = points[j * p_ps + p_cs]; pz = points[j * p_ps + 2 * p_cs];
                    pn = __fadd_rn(__fadd_rn(__fmul_rn(px, px), __fmul_rn(py, py)), __fmul_rn(pz, pz));
                }
                const float dot = __fmaf_rn(qz, pz, __fmaf_rn(qy, py, __fmul_rn(qx, px)));
                float dd = __fadd_rn(__fadd_rn(__fmul_rn(-2.f, dot), qn), pn);
                dd = dd > 0.f ? dd : 0.f;  // torch.maximum(dist, 0)
                cd[a] = __float_as_uint(dd); ci[a] = j;
            }
        }
        kv_bitonic_sort<K>(cd, ci);
        kv_merge_low<K>(bd, bi, cd, ci);
    }
    kv_row_merge<K, KV_LPQ>(bd, bi);
    if (li == 0 && qi_raw < s) {
        int64_t *o = idx + ((size_t)bs * s + qi) * k;
#pragma unroll
        for (int a = 0; a < K; ++a)
            if (a < k) o[a] = bi[a];
    }
}

template <int K, bool USE_LDS>
__global__ __launch_bounds__(256) void knn_point_kernel(int s, int n, int k, const float *__restrict__ query,
                                                        const float *__restrict__ points, int64_t *__restrict__ idx,
                                                        const int *__restrict__ nvalid) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    knn_point_body<K, USE_LDS>(blockIdx.y, blockIdx.x, s, n, k, query, 3, 1, points, 3, 1, idx, nvalid, smem);
}

// Any k (the reference's torch.topk takes any k <= n; the live k is 16): one wave per query, k rounds of "smallest (distance, index)
// key above the last one taken" -- O(k n / 64) per query, no per-k register arrays.  Same distance arithmetic, same output order.
template <bool USE_LDS>
__global__ __launch_bounds__(256) void knn_point_select_kernel(int s, int n, int k, const float *__restrict__ query,
                                                               const float *__restrict__ points, int64_t *__restrict__ idx,
                                                               const int *__restrict__ nvalid) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int bs = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    points += (size_t)bs * n * 3;
    if (nvalid) {
        const int nv = nvalid[bs];
        n = nv < n ? (nv < k ? k : nv) : n;
    }
    float *sx = smem, *sy = smem + n, *sz = smem + 2 * n, *sn = smem + 3 * n;
    if (USE_LDS) {
        for (int j = tid; j < n; j += 256) {
            const float px = points[j * 3 + 0], py = points[j * 3 + 1], pz = points[j * 3 + 2];
            sx[j] = px; sy[j] = py; sz[j] = pz;
            sn[j] = __fadd_rn(__fadd_rn(__fmul_rn(px, px), __fmul_rn(py, py)), __fmul_rn(pz, pz));
        }
        __syncthreads();
    }
    const int qi = blockIdx.x * 4 + (tid >> 6);
    if (qi >= s) return;
    const float *q = query + ((size_t)bs * s + qi) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    const float qn = __fadd_rn(__fadd_rn(__fmul_rn(qx, qx), __fmul_rn(qy, qy)), __fmul_rn(qz, qz));
    int64_t *o = idx + ((size_t)bs * s + qi) * k;
    unsigned long long last = 0ull;
    for (int r = 0; r < k; ++r) {
        unsigned long long best = ~0ull;
        for (int j = lane; j < n; j += 64) {
            float px, py, pz, pn;
            if (USE_LDS) { px = sx[j]; py = sy[j]; pz = sz[j]; pn = sn[j]; }
            else {
                px = points[j * 3 + 0]; py = points[j * 3 + 1]; pz = points[j * 3 + 2];
                pn = __fadd_rn(__fadd_rn(__fmul_rn(px, px), __fmul_rn(py, py)), __fmul_rn(pz, pz));
            }
            const float dot = __fmaf_rn(qz, pz, __fmaf_rn(qy, py, __fmul_rn(qx, px)));
            float dd = __fadd_rn(__fadd_rn(__fmul_rn(-2.f, dot), qn), pn);
            dd = dd > 0.f ? dd : 0.f;
            const unsigned long long key = ((unsigned long long)__float_as_uint(dd) << 32) | (unsigned)j;      // dd >= 0: bit order = value order
            if ((r == 0 || key > last) && key < best) best = key;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const unsigned long long other = __shfl_xor(best, d, 64);
            best = other < best ? other : best;
        }
        last = best;
        if (lane == 0) o[r] = (int64_t)(best & 0xffffffffull);
    }
}

static int knn_point_impl(int b, int s, int n, int k, const float *query, const float *points, int64_t *idx, const int *nvalid,
                          rtk_stream_t stream) {
    RTK_REQUIRE(b > 0 && s > 0 && n > 0 && query && points && idx, "knn_point: bad arguments");
    RTK_REQUIRE(k >= 1 && k <= n, "knn_point: k=%d outside [1, n=%d]", k, n);
    RTK_REQUIRE(b <= 65535, "knn_point: b exceeds grid limits");
    dim3 grid(rtk_divup(s, KV_QPW), b);
    const size_t lds = (size_t)n * 4 * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    const bool use_lds = lds <= 64 * 1024;
    if (k > 32) {
        const dim3 g4(rtk_divup(s, 4), b);
        if (use_lds) knn_point_select_kernel<true><<<g4, 256, lds, st>>>(s, n, k, query, points, idx, nvalid);
        else knn_point_select_kernel<false><<<g4, 256, 0, st>>>(s, n, k, query, points, idx, nvalid);
    } else if (k <= 16) {
        if (use_lds) knn_point_kernel<16, true><<<grid, 256, lds, st>>>(s, n, k, query, points, idx, nvalid);
        else knn_point_kernel<16, false><<<grid, 256, 0, st>>>(s, n, k, query, points, idx, nvalid);
    } else {
        if (use_lds) knn_point_kernel<32, true><<<grid, 256, lds, st>>>(s, n, k, query, points, idx, nvalid);
        else knn_point_kernel<32, false><<<grid, 256, 0, st>>>(s, n, k, query, points, idx, nvalid);
    }
    RTK_CHECK_LAUNCH("knn_point");
    return RTK_OK;
}

extern "C" int rtk_knn_point(int b, int s, int n, int k, const float *query, const float *points, int64_t *idx,
                             rtk_stream_t stream) {
    return knn_point_impl(b, s, n, k, query, points, idx, nullptr, stream);
}

extern "C" int rtk_knn_point_masked(int b, int s, int n, int k, const float *query, const float *points, int64_t *idx,
                                    const int *n_valid, rtk_stream_t stream) {
    return knn_point_impl(b, s, n, k, query, points, idx, n_valid, stream);
}

// ------------------------------------------------------------------------------------------------
// The geometry of a batch in TWO launches (round 6; the same tables, bit for bit, as the eleven launches
// rtk_prepare_inputs + rtk_fps_centroids + 2 x rtk_fps_relevel + 3 x rtk_ball_query_pair + 3 x rtk_three_nn_masked +
// 2 x rtk_knn_point_masked they replace -- those stay as the ops' own entry points and as what tests compare these with).
//
// rtk_geometry_front -- everything that depends on the input clouds only.  A heterogeneous grid of 256-thread workgroups:
//   * workgroups [0, 2B): one per cloud.  All four waves convert the cloud's API tensors to the point-major layout the feature
//     kernels read (rtk_prepare_inputs); then wave 0 runs the furthest-point selection of level 1 straight from the API's
//     channel-major coordinates and goes on with levels 2 and 3 (rtk_fps_relevel's per-cloud decisions: copy / resume / settle) --
//     a level of a cloud depends on nothing but the previous level of the SAME cloud, so the three levels need no launch boundary,
//     only a fence between a level's stores and the next level's loads (same wave);
//   * workgroups [2B, 2B + 2 B ceil(n / KV_QPW)): the two kNN tables of the cost volume (frame 1 -> frame 2, frame 1 -> frame 1),
//     KV_QPW = 64 queries each, reading the API tensors as well.  They fill the chip while the 2B selection waves run their serial rounds.
// rtk_geometry_tables -- everything that needs the three levels of centroids: the three ball-query pair scans and the three
//   three-NN tables, one heterogeneous grid.
// ------------------------------------------------------------------------------------------------
struct GeoFrontParams {
    int B, S, n, npoint, block_n, block_np;      // B clouds in frame 1 (= kNN pairs), S clouds in all (B or 2B)
    const float *fr1, *fr2;      // the frames' clouds: component c of point k of sample b at fr[b * 3 n + k * ps + c * cs]
    int ps, cs;
    const float *f1, *f2;        // (B, 2, n) features (with xyz_out / raw_out)
    float *xyz_out, *raw_out;    // (2B, n, 3), (2B n, 4) or NULL
    const float *q1_w;           // (q1_cout, 2) or NULL: q1_out (2B n, q1_cout) = q1_w . (the two raw features of the point)
    float *q1_out;
    int q1_cout;
    int *fps_idx;                // (3, 2B, npoint)
    float *new_xyz;              // (3, 2B, npoint, 3)
    int *nuniq, *tie;            // (3, 2B) each
    int *first_tie;              // (2B)
    float *snap;                 // (2B, n)
    const int *n_valid;          // (2B) or NULL
    int64_t *knn12, *knn11;      // (B, n, 16) or NULL
    int knn_chunks;
};

template <int PPL1, int PPL2>
__global__ __launch_bounds__(256) void geometry_front_kernel(const GeoFrontParams P) {
    extern __shared__ __attribute__((aligned(16))) float4 s_geo[];
    const int S_ = P.S, n = P.n, m = P.npoint;
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= S_) {                      // ---- a kNN chunk
        const int r = (int)blockIdx.x - S_;
        const int per = P.B * P.knn_chunks;
        const int t = r / per, rem = r - t * per, b = rem / P.knn_chunks, chunk = rem - b * P.knn_chunks;
        const int *nv = P.n_valid ? (t == 0 ? P.n_valid + P.B : P.n_valid) : nullptr;
        knn_point_body<16, true>(b, chunk, n, n, 16, P.fr1, P.ps, P.cs, t == 0 ? P.fr2 : P.fr1, P.ps, P.cs, t == 0 ? P.knn12 : P.knn11, nv,
                                 reinterpret_cast<float *>(s_geo));
        return;
    }
    // ---- a cloud
    const int s = blockIdx.x, second = s >= P.B, sb = second ? s - P.B : s;
    const float *cloud = (second ? P.fr2 : P.fr1) + (size_t)sb * n * 3;
    if (P.xyz_out) {
        const float *f = (second ? P.f2 : P.f1) + (size_t)sb * 2 * n;
        for (int p = tid; p < n; p += 256) {
            const size_t t = (size_t)s * n + p;
            P.xyz_out[t * 3 + 0] = cloud[p * P.ps];
            P.xyz_out[t * 3 + 1] = cloud[p * P.ps + P.cs];
            P.xyz_out[t * 3 + 2] = cloud[p * P.ps + 2 * P.cs];
            const float fa = f[p], fb = f[n + p];
            *reinterpret_cast<float4 *>(P.raw_out + t * 4) = make_float4(fa, fb, 0.f, 0.f);
            if (P.q1_w) {      // the encoder's sa1 layer-1 projection of the raw features (both scales side by side): 2 -> q1_cout, no bias
                float *qo = P.q1_out + t * P.q1_cout;
                for (int c = 0; c < P.q1_cout; c += 4) {
                    float4 o;
                    o.x = __fmaf_rn(P.q1_w[2 * c + 1], fb, __fmul_rn(P.q1_w[2 * c], fa));
                    o.y = __fmaf_rn(P.q1_w[2 * c + 3], fb, __fmul_rn(P.q1_w[2 * c + 2], fa));
                    o.z = __fmaf_rn(P.q1_w[2 * c + 5], fb, __fmul_rn(P.q1_w[2 * c + 4], fa));
                    o.w = __fmaf_rn(P.q1_w[2 * c + 7], fb, __fmul_rn(P.q1_w[2 * c + 6], fa));
                    *reinterpret_cast<float4 *>(qo + c) = o;
                }
            }
        }
    }
    if (tid >= 64) return;                             // (the selection below is one wave's work; it contains no workgroup barrier)
    const int lane = tid;
    const size_t lv = (size_t)S_ * m;                  // one level of indices
    int *i1 = P.fps_idx + (size_t)s * m, *i2 = i1 + lv, *i3 = i2 + lv;
    float *x1 = P.new_xyz + (size_t)s * m * 3, *x2 = x1 + lv * 3, *x3 = x2 + lv * 3;
    float *snap = P.snap + (size_t)s * n;
    int tie1 = 0, tie2 = 0, tie3 = 0, ft1 = 0;
    const int nu1 = fps_cloud_level<PPL1>(n, m, P.block_n, cloud, P.ps, P.cs, nullptr, i1, x1, s_geo, lane, tie1, P.n_valid ? P.n_valid[s] : -1, snap,
                                          P.first_tie + s, -1, 0, 1, nullptr, nullptr, nullptr, &ft1);
    __threadfence();                                   // level 1's centroids, indices and saved state are read back below
    const int j_start = (ft1 > 1 && ft1 < m) ? ft1 : 1;
    const int nu2 = fps_cloud_level<PPL2>(m, m, P.block_np, x1, 3, 1, nullptr, i2, x2, s_geo, lane, tie2, -1, nullptr, nullptr, tie1, nu1, j_start, snap, i1,
                                          nullptr);
    __threadfence();
    const int nu3 = fps_cloud_level<PPL2>(m, m, P.block_np, x2, 3, 1, nullptr, i3, x3, s_geo, lane, tie3, -1, nullptr, nullptr, tie2, nu2, j_start, snap, i1,
                                          i2);
    if (lane == 0) {
        P.first_tie[s] = ft1;                          // (0: none -- written unconditionally: the workspace needs no zero fill)
        P.nuniq[s] = nu1; P.nuniq[S_ + s] = nu2; P.nuniq[2 * S_ + s] = nu3;
        P.tie[s] = tie1; P.tie[S_ + s] = tie2; P.tie[2 * S_ + s] = tie3;
    }
}

extern "C" int rtk_geometry_front(int b, int clouds, int n, int npoint, const float *frame1, const float *frame2, int channel_major,
                                  const float *feature1, const float *feature2, float *xyz, float *raw, int *fps_idx, float *new_xyz,
                                  int *nuniq, int *tie, int *first_tie, float *snap, const int *n_valid, int64_t *knn12, int64_t *knn11,
                                  const float *q1_w, float *q1_out, int q1_cout, rtk_stream_t stream) {
    RTK_REQUIRE(b > 0 && n > 0 && npoint > 0 && frame1 && fps_idx && new_xyz && nuniq && tie && first_tie && snap,
                "geometry_front: bad arguments (b=%d n=%d npoint=%d)", b, n, npoint);
    RTK_REQUIRE(clouds == b || (clouds == 2 * b && frame2), "geometry_front: clouds=%d must be b or 2 b (b=%d) with both frames", clouds, b);
    RTK_REQUIRE(n <= 2048 && npoint <= 512, "geometry_front: n=%d > 2048 or npoint=%d > 512 (use the separate entry points)", n, npoint);
    RTK_REQUIRE((xyz == nullptr) == (raw == nullptr) && (!xyz || (feature1 && feature2)), "geometry_front: xyz, raw and the features go together");
    RTK_REQUIRE((knn12 == nullptr) == (knn11 == nullptr) && (!knn12 || (n >= 16 && clouds == 2 * b)),
                "geometry_front: the two kNN tables go together (n >= 16, both frames)");
    RTK_REQUIRE((q1_w == nullptr) == (q1_out == nullptr) && (!q1_w || (xyz && q1_cout > 0 && q1_cout % 4 == 0)),
                "geometry_front: q1_w, q1_out go together (with the layout conversion; q1_cout a multiple of 4)");
    GeoFrontParams P;
    P.q1_w = q1_w; P.q1_out = q1_out; P.q1_cout = q1_cout;
    P.B = b; P.S = clouds; P.n = n; P.npoint = npoint; P.block_n = fps_block_size(n); P.block_np = fps_block_size(npoint);
    P.fr1 = frame1; P.fr2 = frame2; P.ps = channel_major ? 1 : 3; P.cs = channel_major ? n : 1;
    P.f1 = feature1; P.f2 = feature2; P.xyz_out = xyz; P.raw_out = raw;
    P.fps_idx = fps_idx; P.new_xyz = new_xyz; P.nuniq = nuniq; P.tie = tie; P.first_tie = first_tie; P.snap = snap; P.n_valid = n_valid;
    P.knn12 = knn12; P.knn11 = knn11; P.knn_chunks = rtk_divup(n, KV_QPW);
    const int grid = clouds + (knn12 ? 2 * b * P.knn_chunks : 0);
    const size_t lds = (size_t)(n > npoint ? n : npoint) * sizeof(float4);
    hipStream_t s = (hipStream_t)stream;
    if (n <= 64 * 4) geometry_front_kernel<4, 8><<<grid, 256, lds, s>>>(P);
    else if (n <= 64 * 8) geometry_front_kernel<8, 8><<<grid, 256, lds, s>>>(P);
    else if (n <= 64 * 16) geometry_front_kernel<16, 8><<<grid, 256, lds, s>>>(P);
    else geometry_front_kernel<32, 8><<<grid, 256, lds, s>>>(P);
    RTK_CHECK_LAUNCH("geometry_front");
    return RTK_OK;
}

#ifndef GEO_TABLES_WGS
#define GEO_TABLES_WGS 2048      // workgroups of rtk_geometry_tables (about eight per CU: 1024 / 4096 measured within 1 %)
#endif
struct GeoBallTask {
    int n_src, nsa, nsb;
    float r2a, r2b;
    const float *new_xyz, *xyz;
    int *idxa, *idxb;
    const int *nuniq, *src_nuniq;
};
struct GeoNnTask {
    int n, m, wgs;               // wgs: workgroups per sample
    const float *unknown, *known;
    float *dist2;
    int *idx;
    const int *unknown_nuniq, *known_nuniq;
};
struct GeoTablesParams {
    int samples, npoint, ball_wgs;      // ball_wgs: workgroups per (level, sample)
    GeoBallTask ball[3];
    GeoNnTask nn[3];
};

// Workgroup w of a (task, sample) takes the task's chunks w, w + W, w + 2 W, ... (interleaved: the chunks beyond the unique centroids,
// which cost nothing, are spread evenly).
__global__ __launch_bounds__(256) void geometry_tables_kernel(const GeoTablesParams P) {
    extern __shared__ __attribute__((aligned(16))) float s_tab[];
    int r = blockIdx.x;
    const int per_ball = P.samples * P.ball_wgs;
    if (r < 3 * per_ball) {
        const int t = r / per_ball, rem = r - t * per_ball, bs = rem / P.ball_wgs, w = rem - bs * P.ball_wgs;
        const GeoBallTask &T = P.ball[t];
        ball_query_pair_body(bs, w, P.ball_wgs, T.n_src, P.npoint, T.r2a, T.nsa, T.r2b, T.nsb, T.new_xyz, T.xyz, T.idxa, T.idxb, T.nuniq, s_tab, T.src_nuniq);
        return;
    }
    r -= 3 * per_ball;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const GeoNnTask &T = P.nn[t];
        const int per = P.samples * T.wgs;
        if (r < per) {
            const int bs = r / T.wgs, w = r - bs * T.wgs;
            three_nn_body<true>(bs, w, T.wgs, T.n, T.m, T.unknown, T.known, T.dist2, T.idx, T.unknown_nuniq, T.known_nuniq, s_tab);
            return;
        }
        r -= per;
    }
}

// xyz0 (S, n, 3): the clouds; new_xyz (3, S, npoint, 3), nuniq (3, S): the three levels of rtk_geometry_front.
// radii / nsamples: (3 levels x 2 scales), radii[2 l] <= radii[2 l + 1].  ball[2 l + s] (S, npoint, nsamples[2 l + s]) int32, zero-initialised
// by the caller (rtk_ball_query's contract).  nn_idx / nn_dist2 [fp3, fp2, fp1]: (S, npoint, 3), (S, npoint, 3), (S, n, 3).
extern "C" int rtk_geometry_tables(int samples, int n, int npoint, const float *xyz0, const float *new_xyz, const int *nuniq,
                                   const float *radii, const int *nsamples, int *const *ball, int *const *nn_idx, float *const *nn_dist2,
                                   rtk_stream_t stream) {
    RTK_REQUIRE(samples > 0 && n > 0 && npoint > 0 && xyz0 && new_xyz && nuniq && radii && nsamples && ball && nn_idx && nn_dist2,
                "geometry_tables: bad arguments");
    const int big = n > npoint ? n : npoint;
    RTK_REQUIRE((size_t)big * 12 <= 64 * 1024, "geometry_tables: clouds of %d points do not fit the LDS stage", big);
    GeoTablesParams P;
    // workgroups per (task, sample): enough of them to fill the chip (~2 k in all over the nine tasks), never more than the task has chunks
    const int want = rtk_divup(GEO_TABLES_WGS, 9 * samples);
    const int ball_chunks = rtk_divup(npoint, BQ_WAVES * BQ_CENTROIDS_PER_WAVE);
    P.samples = samples; P.npoint = npoint; P.ball_wgs = want < ball_chunks ? want : ball_chunks;
    const size_t lv = (size_t)samples * npoint * 3;
    const float *lvl_xyz[4] = {xyz0, new_xyz, new_xyz + lv, new_xyz + 2 * lv};
    const int lvl_n[4] = {n, npoint, npoint, npoint};
    for (int l = 0; l < 3; ++l) {
        RTK_REQUIRE(radii[2 * l] <= radii[2 * l + 1] && nsamples[2 * l] > 0 && nsamples[2 * l + 1] > 0 && ball[2 * l] && ball[2 * l + 1],
                    "geometry_tables: level %d: radii must ascend, tables must be given", l);
        GeoBallTask &T = P.ball[l];
        T.n_src = lvl_n[l]; T.nsa = nsamples[2 * l]; T.nsb = nsamples[2 * l + 1];
        T.r2a = radii[2 * l] * radii[2 * l]; T.r2b = radii[2 * l + 1] * radii[2 * l + 1];      // fp32 products, as ball_query_gpu.cu:23
        T.new_xyz = lvl_xyz[l + 1]; T.xyz = lvl_xyz[l]; T.idxa = ball[2 * l]; T.idxb = ball[2 * l + 1]; T.nuniq = nuniq + (size_t)l * samples;
        T.src_nuniq = l > 0 ? nuniq + (size_t)(l - 1) * samples : nullptr;
    }
    // fp3: unknown level 2, known level 3; fp2: unknown level 1, known level 2; fp1: unknown level 0 (the points), known level 1
    const int uk[3][2] = {{2, 3}, {1, 2}, {0, 1}};
    int total = 3 * samples * P.ball_wgs;
    for (int i = 0; i < 3; ++i) {
        const int u = uk[i][0], k = uk[i][1];
        RTK_REQUIRE(nn_idx[i] && nn_dist2[i], "geometry_tables: three-NN table %d missing", i);
        GeoNnTask &T = P.nn[i];
        T.n = lvl_n[u]; T.m = lvl_n[k]; T.wgs = want < rtk_divup(T.n, KV_QPW) ? want : rtk_divup(T.n, KV_QPW);
        T.unknown = lvl_xyz[u]; T.known = lvl_xyz[k]; T.dist2 = nn_dist2[i]; T.idx = nn_idx[i];
        T.unknown_nuniq = u > 0 ? nuniq + (size_t)(u - 1) * samples : nullptr;
        T.known_nuniq = nuniq + (size_t)(k - 1) * samples;
        total += samples * T.wgs;
    }
    geometry_tables_kernel<<<total, 256, (size_t)big * 12, (hipStream_t)stream>>>(P);
    RTK_CHECK_LAUNCH("geometry_tables");
    return RTK_OK;
}
