// Backward of the first layer of a set-abstraction SharedMLP (include/rtk_train.h):
//     z1[s][c][row][k] = proj[s][c][idx[s][row][k]] + wx[c] . dxyz[s][:, row, k]          (rtk_sa_first_layer)
//     dproj[s][c][q]   = sum over the positions p = (row, k) with idx[s][p] == q of dz1[s][c][p]
//     dwx[c][a]        = sum over samples and positions of dz1[s][c][p] dxyz[s][a][p]
// The reference scatters with one atomicAdd per element (group_points_gpu.cu:8-25) and gets dwx from the framework's
// convolution backward; round 1 scattered through LDS float atomics (ds_add_f32 sustains well under one lane per clock: 1 ms
// per step, LDS-atomic bound) and ran a batched GEMM + a reduction over dz1 for dwx (a second full read of dz1).
//
// Here the scatter is turned into a GATHER.  rtk_group_inverse_index sorts every sample's positions by the source point they
// reference -- once per (level, scale) and step, the table does not depend on the features -- and rtk_sa_first_layer_bwd
// streams each dz1 plane through LDS exactly once: while a plane is being staged every thread multiplies its elements with
// the (register-resident) offsets for dwx, then every thread sums an equal share of the sorted positions (runs of one source
// point in registers, one LDS add per run end).  One read of dz1, a few hundred LDS adds per plane instead of one per element.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdlib.h>

#include "rtk_common.h"
#include "rtk_train.h"

namespace {

// off (samples, n_src + 1) int32: positions referencing source point q are inv[s][off[q] .. off[q+1]), ascending.
// II_T threads: at 256 a thread runs a few thousand serial LDS operations and there is only one workgroup per sample
constexpr int II_T = 1024;
// live (optional): only the first live[s] * live_mult positions of sample s enter the table.  For rows that nothing downstream reads
// (the duplicate centroid rows of a de-duplicated level: every consumer redirects them to a live row) the gradient is exactly zero,
// and left in, their positions only lengthen the lists the gather kernels walk.  NOT for the padding rows of level 0: those are
// referenced directly by the ball tables and do carry gradient.
__device__ __forceinline__ void inverse_index_body(int n_src, int Pall, const int *__restrict__ idx, int *__restrict__ off,
                                                   unsigned short *__restrict__ inv, int s, const int *__restrict__ live, int live_mult) {
    extern __shared__ int s_cnt[];                 // [n_src + 1] counts -> offsets, [n_src] cursors
    int *s_cur = s_cnt + n_src + 1;
    const int t = threadIdx.x;
    const int *id = idx + (size_t)s * Pall;
    const int P = live ? min(Pall, max(0, live[s]) * live_mult) : Pall;      // positions that enter the table
    for (int q = t; q <= n_src; q += II_T) s_cnt[q] = 0;
    __syncthreads();
    for (int p = t; p < P; p += II_T) atomicAdd(&s_cnt[id[p]], 1);
    __syncthreads();
    if (t < 64) {                                  // exclusive scan by one wave
        int carry = 0;
        for (int base = 0; base <= n_src; base += 64) {
            const int q = base + t;
            const int v = q <= n_src ? s_cnt[q] : 0;
            int inc = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int u = __shfl_up(inc, o, 64);
                if (t >= o) inc += u;
            }
            if (q <= n_src) s_cnt[q] = carry + inc - v;
            carry += __shfl(inc, 63, 64);
        }
    }
    __syncthreads();
    for (int q = t; q < n_src; q += II_T) s_cur[q] = s_cnt[q];
    for (int q = t; q <= n_src; q += II_T) off[(size_t)s * (n_src + 1) + q] = s_cnt[q];
    __syncthreads();
    unsigned short *iv = inv + (size_t)s * Pall;
    unsigned short *s_tmp = reinterpret_cast<unsigned short *>(s_cur + n_src);      // [P] the lists as the atomic cursors filled them
    for (int p = t; p < P; p += II_T) s_tmp[atomicAdd(&s_cur[id[p]], 1)] = (unsigned short)p;
    __syncthreads();
    // the cursors filled every list in arbitrary order: every element finds its rank inside its own list (short lists, all
    // elements in parallel) so that the table -- and every sum taken in its order -- is reproducible
    for (int e = t; e < P; e += II_T) {
        const unsigned short v = s_tmp[e];
        const int q = id[v], a = s_cnt[q], b = s_cnt[q + 1];
        int r = 0;
        for (int i = a; i < b; ++i) r += s_tmp[i] < v ? 1 : 0;
        iv[a + r] = v;
    }
}

__global__ __launch_bounds__(II_T) void inverse_index_kernel(int n_src, int P, const int *__restrict__ idx, int *__restrict__ off,
                                                            unsigned short *__restrict__ inv) {
    inverse_index_body(n_src, P, idx, off, inv, blockIdx.x, nullptr, 1);
}

// Several tables in one launch (blockIdx.y = table): a table is one workgroup per cloud walking serial phases, so at small batches
// ten tables in ten launches are ten times the latency of one launch with ten times the workgroups (B = 1: 0.6 of a 2.9 ms step).
constexpr int II_MAX_JOBS = 12;
struct IIJobs {
    int n;
    rtk_inverse_index_job_t j[II_MAX_JOBS];
};
__global__ __launch_bounds__(II_T) void inverse_index_multi_kernel(const IIJobs J) {
    const rtk_inverse_index_job_t &q = J.j[blockIdx.y];
    inverse_index_body(q.n_src, q.positions, q.idx, q.off, q.inv, blockIdx.x, q.live, q.live_mult);
}

// Order-independent segmented sums: the threads' run partials (floats, each summed in a fixed order) are added to the source point's
// accumulator as 64-bit FIXED POINT with LDS integer atomics -- a run that straddles several threads' chunks gets the same sum
// whatever order its pieces arrive in (float atomics rounded after every piece: the gradients of a step differed from run to run
// in the last bits).  The unit is set per plane from its largest |dz|: 2^-k with k = 61 - ceil(log2 P) - exponent(max), so that the
// P values of a plane cannot overflow 63 bits; what a piece loses is below 2^-(60 - log2 P) of the plane's largest element (2^-47 at
// 8192 positions).  A plane with a non-finite element gives NaN sums.
struct FxPlane {
    double scale, inv;      // 2^k, 2^-k
    bool bad;
};
__device__ __forceinline__ FxPlane fx_plane(float amax, bool bad, int pshift) {
    const int e = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 127;      // amax >= 0 and finite unless bad
    const int k = pshift - e;
    FxPlane f;
    f.scale = __hiloint2double((1023 + k) << 20, 0);
    f.inv = __hiloint2double((1023 - k) << 20, 0);
    f.bad = bad;
    return f;
}
__device__ __forceinline__ void fx_add(long long *acc, float v, const FxPlane &f) {
    atomicAdd(reinterpret_cast<unsigned long long *>(acc), (unsigned long long)__double2ll_rn((double)v * f.scale));
}
__device__ __forceinline__ float fx_value(long long q, const FxPlane &f) {
    return f.bad ? __uint_as_float(0x7fc00000u) : (float)((double)q * f.inv);
}
__device__ __forceinline__ int fx_pshift(int P) {
    int lg = 0;
    while ((1 << lg) < P) ++lg;
    return 61 - lg;
}
__device__ __forceinline__ float fx_amax4(float am, const float4 v, unsigned &bad) {
    bad |= (unsigned)(!(fabsf(v.x) <= 3.0e38f)) | (unsigned)(!(fabsf(v.y) <= 3.0e38f)) | (unsigned)(!(fabsf(v.z) <= 3.0e38f)) |
           (unsigned)(!(fabsf(v.w) <= 3.0e38f));      // infinity or NaN
    return fmaxf(am, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
}
// wave partials of the plane's (dwx_x, dwx_y, dwx_z, max |dz|, non-finite flag) -> LDS; after the next barrier every thread combines them
__device__ __forceinline__ void fx_wave_partials(float (&s_red)[4][5], int wave, int lane, float ax, float ay, float az, float am, unsigned bad) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ax += __shfl_xor(ax, o, 64); ay += __shfl_xor(ay, o, 64); az += __shfl_xor(az, o, 64);
        am = fmaxf(am, __shfl_xor(am, o, 64));
    }
    const bool anybad = __ballot(bad != 0u) != 0ull;
    if (lane == 0) { s_red[wave][0] = ax; s_red[wave][1] = ay; s_red[wave][2] = az; s_red[wave][3] = am; s_red[wave][4] = anybad ? 1.f : 0.f; }
}

// dwx[c][t] += the samples' shares (samples, channels, 3) in a fixed order, no atomics: one wave per element -- lane l adds samples
// l, l + 64, ... (all loads in flight at once), then the xor tree over the lanes.  (A second launch: letting the last workgroup of the
// main kernel to arrive do it kept every such launch waiting for one workgroup's 24 k loads -- train step 7.6 -> 7.9 ms.)
__global__ __launch_bounds__(256) void dwx_reduce_kernel(int samples, int channels, const float *__restrict__ ws, float *__restrict__ dwx, int dwx_pitch) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= channels * 3) return;
    float a = 0.f;
    for (int s = lane; s < samples; s += 64) a += ws[(size_t)s * channels * 3 + e];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if (lane == 0) dwx[(size_t)(e / 3) * dwx_pitch + e % 3] += a;
}

constexpr int FB_MAXQ = 8;      // float4 per thread and plane: planes up to 8192 positions keep their offsets in registers

__global__ __launch_bounds__(256) void sa_first_layer_bwd_kernel(int channels, int cg, int n_src, int P, const float *__restrict__ dz,
                                                                 const float *__restrict__ dxyz, const int *__restrict__ off,
                                                                 const unsigned short *__restrict__ inv, float *__restrict__ dproj,
                                                                 float *__restrict__ dwx_ws, float *__restrict__ dwx, int dwx_pitch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fb_smem[];
    float *s_plane = reinterpret_cast<float *>(fb_smem);                                  // [P + 4]
    long long *s_out = reinterpret_cast<long long *>(s_plane + P + 4);                    // [n_src] fixed point (P % 4 == 0: 16-byte aligned)
    int *s_off = reinterpret_cast<int *>(s_out + n_src);                                  // [n_src + 1]
    unsigned short *s_inv = reinterpret_cast<unsigned short *>(s_off + n_src + 1);        // [P + 256]
    __shared__ float s_red[4][5];
    const int pshift = fx_pshift(P);
    const int s = blockIdx.y, c0 = blockIdx.x * cg, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int q = t; q <= n_src; q += 256) s_off[q] = off[(size_t)s * (n_src + 1) + q];
    // thread t later walks the sorted positions [t E, (t+1) E): one pad element per chunk keeps the 64 lanes of a wave on
    // different LDS banks (a plain layout puts them 2E bytes apart: a 32-way conflict per read)
    const int E = (P + 255) >> 8;
    for (int p = t; p < P; p += 256) s_inv[p + p / E] = inv[(size_t)s * P + p];
    const int n4 = P >> 2;                        // P % 4 == 0 (ns >= 4)
    const float4 *dx4 = reinterpret_cast<const float4 *>(dxyz + (size_t)s * 3 * P);
    float4 ox[FB_MAXQ], oy[FB_MAXQ], oz[FB_MAXQ];
    const bool in_regs = n4 <= 256 * FB_MAXQ;
    if (in_regs) {
#pragma unroll
        for (int i = 0; i < FB_MAXQ; ++i) {
            const int e = t + 256 * i;
            const int ec = e < n4 ? e : n4 - 1;      // unconditional (clamped) loads; tail threads never use theirs
            ox[i] = dx4[ec]; oy[i] = dx4[n4 + ec]; oz[i] = dx4[2 * n4 + ec];
        }
    }
    // position of this thread's chunk of the sorted positions in the run structure (the same for every plane)
    const int e0 = t * E, e1 = min(P, e0 + E);
    int q_first = 0;
    __syncthreads();                               // s_off is complete
    if (e0 < e1) {
        int lo = 0, hi = n_src;                    // the source point of position e0: last q with off[q] <= e0
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_off[mid] <= e0) lo = mid; else hi = mid; }
        q_first = lo;
    }
    const int nplanes = min(cg, channels - c0);
    float4 v[FB_MAXQ];
    auto request = [&](int c) {                    // all of a plane's loads in flight at once (tail threads re-read its last float4)
        const float4 *pl = reinterpret_cast<const float4 *>(dz + ((size_t)s * channels + c) * P);
#pragma unroll
        for (int i = 0; i < FB_MAXQ; ++i) {
            const int e = t + 256 * i;
            v[i] = pl[e < n4 ? e : n4 - 1];
        }
    };
    if (in_regs) request(c0);
    for (int cc = 0; cc < nplanes; ++cc) {
        const int c = c0 + cc;
        const float4 *pl = reinterpret_cast<const float4 *>(dz + ((size_t)s * channels + c) * P);
        float ax = 0.f, ay = 0.f, az = 0.f, am = 0.f;
        unsigned bad = 0u;
        __syncthreads();                           // the previous plane has been consumed
        if (in_regs) {
#pragma unroll
            for (int i = 0; i < FB_MAXQ; ++i) {
                const int e = t + 256 * i;
                if (e < n4) {
                    reinterpret_cast<float4 *>(s_plane)[e] = v[i];
                    am = fx_amax4(am, v[i], bad);
                    ax += (v[i].x * ox[i].x + v[i].y * ox[i].y) + (v[i].z * ox[i].z + v[i].w * ox[i].w);
                    ay += (v[i].x * oy[i].x + v[i].y * oy[i].y) + (v[i].z * oy[i].z + v[i].w * oy[i].w);
                    az += (v[i].x * oz[i].x + v[i].y * oz[i].y) + (v[i].z * oz[i].z + v[i].w * oz[i].w);
                }
            }
            if (cc + 1 < nplanes) request(c + 1);  // the next plane travels while this one is gathered
        } else {
            for (int e = t; e < n4; e += 256) {
                const float4 v = pl[e], a = dx4[e], b = dx4[n4 + e], d = dx4[2 * n4 + e];
                reinterpret_cast<float4 *>(s_plane)[e] = v;
                am = fx_amax4(am, v, bad);
                ax += (v.x * a.x + v.y * a.y) + (v.z * a.z + v.w * a.w);
                ay += (v.x * b.x + v.y * b.y) + (v.z * b.z + v.w * b.w);
                az += (v.x * d.x + v.y * d.y) + (v.z * d.z + v.w * d.w);
            }
        }
        fx_wave_partials(s_red, wave, lane, ax, ay, az, am, bad);
        for (int q = t; q < n_src; q += 256) s_out[q] = 0;
        __syncthreads();                           // plane staged, partials visible, accumulator clear
        // this sample's share of dwx[c]: summed over the samples in a fixed order by dwx_reduce_kernel, or (no workspace) added with a
        // float atomic, in the order of arrival
        if (t < 3) {
            const float share = (s_red[0][t] + s_red[1][t]) + (s_red[2][t] + s_red[3][t]);
            if (dwx_ws) dwx_ws[((size_t)s * channels + c) * 3 + t] = share;
            else atomicAdd(dwx + (size_t)c * dwx_pitch + t, share);
        }
        const FxPlane fx = fx_plane(fmaxf(fmaxf(s_red[0][3], s_red[1][3]), fmaxf(s_red[2][3], s_red[3][3])),
                                    (s_red[0][4] + s_red[1][4]) + (s_red[2][4] + s_red[3][4]) != 0.f, pshift);
        // balanced segmented sum: thread t owns the sorted positions [t E, (t+1) E); runs of one source point inside the chunk
        // are summed in registers, only the (few) run ends go to the LDS accumulator
        {
            if (e0 < e1) {
                int q = q_first, nb = s_off[q + 1];
                float acc = 0.f;
                // eight gathered values at a time: the two dependent LDS reads (inverse index, then plane) of a batch are in flight
                // together, the run bookkeeping is register work.  (Measured and rejected: 1024 threads per workgroup to overlap more
                // LDS round trips -- 40 % slower, the per-plane barriers and chunk-boundary atomics grow with the thread count.)
                for (int eb = e0; eb < e1; eb += 8) {
                    float val[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) val[k] = s_plane[s_inv[min(eb + k, e1 - 1) + t]];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int e = eb + k;
                        if (e < e1) {
                            if (e >= nb) {
                                fx_add(&s_out[q], acc, fx);
                                acc = 0.f;
                                do { ++q; nb = s_off[q + 1]; } while (e >= nb);
                            }
                            acc += val[k];
                        }
                    }
                }
                fx_add(&s_out[q], acc, fx);
            }
        }
        __syncthreads();
        float *out = dproj + ((size_t)s * channels + c) * n_src;
        for (int q = t; q < n_src; q += 256) out[q] = fx_value(s_out[q], fx);
    }
}

// The same for planes of at most 8192 positions (every level of the N = 256 configurations), with the run structure -- which sorted
// positions a thread owns, which source point each belongs to, where its runs end -- worked out ONCE per workgroup into registers
// instead of once per plane inside the summation loop.  The generic kernel above re-walks the offset table while it sums: with runs
// of ~32 positions and 64 lanes, some lane ends a run at almost every step and the whole wave takes the divergent
// atomic-and-advance path (~300 cycles) 32 times per plane: 20 k cycles per plane, 1.5 TB/s at the largest shape.  Here a plane is
// EMAX independent LDS reads, EMAX adds and a predicated ds_add_f32 at the precomputed run ends.
// iq[k] = (source point << 16) | position of the thread's k-th sorted element; elements past the chunk point at a zero slot.
template <int EMAX>
__global__ __launch_bounds__(256, 2) void sa_first_layer_bwd_fast_kernel(int channels, int cg, int gx, int n_src, int P, const float *__restrict__ dz,
                                                                      const float *__restrict__ dxyz, const int *__restrict__ off,
                                                                      const unsigned short *__restrict__ inv, float *__restrict__ dproj,
                                                                      float *__restrict__ dwx_ws, float *__restrict__ dwx, int dwx_pitch) {
    constexpr int NQ = EMAX / 4;      // float4 per thread and plane (P <= 256 EMAX)
    extern __shared__ __attribute__((aligned(16))) unsigned char fb_smem[];
    float *s_plane = reinterpret_cast<float *>(fb_smem);                                  // [P + 4]: s_plane[P] = 0 (the slot of absent elements)
    long long *s_out = reinterpret_cast<long long *>(s_plane + P + 4);                    // [n_src] fixed point (see fx_plane)
    int *s_off = reinterpret_cast<int *>(s_out + n_src);                                  // [n_src + 1]
    unsigned short *s_inv = reinterpret_cast<unsigned short *>(s_off + n_src + 1);        // [P + 256] (padded, see the generic kernel)
    __shared__ float s_red[4][5];
    const int pshift = fx_pshift(P);
    // gx > 0: 1-D grid decoded so that all workgroups of sample s run on XCD s % 8 (workgroup ids go round-robin over the XCDs, each
    // with its own L2): the sample's index table and offset planes (16 + 96 KiB at the largest shape) are fetched from HBM once, not
    // once per XCD that happens to get one of the sample's channel groups
    int s, bx;
    if (gx > 0) {
        const int L = blockIdx.x, slot = L >> 3;
        s = (slot / gx) * 8 + (L & 7);
        bx = slot - (slot / gx) * gx;
    } else {
        s = blockIdx.y;
        bx = blockIdx.x;
    }
    const int c0 = bx * cg, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int q = t; q <= n_src; q += 256) s_off[q] = off[(size_t)s * (n_src + 1) + q];
    const int E = (P + 255) >> 8;                 // <= EMAX
    for (int p = t; p < P; p += 256) s_inv[p + p / E] = inv[(size_t)s * P + p];
    if (t < 4) s_plane[P + t] = 0.f;
    const int n4 = P >> 2;                        // P % 4 == 0 (ns >= 4), n4 <= 256 NQ
    const float4 *dx4 = reinterpret_cast<const float4 *>(dxyz + (size_t)s * 3 * P);
    float4 ox[NQ], oy[NQ], oz[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int e = t + 256 * i;
        const int ec = e < n4 ? e : n4 - 1;      // unconditional (clamped) loads; tail threads never use theirs
        ox[i] = dx4[ec]; oy[i] = dx4[n4 + ec]; oz[i] = dx4[2 * n4 + ec];
    }
    const int nplanes = min(cg, channels - c0);
    float4 v[NQ];
    auto request = [&](int c) {                    // all of a plane's loads in flight at once (tail threads re-read its last float4)
        const float4 *pl = reinterpret_cast<const float4 *>(dz + ((size_t)s * channels + c) * P);
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int e = t + 256 * i;
            v[i] = pl[e < n4 ? e : n4 - 1];
        }
    };
    request(c0);
    // ---- the run structure of this thread's chunk [t E, (t + 1) E) of the sorted positions, once ----------------------------------
    const int e0 = t * E, cnt = max(0, min(P, e0 + E) - e0);
    unsigned iq[EMAX];
    unsigned endmask = 0;
    __syncthreads();                               // s_off, s_inv complete
    {
        int q = 0, nb = 0;
        if (cnt > 0) {
            int lo = 0, hi = n_src;                // the source point of position e0: last q with off[q] <= e0
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_off[mid] <= e0) lo = mid; else hi = mid; }
            q = lo;
            nb = s_off[q + 1];
        }
#pragma unroll
        for (int k = 0; k < EMAX; ++k) {
            const int e = e0 + k;
            if (k < cnt) {
                while (e >= nb) { ++q; nb = s_off[q + 1]; }
                iq[k] = (unsigned)s_inv[e + t] | ((unsigned)q << 16);
                if (k == cnt - 1 || e + 1 >= nb) endmask |= 1u << k;
            } else {
                iq[k] = (unsigned)P;               // the zero slot
            }
        }
    }
    for (int cc = 0; cc < nplanes; ++cc) {
        const int c = c0 + cc;
        float ax = 0.f, ay = 0.f, az = 0.f, am = 0.f;
        unsigned bad = 0u;
        __syncthreads();                           // the previous plane has been consumed
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int e = t + 256 * i;
            if (e < n4) {
                reinterpret_cast<float4 *>(s_plane)[e] = v[i];
                am = fx_amax4(am, v[i], bad);
                ax += (v[i].x * ox[i].x + v[i].y * ox[i].y) + (v[i].z * ox[i].z + v[i].w * ox[i].w);
                ay += (v[i].x * oy[i].x + v[i].y * oy[i].y) + (v[i].z * oy[i].z + v[i].w * oy[i].w);
                az += (v[i].x * oz[i].x + v[i].y * oz[i].y) + (v[i].z * oz[i].z + v[i].w * oz[i].w);
            }
        }
        if (cc + 1 < nplanes) request(c + 1);      // the next plane travels while this one is gathered
        fx_wave_partials(s_red, wave, lane, ax, ay, az, am, bad);
        for (int q = t; q < n_src; q += 256) s_out[q] = 0;
        __syncthreads();                           // plane staged, partials visible, accumulator clear
        if (t < 3) {
            const float share = (s_red[0][t] + s_red[1][t]) + (s_red[2][t] + s_red[3][t]);
            if (dwx_ws) dwx_ws[((size_t)s * channels + c) * 3 + t] = share;
            else atomicAdd(dwx + (size_t)c * dwx_pitch + t, share);
        }
        const FxPlane fx = fx_plane(fmaxf(fmaxf(s_red[0][3], s_red[1][3]), fmaxf(s_red[2][3], s_red[3][3])),
                                    (s_red[0][4] + s_red[1][4]) + (s_red[2][4] + s_red[3][4]) != 0.f, pshift);
        float acc = 0.f;
        constexpr int VB = EMAX < 16 ? EMAX : 16;      // values in flight (32 at once spill at two workgroups per CU)
#pragma unroll
        for (int k0 = 0; k0 < EMAX; k0 += VB) {
            float val[VB];
#pragma unroll
            for (int k = 0; k < VB; ++k) val[k] = s_plane[iq[k0 + k] & 0xffffu];
#pragma unroll
            for (int k = 0; k < VB; ++k) {
                acc += val[k];
                if (endmask & (1u << (k0 + k))) {
                    fx_add(&s_out[iq[k0 + k] >> 16], acc, fx);
                    acc = 0.f;
                }
            }
        }
        __syncthreads();
        float *out = dproj + ((size_t)s * channels + c) * n_src;
        for (int q = t; q < n_src; q += 256) out[q] = fx_value(s_out[q], fx);
    }
}

// Backward of three_interpolate in gather form: known point k sums w * grad over the (unknown point, neighbour slot) positions that
// reference it (the inverse table of the interpolation indices, rtk_group_inverse_index with positions = 3 n), for TG_CPB channels
// at a time with the gradient planes staged in LDS.  The scatter form (one ds_add_f32 per term, ops_pointnet2.hip) runs at
// ~3 clocks per LANE: LDS float atomics are the slowest instruction of this path (270 us per step for 50 MB of data).
constexpr int TG_CPB = 8;
// n_valid (optional): padded clouds.  The unknown points from n_valid[b] on are copies of point 0 -- same neighbours, same weights --,
// so their gradient is folded into point 0's plane element here and their positions are left out of the table
// (rtk_inverse_index_job_t.live): sum_p go[p] w[p] over the copies = w[0] sum_p go[p].  Otherwise the three known points they all
// reference get lists of 60-100 entries next to lists of 3, each walked by one thread.
__global__ __launch_bounds__(256) void three_interp_grad_gather_kernel(int c, int n, int m, const float *__restrict__ grad_out,
                                                                       const float *__restrict__ weight, const int *__restrict__ off,
                                                                       const unsigned short *__restrict__ inv, float *__restrict__ grad_points,
                                                                       const int *__restrict__ n_valid) {
    extern __shared__ float s_go[];                                // [TG_CPB][n]
    const int bs = blockIdx.y, c0 = blockIdx.x * TG_CPB, tid = threadIdx.x;
    const int nc = min(TG_CPB, c - c0);
    const float *go = grad_out + ((size_t)bs * c + c0) * n;
    for (int e = tid; e < nc * n; e += 256) s_go[e] = go[e];
    __syncthreads();
    if (n_valid) {                                                 // 32 lanes per channel plane
        const int nv = n_valid[bs], q = tid >> 5, l = tid & 31;
        float part = 0.f;
        if (q < nc)
            for (int p = nv + l; p < n; p += 32) part += s_go[q * n + p];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        if (q < nc && l == 0 && nv < n) s_go[q * n] += part;
        __syncthreads();
    }
    const int *ob = off + (size_t)bs * (m + 1);
    const unsigned short *ib = inv + (size_t)bs * 3 * n;
    const float *wb = weight + (size_t)bs * 3 * n;
    for (int k = tid; k < m; k += 256) {
        const int a = ob[k], b = ob[k + 1];
        float acc[TG_CPB];
#pragma unroll
        for (int q = 0; q < TG_CPB; ++q) acc[q] = 0.f;
        for (int i = a; i < b; ++i) {
            const int pos = ib[i], pt = pos / 3;
            const float w = wb[pos];
#pragma unroll
            for (int q = 0; q < TG_CPB; ++q) acc[q] += s_go[min(q, nc - 1) * n + pt] * w;
        }
        float *gp = grad_points + ((size_t)bs * c + c0) * m + k;
#pragma unroll
        for (int q = 0; q < TG_CPB; ++q)
            if (q < nc) gp[(size_t)q * m] = acc[q];
    }
}

// Feature gradient of the patch aggregation (rtk_patch_cost: out[i] = sum_k wn(i,k) * feat[knn[i,k]]) in gather form:
// dfeat[m] = sum over the positions (i, k) with knn[i,k] = m of wn(i,k) * dout[i], wn recomputed from the position's hidden
// activation t2 (8 values, stored by rtk_patch_cost_bwd): 8 multiply-adds per element instead of materialising wn * dout for every
// position (268 MB at B = 64) and scattering it.  Thread = channel, PG_R destination rows per workgroup, four positions in flight.
constexpr int PG_R = 8;
// row0 (optional, (samples, 256)): padded clouds -- the gradient row to use for query point 0: its own plus those of the padding copies
// of point 0 (same neighbours, same WeightNet output), whose positions are left out of the table (patch_fold_padding_kernel).
__global__ __launch_bounds__(256) void patch_fold_padding_kernel(int n, const int *__restrict__ n_valid, const float *__restrict__ dout,
                                                                 int dout_pitch, float *__restrict__ row0) {
    const int b = blockIdx.x, c = threadIdx.x;
    const float *db = dout + (size_t)b * n * dout_pitch + c;
    float a = db[0];
    for (int i = n_valid[b]; i < n; ++i) a += db[(size_t)i * dout_pitch];
    row0[(size_t)b * 256 + c] = a;
}

__global__ __launch_bounds__(256) void patch_dfeat_gather_kernel(int n, const int *__restrict__ off, const unsigned short *__restrict__ inv,
                                                                 const float *__restrict__ t2, const float *__restrict__ wc,
                                                                 const float *__restrict__ bc, const float *__restrict__ dout, int dout_pitch,
                                                                 float *__restrict__ dfeat, const float *__restrict__ row0) {
    const int b = blockIdx.y, m0 = blockIdx.x * PG_R, c = threadIdx.x;
    float w[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) w[h] = wc[c * 8 + h];
    const float bias = bc[c];
    const int *ob = off + (size_t)b * (n + 1);
    const unsigned short *ib = inv + (size_t)b * 16 * n;
    const float *tb = t2 + (size_t)b * 16 * n * 8;
    const float *db = dout + (size_t)b * n * dout_pitch + c;
    for (int r = 0; r < PG_R && m0 + r < n; ++r) {
        const int m = m0 + r, a = ob[m], e = ob[m + 1];
        float acc = 0.f;
        for (int i = a; i < e; i += 4) {
            int pos[4];
            float d[4];
            float4 ta[4], tb4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) pos[k] = ib[min(i + k, e - 1)];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                d[k] = (row0 && (pos[k] >> 4) == 0) ? row0[(size_t)b * 256 + c] : db[(size_t)(pos[k] >> 4) * dout_pitch];
                ta[k] = *reinterpret_cast<const float4 *>(tb + (size_t)pos[k] * 8);
                tb4[k] = *reinterpret_cast<const float4 *>(tb + (size_t)pos[k] * 8 + 4);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float o = bias;
                o = __fmaf_rn(w[0], ta[k].x, o); o = __fmaf_rn(w[1], ta[k].y, o); o = __fmaf_rn(w[2], ta[k].z, o); o = __fmaf_rn(w[3], ta[k].w, o);
                o = __fmaf_rn(w[4], tb4[k].x, o); o = __fmaf_rn(w[5], tb4[k].y, o); o = __fmaf_rn(w[6], tb4[k].z, o); o = __fmaf_rn(w[7], tb4[k].w, o);
                if (i + k < e) acc += fmaxf(o, 0.f) * d[k];
            }
        }
        dfeat[((size_t)b * n + m) * 256 + c] = acc;
    }
}

}  // namespace

extern "C" int rtk_group_inverse_index(int samples, int n_src, int positions, const int *idx, int *off, unsigned short *inv,
                                       rtk_stream_t stream) {
    RTK_REQUIRE(samples > 0 && n_src > 0 && positions > 0 && idx && off && inv, "group_inverse_index: bad arguments");
    RTK_REQUIRE(positions <= 65536 && n_src <= 8192, "group_inverse_index: %d positions / %d source points exceed the 16-bit table",
                positions, n_src);
    const size_t lds = (2 * (size_t)n_src + 1) * sizeof(int) + (size_t)positions * sizeof(unsigned short);
    RTK_REQUIRE(lds <= 150 * 1024, "group_inverse_index: table exceeds the LDS budget");
    (void)hipFuncSetAttribute((const void *)inverse_index_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);      // per device and cheap: every call
    inverse_index_kernel<<<samples, II_T, lds, (hipStream_t)stream>>>(n_src, positions, idx, off, inv);
    RTK_CHECK_LAUNCH("group_inverse_index");
    return RTK_OK;
}

extern "C" int rtk_group_inverse_index_multi(int samples, int njobs, const rtk_inverse_index_job_t *jobs, rtk_stream_t stream) {
    RTK_REQUIRE(samples > 0 && samples <= 65535 && njobs > 0 && njobs <= II_MAX_JOBS && jobs, "group_inverse_index_multi: bad arguments (%d jobs)", njobs);
    IIJobs J;
    J.n = njobs;
    size_t lds = 0;
    for (int i = 0; i < njobs; ++i) {
        const rtk_inverse_index_job_t &q = jobs[i];
        RTK_REQUIRE(q.n_src > 0 && q.positions > 0 && q.idx && q.off && q.inv && q.positions <= 65536 && q.n_src <= 8192,
                    "group_inverse_index_multi: bad job %d", i);
        const size_t l = (2 * (size_t)q.n_src + 1) * sizeof(int) + (size_t)q.positions * sizeof(unsigned short);
        lds = l > lds ? l : lds;
        J.j[i] = q;
    }
    RTK_REQUIRE(lds <= 150 * 1024, "group_inverse_index_multi: a table exceeds the LDS budget");
    (void)hipFuncSetAttribute((const void *)inverse_index_multi_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    inverse_index_multi_kernel<<<dim3(samples, njobs), II_T, lds, (hipStream_t)stream>>>(J);
    RTK_CHECK_LAUNCH("group_inverse_index_multi");
    return RTK_OK;
}

extern "C" int rtk_sa_first_layer_bwd(int samples, int channels, int rows, int ns, int n_src, const float *dz, const float *dxyz,
                                      const int *off, const unsigned short *inv, float *dproj, float *dwx, int dwx_pitch,
                                      float *dwx_ws, rtk_stream_t stream) {
    RTK_REQUIRE(samples > 0 && channels > 0 && rows > 0 && ns >= 4 && (ns & 3) == 0 && n_src > 0 && dz && dxyz && off && inv && dproj &&
                dwx && dwx_pitch >= 3, "sa_first_layer_bwd: bad arguments");
    const int P = rows * ns;
    const size_t lds = (size_t)(P + 4) * sizeof(float) + (size_t)n_src * sizeof(long long) + (size_t)(n_src + 1) * sizeof(int) +
                       (size_t)(P + 256) * sizeof(unsigned short);
    RTK_REQUIRE(P <= 65536 && lds <= 150 * 1024 && samples <= 65535, "sa_first_layer_bwd: %d positions exceed the LDS budget", P);
    hipStream_t st = (hipStream_t)stream;
    if (P <= 8192 && n_src < 65536) {
        // planes of up to 8192 positions: the run structure lives in registers, a plane costs little, so more planes per workgroup
        // amortise the per-workgroup setup (index tables, offsets, the walk over the offset table)
        const long planes = (long)channels * samples;
        const int cg = planes >= 4096 ? 8 : planes >= 1024 ? 4 : 1;
        const int nbx = (channels + cg - 1) / cg, gx = samples % 8 == 0 ? nbx : 0;
        const dim3 grid(gx ? nbx * samples : nbx, gx ? 1 : samples);
        const int E = (P + 255) >> 8;
#define FB_CASE(EM)                                                                                                                             \
    {                                                                                                                                           \
        (void)hipFuncSetAttribute((const void *)sa_first_layer_bwd_fast_kernel<EM>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);   \
        sa_first_layer_bwd_fast_kernel<EM><<<grid, 256, lds, st>>>(channels, cg, gx, n_src, P, dz, dxyz, off, inv, dproj, dwx_ws, dwx, dwx_pitch);           \
    }
        if (E <= 8) FB_CASE(8) else if (E <= 16) FB_CASE(16) else FB_CASE(32)
#undef FB_CASE
        if (dwx_ws) dwx_reduce_kernel<<<rtk_divup(channels * 3, 4), 256, 0, st>>>(samples, channels, dwx_ws, dwx, dwx_pitch);
        RTK_CHECK_LAUNCH("sa_first_layer_bwd");
        return RTK_OK;
    }
    (void)hipFuncSetAttribute((const void *)sa_first_layer_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);      // per device and cheap: every call
    // channel planes per workgroup: 4 (2, 8 and 16 measured within 5 % of it or worse, tools/experiments/exp_firstbwd.py: the per-plane phases,
    // not the per-workgroup staging of the index tables and offset planes, set the pace)
    // ... at large batches; with few samples one plane per workgroup (the serial chain per workgroup is what counts there)
    const int cg = ((long)((channels + 3) / 4) * samples >= 256 ? 4 : 1);
    const dim3 grid((channels + cg - 1) / cg, samples);
    sa_first_layer_bwd_kernel<<<grid, 256, lds, st>>>(channels, cg, n_src, P, dz, dxyz, off, inv, dproj, dwx_ws, dwx, dwx_pitch);
    if (dwx_ws) dwx_reduce_kernel<<<rtk_divup(channels * 3, 4), 256, 0, st>>>(samples, channels, dwx_ws, dwx, dwx_pitch);
    RTK_CHECK_LAUNCH("sa_first_layer_bwd");
    return RTK_OK;
}

extern "C" int rtk_three_interpolate_grad_gather(int b, int c, int n, int m, const float *grad_out, const float *weight, const int *off,
                                                 const unsigned short *inv, float *grad_points, const int *n_valid, rtk_stream_t stream) {
    RTK_REQUIRE(b > 0 && c > 0 && n > 0 && m > 0 && grad_out && weight && off && inv && grad_points, "three_interpolate_grad_gather: bad arguments");
    RTK_REQUIRE(3 * (long)n <= 65536 && (size_t)TG_CPB * n * sizeof(float) <= 64 * 1024 && b <= 65535,
                "three_interpolate_grad_gather: %d unknown points exceed the 16-bit table / the LDS planes", n);
    three_interp_grad_gather_kernel<<<dim3((c + TG_CPB - 1) / TG_CPB, b), 256, (size_t)TG_CPB * n * sizeof(float), (hipStream_t)stream>>>(
        c, n, m, grad_out, weight, off, inv, grad_points, n_valid);
    RTK_CHECK_LAUNCH("three_interpolate_grad_gather");
    return RTK_OK;
}

extern "C" int rtk_patch_dfeat_gather(int samples, int n, const int *off, const unsigned short *inv, const float *t2, const float *wc,
                                      const float *bc, const float *dout, int dout_pitch, float *dfeat, const int *n_valid, float *row0,
                                      rtk_stream_t stream) {
    RTK_REQUIRE(!n_valid == !row0, "patch_dfeat_gather: n_valid and the row-0 workspace go together");
    if (n_valid) {
        patch_fold_padding_kernel<<<samples, 256, 0, (hipStream_t)stream>>>(n, n_valid, dout, dout_pitch, row0);
        RTK_CHECK_LAUNCH("patch_dfeat_gather");
    }
    RTK_REQUIRE(samples > 0 && n >= 16 && 16 * (long)n <= 65536 && off && inv && t2 && wc && bc && dout && dfeat && dout_pitch >= 256 &&
                samples <= 65535, "patch_dfeat_gather: bad arguments");
    patch_dfeat_gather_kernel<<<dim3((n + PG_R - 1) / PG_R, samples), 256, 0, (hipStream_t)stream>>>(n, off, inv, t2, wc, bc, dout, dout_pitch,
                                                                                                     dfeat, row0);
    RTK_CHECK_LAUNCH("patch_dfeat_gather");
    return RTK_OK;
}
