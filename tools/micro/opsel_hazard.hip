// opsel_hazard.hip -- standalone reproducer (no torch, no library) for the round-4 finding of DESIGN section 8: a packed fp32
// operation that takes a broadcast operand from the ODD half of a register pair through op_sel misread about once in 10^4
// executions while waves of other kernels issuing bf16 MFMAs shared its SIMD.
//
//   hipcc -O2 --offload-arch=gfx950 tools/micro/opsel_hazard.hip -o tools/micro/opsel_hazard.bin && tools/micro/opsel_hazard.bin
//   (with the library's own kernels as noise -- what shared the SIMDs with the failing kernel in round 4:
//    hipcc -O2 --offload-arch=gfx950 -DWITH_LIBRARY -Iinclude tools/micro/opsel_hazard.hip -Lratrack_amd/lib -lrtk_hip
//          -Wl,-rpath,$PWD/ratrack_amd/lib -o tools/micro/opsel_hazard_lib.bin)
//
// Victim: one wave per workgroup runs ROUNDS rounds of a furthest-point-selection-like dependent chain (the loop of
// fps_wave_body, ops_pointnet2.hip): read the current point o = (x, y, z, id) with ONE uniform ds_read_b128, take the squared
// distances of the lane's four points to it with packed fp32 arithmetic, keep running minima, pick the wave-wide maximum as the
// next point.  The only thing that differs between the FORMs is how o's components reach the packed arithmetic:
//   FORM 0  three v_mov copies into pairs of their own, op_sel_hi:[1,0]            (the form the library pins; never misread)
//   FORM 1  straight out of the ds_read_b128 destination: y through op_sel:[0,1]    (the form round 4 blamed)
//   FORM 2  as 1 with `s_nop 7` between the read's wait and the packed op
//   FORM 3  as 1 with the packed SUBTRACT replaced by v_pk_fma_f32 (x * 1 + (-o)), op_sel:[0,0,1]
//   FORM 6  as 1 with the read's destination at v[98:101]: (x, y) in register banks 2, 3 like the v[14:15] of the failing build
//   FORM 7  the odd half through v_pk_mul_f32 op_sel:[0,1];  FORM 8  the pair in the OTHER operand slot (op_sel:[1,0]);
//   FORM 9  as 1 on a v_mov copy of the pair (not the LDS read's own destination)
//   FORM 4  two-piece fp16 split of the lane's values: residual through v_fma_mix_f32 taking the HIGH half of a packed f16
//           pair (op_sel on a 16-bit half -- the form the round-5 split layers would use), checked against the low-half form
//   FORM 5  v_fma_mixlo_f16 / v_fma_mixhi_f16 residuals (what split_f16.h uses)
// Every form's trace is compared bit for bit with FORM 0's arithmetic run on an idle GPU (forms 4/5: with their own low-half
// twins); mismatching workgroups are counted per launch, with and without the noise kernel (bf16 MFMA + LDS traffic, few
// registers: its waves share SIMDs with the victim's) on a second stream.
#include <hip/hip_runtime.h>
#ifdef WITH_LIBRARY      // -DWITH_LIBRARY -Iinclude -Lratrack_amd/lib -lrtk_hip: the library's own split kernels as a third kind of noise
#include "rtk_fused.h"
#endif
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int NPTS = 256, ROUNDS = 255;

__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const unsigned o = __shfl_xor(v, s, 64);
        v = o > v ? o : v;
    }
    return v;
}

template <int FORM>
__global__ __launch_bounds__(64) void victim(const float4 *__restrict__ clouds, int *__restrict__ trace) {
    __shared__ float4 s_pt[NPTS];
    const int lane = threadIdx.x, b = blockIdx.x;
    for (int k = lane; k < NPTS; k += 64) s_pt[k] = clouds[(size_t)b * NPTS + k];
    __syncthreads();
    f2 x[2], y[2], z[2];
    unsigned t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 v = s_pt[lane * 4 + i];
        x[i / 2][i % 2] = v.x; y[i / 2][i % 2] = v.y; z[i / 2][i % 2] = v.z;
        t[i] = __float_as_uint(1e10f);
    }
    int pos = 0;
    int *tr = trace + (size_t)b * ROUNDS;
    for (int j = 0; j < ROUNDS; ++j) {
        const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) void *)(s_pt) + 16u * (unsigned)pos;
        f2 dx[2], dy[2], dz[2];
        if constexpr (FORM == 0) {
            f4 o;
            asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(o) : "v"(addr));
            f2 ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z};
            asm volatile("" : "+v"(ox), "+v"(oy), "+v"(oz));
#pragma unroll
            for (int h = 0; h < 2; ++h) { dx[h] = x[h] - ox; dy[h] = y[h] - oy; dz[h] = z[h] - oz; }
        } else if constexpr ((FORM >= 1 && FORM <= 3) || (FORM >= 6 && FORM <= 9)) {
            // the winner's coordinates stay where ds_read_b128 left them (v[100:103]); the packed ops select halves through op_sel
#define PK(op, extra)                                                                                                         \
            asm volatile("ds_read_b128 v[100:103], %6\n s_waitcnt lgkmcnt(0)\n" extra                                            \
                         op(0, 7, "v[100:101]", "op_sel_hi:[1,0]") op(1, 8, "v[100:101]", "op_sel_hi:[1,0]")                     \
                         op(2, 9, "v[100:101]", "op_sel:[0,1]") op(3, 10, "v[100:101]", "op_sel:[0,1]")                          \
                         op(4, 11, "v[102:103]", "op_sel_hi:[1,0]") op(5, 12, "v[102:103]", "op_sel_hi:[1,0]")                   \
                         : "=&v"(dx[0]), "=&v"(dx[1]), "=&v"(dy[0]), "=&v"(dy[1]), "=&v"(dz[0]), "=&v"(dz[1])                    \
                         : "v"(addr), "v"(x[0]), "v"(x[1]), "v"(y[0]), "v"(y[1]), "v"(z[0]), "v"(z[1]), "v"(one)                 \
                         : "v100", "v101", "v102", "v103", "memory")
#define OP_ADD(d, s, pair, sel) "v_pk_add_f32 %" #d ", %" #s ", " pair " " sel " neg_lo:[0,1] neg_hi:[0,1]\n"
            const f2 one = {1.0f, 1.0f};
            if constexpr (FORM == 7) {          // the odd half through v_pk_MUL_f32 op_sel:[0,1] (one * o.y, exact), then a plain packed subtract
                f2 oy0, oy1;
                asm volatile("ds_read_b128 v[100:103], %8\n s_waitcnt lgkmcnt(0)\n"
                             "v_pk_add_f32 %0, %9, v[100:101] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
                             "v_pk_add_f32 %1, %10, v[100:101] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
                             "v_pk_mul_f32 %6, %15, v[100:101] op_sel:[0,1]\n"
                             "v_pk_mul_f32 %7, %15, v[100:101] op_sel:[0,1]\n"
                             "v_pk_add_f32 %4, %13, v[102:103] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
                             "v_pk_add_f32 %5, %14, v[102:103] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
                             "v_pk_add_f32 %2, %11, %6 neg_lo:[0,1] neg_hi:[0,1]\n"
                             "v_pk_add_f32 %3, %12, %7 neg_lo:[0,1] neg_hi:[0,1]\n"
                             : "=&v"(dx[0]), "=&v"(dx[1]), "=&v"(dy[0]), "=&v"(dy[1]), "=&v"(dz[0]), "=&v"(dz[1]), "=&v"(oy0), "=&v"(oy1)
                             : "v"(addr), "v"(x[0]), "v"(x[1]), "v"(y[0]), "v"(y[1]), "v"(z[0]), "v"(z[1]), "v"(one)
                             : "v100", "v101", "v102", "v103", "memory");
            } else if constexpr (FORM == 8) {   // the pair as src0: v_pk_add_f32 d, -o, x op_sel:[1,0] (the other operand slot)
                asm volatile("ds_read_b128 v[100:103], %6\n s_waitcnt lgkmcnt(0)\n"
                             "v_pk_add_f32 %0, %7, v[100:101] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
                             "v_pk_add_f32 %1, %8, v[100:101] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
                             "v_pk_add_f32 %2, v[100:101], %9 op_sel:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n"
                             "v_pk_add_f32 %3, v[100:101], %10 op_sel:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n"
                             "v_pk_add_f32 %4, %11, v[102:103] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
                             "v_pk_add_f32 %5, %12, v[102:103] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
                             : "=&v"(dx[0]), "=&v"(dx[1]), "=&v"(dy[0]), "=&v"(dy[1]), "=&v"(dz[0]), "=&v"(dz[1])
                             : "v"(addr), "v"(x[0]), "v"(x[1]), "v"(y[0]), "v"(y[1]), "v"(z[0]), "v"(z[1]), "v"(one)
                             : "v100", "v101", "v102", "v103", "memory");
            } else if constexpr (FORM == 9) {   // as 1, but the pair is a VALU COPY of the read's destination (is it the LDS return path?)
                asm volatile("ds_read_b128 v[100:103], %6\n s_waitcnt lgkmcnt(0)\n"
                             "v_mov_b32 v104, v100\n v_mov_b32 v105, v101\n v_mov_b32 v106, v102\n v_mov_b32 v107, v103\n s_nop 1\n"
                             "v_pk_add_f32 %0, %7, v[104:105] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
                             "v_pk_add_f32 %1, %8, v[104:105] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
                             "v_pk_add_f32 %2, %9, v[104:105] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n"
                             "v_pk_add_f32 %3, %10, v[104:105] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n"
                             "v_pk_add_f32 %4, %11, v[106:107] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
                             "v_pk_add_f32 %5, %12, v[106:107] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
                             : "=&v"(dx[0]), "=&v"(dx[1]), "=&v"(dy[0]), "=&v"(dy[1]), "=&v"(dz[0]), "=&v"(dz[1])
                             : "v"(addr), "v"(x[0]), "v"(x[1]), "v"(y[0]), "v"(y[1]), "v"(z[0]), "v"(z[1]), "v"(one)
                             : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "memory");
            } else if constexpr (FORM == 6) {
                asm volatile("ds_read_b128 v[98:101], %6\n s_waitcnt lgkmcnt(0)\n"
                             "v_pk_add_f32 %0, %7, v[98:99] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
                             "v_pk_add_f32 %1, %8, v[98:99] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
                             "v_pk_add_f32 %2, %9, v[98:99] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n"
                             "v_pk_add_f32 %3, %10, v[98:99] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n"
                             "v_pk_add_f32 %4, %11, v[100:101] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
                             "v_pk_add_f32 %5, %12, v[100:101] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
                             : "=&v"(dx[0]), "=&v"(dx[1]), "=&v"(dy[0]), "=&v"(dy[1]), "=&v"(dz[0]), "=&v"(dz[1])
                             : "v"(addr), "v"(x[0]), "v"(x[1]), "v"(y[0]), "v"(y[1]), "v"(z[0]), "v"(z[1]), "v"(one)
                             : "v98", "v99", "v100", "v101", "memory");
            } else if constexpr (FORM == 1) PK(OP_ADD, "");
            else if constexpr (FORM == 2) PK(OP_ADD, "s_nop 7\n");
            else {
                // v_pk_fma_f32 d, x, 1, -o : the three-operand op_sel list, odd half of src2 for y
                asm volatile("ds_read_b128 v[100:103], %6\n s_waitcnt lgkmcnt(0)\n"
                             "v_pk_fma_f32 %0, %7, %13, v[100:101] op_sel_hi:[1,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
                             "v_pk_fma_f32 %1, %8, %13, v[100:101] op_sel_hi:[1,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
                             "v_pk_fma_f32 %2, %9, %13, v[100:101] op_sel:[0,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
                             "v_pk_fma_f32 %3, %10, %13, v[100:101] op_sel:[0,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
                             "v_pk_fma_f32 %4, %11, %13, v[102:103] op_sel_hi:[1,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
                             "v_pk_fma_f32 %5, %12, %13, v[102:103] op_sel_hi:[1,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
                             : "=&v"(dx[0]), "=&v"(dx[1]), "=&v"(dy[0]), "=&v"(dy[1]), "=&v"(dz[0]), "=&v"(dz[1])
                             : "v"(addr), "v"(x[0]), "v"(x[1]), "v"(y[0]), "v"(y[1]), "v"(z[0]), "v"(z[1]), "v"(one)
                             : "v100", "v101", "v102", "v103", "memory");
            }
#undef PK
        }
        if constexpr (FORM <= 3 || (FORM >= 6 && FORM <= 9)) {
            unsigned mloc = 0u;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f2 d = dx[h] * dx[h];
                d = __builtin_elementwise_fma(dy[h], dy[h], d);
                d = __builtin_elementwise_fma(dz[h], dz[h], d);
                const unsigned d0 = __float_as_uint(d[0]), d1 = __float_as_uint(d[1]);
                t[2 * h] = d0 < t[2 * h] ? d0 : t[2 * h];
                t[2 * h + 1] = d1 < t[2 * h + 1] ? d1 : t[2 * h + 1];
                const unsigned mm = t[2 * h] > t[2 * h + 1] ? t[2 * h] : t[2 * h + 1];
                mloc = mm > mloc ? mm : mloc;
            }
            int sl = 0;
#pragma unroll
            for (int i = 3; i >= 0; --i) sl = t[i] == mloc ? i : sl;
            const unsigned M = wave_max_u32(mloc);
            const unsigned long long mask = __ballot(mloc == M);
            const int wl = __builtin_ctzll(mask);
            pos = wl * 4 + __builtin_amdgcn_readlane(sl, wl);
            if (lane == 0) tr[j] = pos;
        }
    }
}

// FORMs 4, 5: the fp16 two-piece split.  x -> h = f16(x), l = f16(x - h); the residual takes h from the LOW or the HIGH half of
// the packed pair.  A dependent chain: the next round's operands are this round's residual bits mixed back in.
template <int FORM, bool TWIN>
__global__ __launch_bounds__(64) void victim_f16(const float4 *__restrict__ clouds, unsigned *__restrict__ trace, int rounds) {
    const int lane = threadIdx.x, b = blockIdx.x;
    const float4 v = clouds[(size_t)b * NPTS + lane * 4];
    float a0 = v.x, a1 = v.y;
    unsigned acc = 0u;
    for (int j = 0; j < rounds; ++j) {
        unsigned hp, lp;
        float r0, r1;
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hp) : "v"(a0), "v"(a1));
        if constexpr (FORM == 4) {
            if constexpr (!TWIN) {
                asm volatile("s_nop 1\n v_fma_mix_f32 %0, %2, 1.0, -%4 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n"
                             "v_fma_mix_f32 %1, %3, 1.0, -%4 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                             : "=&v"(r0), "=&v"(r1) : "v"(a0), "v"(a1), "v"(hp));
            } else {
                unsigned hi;
                asm volatile("s_nop 1\n v_lshrrev_b32 %0, 16, %1" : "=v"(hi) : "v"(hp));
                asm volatile("s_nop 1\n v_fma_mix_f32 %0, %2, 1.0, -%4 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n"
                             "v_fma_mix_f32 %1, %3, 1.0, -%5 op_sel:[0,0,0] op_sel_hi:[0,0,1]"
                             : "=&v"(r0), "=&v"(r1) : "v"(a0), "v"(a1), "v"(hp), "v"(hi));
            }
            asm volatile("s_nop 1\n v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lp) : "v"(r0), "v"(r1));
        } else {
            if constexpr (!TWIN) {
                lp = 0u;
                asm volatile("s_nop 1\n v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n"
                             "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                             : "+v"(lp) : "v"(a0), "v"(a1), "v"(hp));
            } else {
                unsigned hi;
                asm volatile("s_nop 1\n v_lshrrev_b32 %0, 16, %1" : "=v"(hi) : "v"(hp));
                asm volatile("s_nop 1\n v_fma_mix_f32 %0, %2, 1.0, -%4 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n"
                             "v_fma_mix_f32 %1, %3, 1.0, -%5 op_sel:[0,0,0] op_sel_hi:[0,0,1]"
                             : "=&v"(r0), "=&v"(r1) : "v"(a0), "v"(a1), "v"(hp), "v"(hi));
                asm volatile("s_nop 1\n v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lp) : "v"(r0), "v"(r1));
            }
        }
        acc = acc * 0x9E3779B1u + (hp ^ (lp * 0x85EBCA6Bu));
        // next operands: a pseudo-random float in [1, 2) x a slowly varying scale, from the bits just produced
        a0 = __uint_as_float(0x3f800000u | ((acc >> 9) & 0x7fffffu)) * 0.37f;
        a1 = __uint_as_float(0x3f800000u | ((acc * 0xC2B2AE35u) >> 9)) * 1.91f;
    }
    trace[(size_t)b * 64 + lane] = acc;
}

// Noise: bf16 MFMA stream with LDS reads, 8 waves per workgroup, few registers -- the shape of the split per-point / SA kernels.
__global__ __launch_bounds__(256) void noise(float *out, int iters) {
    __shared__ float4 s_w[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) s_w[i] = make_float4(1e-3f * i, 2e-3f * i, 1.f, 0.5f);
    __syncthreads();
    f4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        const float4 w = s_w[(threadIdx.x + 64 * it) & 1023];
        bf8v a = __builtin_bit_cast(bf8v, w), bb = __builtin_bit_cast(bf8v, s_w[(threadIdx.x * 7 + it) & 1023]);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bb, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (s == 12345.678f) out[0] = s;
}

// Noise B: the shape of the library's split kernels -- a global_load_lds (LDS-DMA) weight stream through a double buffer, barriers,
// 16-byte LDS reads, 32x32x16 bf16 MFMAs.
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void noise_dma(const float4 *__restrict__ blob, float *out, int iters) {
    __shared__ __attribute__((aligned(16))) float4 s_w[2 * 16 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f16v acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    int buf = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int f = 0; f < 4; ++f)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(blob + ((it * 16 + wave * 4 + f) & 1023) * 64 + lane),
                                             (__attribute__((address_space(3))) void *)(s_w + (buf * 16 + wave * 4 + f) * 64), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const float4 a = s_w[(buf * 16 + f) * 64 + lane], bb = s_w[(buf * 16 + 8 + f) * 64 + lane];
            acc[f & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8v, a), __builtin_bit_cast(bf8v, bb), acc[f & 1], 0, 0, 0);
        }
        buf ^= 1;
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[0][e] + acc[1][e];
    if (s == 12345.678f) out[0] = s;
}
static const float4 *g_blob = nullptr;

// Noise by INGREDIENT: one instruction class per kernel, 8 waves per workgroup, few registers (they share SIMDs with the victim).
//   0 plain v_fma_f32   1 v_pk_fma_f32 / v_pk_mul_f32 (no op_sel)   2 v_pk_mul_f32 op_sel_hi:[1,0] (the lo-half broadcast compilers emit)
//   3 v_max_f32_dpp row_ror   4 v_mfma_f32_32x32x2_f32 (fp32-input MFMA)   5 v_cvt_pk_f16_f32 + v_fma_mixlo/hi   6 v_permlane32_swap
//   7 ds_read_b128 + ds_write_b128 traffic   8 v_pk_add_f32 op_sel:[0,1] itself   9-12 v_mfma_f32_16x16x32_f16 with, on its results, v_pk_fma /
//   v_fma / v_pk_mul (broadcast) / v_max3   13 v_readlane + v_writelane   14 v_exp_f32 + v_rcp_f32   15 global_load_dwordx4
template <int KIND>
__global__ __launch_bounds__(256) void noise_kind(float *out, int iters) {
    __shared__ float4 s_w[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) s_w[i] = make_float4(1e-3f * i, 2e-3f * i, 1.f, 0.5f);
    __syncthreads();
    f2 a = {1.0f + threadIdx.x * 1e-6f, 0.5f}, b = {0.999f, 1.001f}, c = {0.f, 1e-9f};
    f16v acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    unsigned u = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a.x) : "v"(b.x), "v"(c.y));
            else if constexpr (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b), "v"(c));
            else if constexpr (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(a) : "v"(b));
            else if constexpr (KIND == 3) asm volatile("s_nop 1\n v_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a.x));
            else if constexpr (KIND == 4) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            else if constexpr (KIND == 5) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2\n s_nop 1\n v_fma_mixlo_f16 %0, %1, 1.0, -%0 op_sel_hi:[0,0,1]\n"
                                                       "v_fma_mixhi_f16 %0, %2, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(u) : "v"(a.x), "v"(a.y));
            else if constexpr (KIND == 6) { auto r = __builtin_amdgcn_permlane32_swap(u, u + k, false, false); u = r[0] ^ r[1]; }
            else if constexpr (KIND == 7) { float4 t = s_w[(threadIdx.x * 5 + it + k) & 1023]; t.x += 1.f; s_w[(threadIdx.x + 64 * k) & 1023] = t; }
            else if constexpr (KIND == 8) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1]" : "+v"(a) : "v"(b));
            else if constexpr (KIND >= 9 && KIND <= 12) {      // a 16-bit MFMA with VALU work on its RESULTS in the same wave (a layer + its epilogue)
                typedef _Float16 h8 __attribute__((ext_vector_type(8)));
                h8 ha, hb;
#pragma unroll
                for (int t = 0; t < 8; ++t) { ha[t] = (_Float16)(a.x + t); hb[t] = (_Float16)(b.y + k); }
                f4 c4 = {acc[0], acc[1], acc[2], acc[3]};
                c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c4, 0, 0, 0);
                if constexpr (KIND == 9) { f2 lo = {c4.x, c4.y}, hi = {c4.z, c4.w}; lo = __builtin_elementwise_fma(lo, b, c); hi = lo * hi; acc[0] = lo.x; acc[1] = lo.y; acc[2] = hi.x; acc[3] = hi.y; }
                else if constexpr (KIND == 10) { acc[0] = __builtin_fmaf(c4.x, b.x, c.y); acc[1] = c4.y * b.y; acc[2] = c4.z; acc[3] = c4.w; }
                else if constexpr (KIND == 11) { f2 lo = {c4.x, c4.y}; lo = lo * b.x; acc[0] = lo.x; acc[1] = lo.y; acc[2] = c4.z; acc[3] = c4.w; }      // op_sel_hi:[1,0] broadcast
                else { acc[0] = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(c4.x), __builtin_fabsf(c4.y)), c4.z); acc[1] = c4.y; acc[2] = c4.z; acc[3] = c4.w; }
            } else if constexpr (KIND == 13) { unsigned sv_ = __builtin_amdgcn_readlane(u, k) + 1; asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(u) : "s"(sv_)); }
            else if constexpr (KIND == 14) asm volatile("v_exp_f32 %0, %0\n v_rcp_f32 %0, %0" : "+v"(a.x));
            else { float4 t = reinterpret_cast<const float4 *>(out)[(blockIdx.x * 256 + threadIdx.x + 1024 * (it * 8 + k)) & 0xfffff]; a.x += t.x; }      // 15: global loads
        }
    }
    float sum = a.x + a.y + (float)u + s_w[threadIdx.x].x;
#pragma unroll
    for (int e = 0; e < 16; ++e) sum += acc[e];
    if (sum == 12345.678f) out[0] = sum;
}

#ifdef WITH_LIBRARY
// Noise C: rtk_pointwise_mlp (128 -> 128 on 32 768 rows, split images) and rtk_sa_scale_split (128 clouds x 256 centroids x 32
// neighbours, 64 -> 64) of the library itself, on zero-filled operands (the instruction streams do not depend on the data).
struct LibNoise {
    float *rows, *out, *bias, *xyz, *q, *w1, *inv;
    void *img;
    int *idx, *nu;
    void init() {
        CHECK(hipMalloc(&rows, 32768 * 128 * 4)); CHECK(hipMemset(rows, 0, 32768 * 128 * 4));
        CHECK(hipMalloc(&out, 65536 * 128 * 4)); CHECK(hipMalloc(&bias, 1024 * 4)); CHECK(hipMemset(bias, 0, 1024 * 4));
        CHECK(hipMalloc(&img, 1 << 20)); CHECK(hipMemset(img, 0, 1 << 20));
        CHECK(hipMalloc(&xyz, 128 * 512 * 3 * 4)); CHECK(hipMemset(xyz, 0, 128 * 512 * 3 * 4));
        CHECK(hipMalloc(&q, 128 * 512 * 64 * 4)); CHECK(hipMemset(q, 0, 128 * 512 * 64 * 4));
        CHECK(hipMalloc(&w1, 4096 * 4)); CHECK(hipMemset(w1, 0, 4096 * 4));
        CHECK(hipMalloc(&idx, 128 * 512 * 32 * 4)); CHECK(hipMemset(idx, 0, 128 * 512 * 32 * 4));
        CHECK(hipMalloc(&nu, 128 * 4));
        int h[128]; for (int i = 0; i < 128; ++i) h[i] = 256;
        CHECK(hipMemcpy(nu, h, sizeof(h), hipMemcpyHostToDevice));
        CHECK(hipMalloc(&inv, 16)); float one[4] = {1.f, 1.f, 1.f, 1.f}; CHECK(hipMemcpy(inv, one, 16, hipMemcpyHostToDevice));
    }
    void launch(hipStream_t st, int which) {      // which: 1 = rtk_pointwise_mlp, 2 = rtk_sa_scale_split, 3 = both
        rtk_src_t src = {rows, 128, 128, 0};
        rtk_layer_t L = {reinterpret_cast<const float *>(img), bias, 8, 8, RTK_LAYER_SPLIT | 1, 1.0f};
        for (int k = 0; k < 6; ++k) {
            if ((which & 1) && rtk_pointwise_mlp(32768, 256, nullptr, 1, &src, nullptr, 1, &L, out, 128, 128, 0, nullptr, nullptr, (rtk_stream_t)st) != 0) { printf("rtk_pointwise_mlp failed\n"); exit(1); }
            if ((which & 2) && rtk_sa_scale_split(128, 512, 512, 32, xyz, xyz, idx, q, 64, 64, w1, img, inv, bias, out, 128, 0, nu, nu, (rtk_stream_t)st) != 0) { printf("rtk_sa_scale_split failed\n"); exit(1); }
        }
    }
};
static LibNoise g_lib;
#endif

struct Noise { const char *name; void (*launch)(hipStream_t, float *); };
template <int KIND> static void launch_kind(hipStream_t st, float *d) { noise_kind<KIND><<<1024, 256, 0, st>>>(d, 2500); }
static void launch_mfma(hipStream_t st, float *d) { noise<<<512, 256, 0, st>>>(d, 6000); }
static void launch_dma(hipStream_t st, float *d) { noise_dma<<<1024, 256, 0, st>>>(g_blob, d, 1500); }
#ifdef WITH_LIBRARY
static void launch_lib(hipStream_t st, float *) { g_lib.launch(st, 3); }
static void launch_lib_pw(hipStream_t st, float *) { g_lib.launch(st, 1); }
static void launch_lib_sa(hipStream_t st, float *) { g_lib.launch(st, 2); }
#endif
static std::vector<Noise> g_noises;

template <class Launch>
static void campaign(const char *name, Launch launch, void *d_out, size_t bytes, const std::vector<unsigned char> &expect, int launches,
                     int groups, size_t per_group, hipStream_t sv, hipStream_t sn, float *d_noise, long execs_per_launch) {
    std::vector<unsigned char> got(bytes);
    for (size_t kind = 0; kind <= g_noises.size(); ++kind) {
        long bad_groups = 0, bad_launches = 0;
        for (int L = 0; L < launches; ++L) {
            if (kind > 0) g_noises[kind - 1].launch(sn, d_noise);
            CHECK(hipMemsetAsync(d_out, 0, bytes, sv));
            launch(sv);
            CHECK(hipStreamSynchronize(sv));
            CHECK(hipMemcpy(got.data(), d_out, bytes, hipMemcpyDeviceToHost));
            long bad = 0;
            for (int g = 0; g < groups; ++g) bad += memcmp(got.data() + g * per_group, expect.data() + g * per_group, per_group) != 0;
            bad_groups += bad;
            bad_launches += bad != 0;
        }
        CHECK(hipDeviceSynchronize());
        printf("%-44s %-30s %3ld / %d launches differ, %6ld / %ld workgroups   (%.2e packed executions)\n", name,
               kind ? g_noises[kind - 1].name : "idle GPU", bad_launches, launches, bad_groups, (long)launches * groups,
               (double)launches * execs_per_launch);
        fflush(stdout);
    }
}

int main(int argc, char **argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 40, B = 2048;
    // noises: argv[2] = comma list out of  mfma,dma,fma,pk,pksel,dpp,mfma32,f16,swap,lds,pkodd,mfmapk,mfmafma,mfmabc,mfmamax,lane,trans,gload,lib,libpw,libsa  (default: mfma,dma[,lib])
    const char *sel = argc > 2 ? argv[2] : "mfma,dma,lib";
    auto want = [&](const char *k) { const char *p = strstr(sel, k); return p && (p == sel || p[-1] == ',') && (p[strlen(k)] == 0 || p[strlen(k)] == ','); };
    if (want("mfma")) g_noises.push_back({"bf16 MFMA + LDS reads", launch_mfma});
    if (want("dma")) g_noises.push_back({"bf16 MFMA + LDS-DMA stream", launch_dma});
    if (want("fma")) g_noises.push_back({"v_fma_f32", launch_kind<0>});
    if (want("pk")) g_noises.push_back({"v_pk_fma_f32 / v_pk_mul_f32", launch_kind<1>});
    if (want("pksel")) g_noises.push_back({"v_pk_mul_f32 op_sel_hi:[1,0]", launch_kind<2>});
    if (want("dpp")) g_noises.push_back({"v_max_f32_dpp", launch_kind<3>});
    if (want("mfma32")) g_noises.push_back({"v_mfma_f32_32x32x2_f32", launch_kind<4>});
    if (want("f16")) g_noises.push_back({"v_cvt_pk_f16_f32 + v_fma_mix", launch_kind<5>});
    if (want("swap")) g_noises.push_back({"v_permlane32_swap", launch_kind<6>});
    if (want("lds")) g_noises.push_back({"ds_read_b128 / ds_write_b128", launch_kind<7>});
    if (want("pkodd")) g_noises.push_back({"v_pk_add_f32 op_sel:[0,1]", launch_kind<8>});
    if (want("mfmapk")) g_noises.push_back({"f16 MFMA -> v_pk_fma_f32 on it", launch_kind<9>});
    if (want("mfmafma")) g_noises.push_back({"f16 MFMA -> v_fma_f32 on it", launch_kind<10>});
    if (want("mfmabc")) g_noises.push_back({"f16 MFMA -> v_pk_mul bcast", launch_kind<11>});
    if (want("mfmamax")) g_noises.push_back({"f16 MFMA -> v_max3_f32", launch_kind<12>});
    if (want("lane")) g_noises.push_back({"v_readlane / v_writelane", launch_kind<13>});
    if (want("trans")) g_noises.push_back({"v_exp_f32 / v_rcp_f32", launch_kind<14>});
    if (want("gload")) g_noises.push_back({"global_load_dwordx4", launch_kind<15>});
#ifdef WITH_LIBRARY
    if (want("lib")) g_noises.push_back({"library: pointwise + sa_scale_split", launch_lib});
    if (want("libpw")) g_noises.push_back({"library: rtk_pointwise_mlp (split)", launch_lib_pw});
    if (want("libsa")) g_noises.push_back({"library: rtk_sa_scale_split", launch_lib_sa});
#endif
    std::vector<float4> clouds((size_t)B * NPTS);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.0f; };
    for (auto &p : clouds) p = make_float4(rnd() * 20.f - 10.f, rnd() * 20.f - 10.f, rnd() * 4.f, 0.f);
    float4 *d_clouds; int *d_trace; unsigned *d_tr2; float *d_noise;
    CHECK(hipMalloc(&d_clouds, clouds.size() * sizeof(float4)));
    CHECK(hipMemcpy(d_clouds, clouds.data(), clouds.size() * sizeof(float4), hipMemcpyHostToDevice));
    const size_t tb = (size_t)B * ROUNDS * sizeof(int), tb2 = (size_t)B * 64 * sizeof(unsigned);
    CHECK(hipMalloc(&d_trace, tb)); CHECK(hipMalloc(&d_tr2, tb2)); CHECK(hipMalloc(&d_noise, 16 << 20)); CHECK(hipMemset(d_noise, 0, 16 << 20));
    float4 *d_blob;
    CHECK(hipMalloc(&d_blob, 1024 * 64 * sizeof(float4)));
    CHECK(hipMemset(d_blob, 0x3c, 1024 * 64 * sizeof(float4)));
    g_blob = d_blob;
    hipStream_t sv, sn;
    CHECK(hipStreamCreate(&sv)); CHECK(hipStreamCreate(&sn));
#ifdef WITH_LIBRARY
    g_lib.init();
#endif
    // expected traces: FORM 0 on an idle GPU, three times (must agree with itself)
    std::vector<unsigned char> expect(tb), again(tb);
    for (int k = 0; k < 3; ++k) {
        victim<0><<<B, 64, 0, sv>>>(d_clouds, d_trace);
        CHECK(hipStreamSynchronize(sv));
        CHECK(hipMemcpy(k ? again.data() : expect.data(), d_trace, tb, hipMemcpyDeviceToHost));
        if (k && memcmp(again.data(), expect.data(), tb)) { printf("FORM 0 is not reproducible on an idle GPU\n"); return 1; }
    }
    const long ex = (long)B * ROUNDS * 6;
#define VICTIM(F) [&](hipStream_t st) { victim<F><<<B, 64, 0, st>>>(d_clouds, d_trace); }
    campaign("0: broadcast copies, op_sel_hi:[1,0]", VICTIM(0), d_trace, tb, expect, launches, B, ROUNDS * sizeof(int), sv, sn, d_noise, ex);
    campaign("1: v_pk_add_f32 ... op_sel:[0,1]", VICTIM(1), d_trace, tb, expect, launches, B, ROUNDS * sizeof(int), sv, sn, d_noise, ex);
    campaign("7: v_pk_mul_f32 ... op_sel:[0,1]", VICTIM(7), d_trace, tb, expect, launches, B, ROUNDS * sizeof(int), sv, sn, d_noise, ex);
    campaign("8: v_pk_add_f32 (pair as src0) op_sel:[1,0]", VICTIM(8), d_trace, tb, expect, launches, B, ROUNDS * sizeof(int), sv, sn, d_noise, ex);
    campaign("9: as 1 on a VALU copy of the pair", VICTIM(9), d_trace, tb, expect, launches, B, ROUNDS * sizeof(int), sv, sn, d_noise, ex);
    campaign("6: as 1, pair in banks 2,3 (v[98:99])", VICTIM(6), d_trace, tb, expect, launches, B, ROUNDS * sizeof(int), sv, sn, d_noise, ex);
    campaign("2: as 1, s_nop 7 after the wait", VICTIM(2), d_trace, tb, expect, launches, B, ROUNDS * sizeof(int), sv, sn, d_noise, ex);
    campaign("3: v_pk_fma_f32 ... op_sel:[0,0,1]", VICTIM(3), d_trace, tb, expect, launches, B, ROUNDS * sizeof(int), sv, sn, d_noise, ex);
    const int r16 = 4000;
    std::vector<unsigned char> e16(tb2), e16b(tb2);
    victim_f16<4, true><<<B, 64, 0, sv>>>(d_clouds, d_tr2, r16);
    CHECK(hipStreamSynchronize(sv));
    CHECK(hipMemcpy(e16.data(), d_tr2, tb2, hipMemcpyDeviceToHost));
    victim_f16<5, true><<<B, 64, 0, sv>>>(d_clouds, d_tr2, r16);
    CHECK(hipStreamSynchronize(sv));
    CHECK(hipMemcpy(e16b.data(), d_tr2, tb2, hipMemcpyDeviceToHost));
    if (memcmp(e16.data(), e16b.data(), tb2)) { printf("the two low-half twins disagree\n"); return 1; }
    const long ex16 = (long)B * 64 * r16;
    campaign("4: v_fma_mix_f32, f16 HIGH half (op_sel)", [&](hipStream_t st) { victim_f16<4, false><<<B, 64, 0, st>>>(d_clouds, d_tr2, r16); },
             d_tr2, tb2, e16, launches, B, 64 * sizeof(unsigned), sv, sn, d_noise, ex16);
    campaign("5: v_fma_mixlo_f16 / v_fma_mixhi_f16", [&](hipStream_t st) { victim_f16<5, false><<<B, 64, 0, st>>>(d_clouds, d_tr2, r16); },
             d_tr2, tb2, e16, launches, B, 64 * sizeof(unsigned), sv, sn, d_noise, ex16);
    campaign("4t: low-half twin of 4 / 5", [&](hipStream_t st) { victim_f16<4, true><<<B, 64, 0, st>>>(d_clouds, d_tr2, r16); },
             d_tr2, tb2, e16, launches, B, 64 * sizeof(unsigned), sv, sn, d_noise, ex16);
    return 0;
}
