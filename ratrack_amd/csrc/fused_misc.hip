// fused_misc.hip -- the small stages around the MFMA kernels, written so that one backbone() issues a few
// dozen launches instead of the ~500 framework ops of the reference graph:
// input layout, FPS+gather, the GRU step, channel-major output layout.
#include <math.h>

#include "rtk_common.h"
#include "rtk_fused.h"
#include "rtk_train.h"

// ------------------------------------------------------------------------------------------------
// rtk_prepare_inputs
// ------------------------------------------------------------------------------------------------
__global__ void prepare_inputs_kernel(int b, int n, const float *__restrict__ pc1, const float *__restrict__ pc2,
                                      const float *__restrict__ f1, const float *__restrict__ f2, float *__restrict__ xyz,
                                      float *__restrict__ raw) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2L * b * n) return;
    const int s = (int)(t / n), p = (int)(t % n);
    const bool second = s >= b;
    const int sb = second ? s - b : s;
    const float *pc = second ? pc2 : pc1;
    const float *f = second ? f2 : f1;
    xyz[t * 3 + 0] = pc[((long)sb * 3 + 0) * n + p];
    xyz[t * 3 + 1] = pc[((long)sb * 3 + 1) * n + p];
    xyz[t * 3 + 2] = pc[((long)sb * 3 + 2) * n + p];
    *reinterpret_cast<float4 *>(raw + t * 4) = make_float4(f[((long)sb * 2 + 0) * n + p], f[((long)sb * 2 + 1) * n + p], 0.f, 0.f);
}

extern "C" int rtk_prepare_inputs(int b, int n, const float *pc1, const float *pc2, const float *feature1,
                                  const float *feature2, float *xyz, float *raw, rtk_stream_t stream) {
    RTK_REQUIRE(b > 0 && n > 0 && pc1 && pc2 && feature1 && feature2 && xyz && raw, "prepare_inputs: bad arguments");
    const long total = 2L * b * n;
    prepare_inputs_kernel<<<rtk_divup(total, 256), 256, 0, (hipStream_t)stream>>>(b, n, pc1, pc2, feature1, feature2, xyz, raw);
    RTK_CHECK_LAUNCH("prepare_inputs");
    return RTK_OK;
}

// ------------------------------------------------------------------------------------------------
// rtk_gru_step: one workgroup per sample, 3H threads.  Thread t owns gate row t of both matrices; weights come
// TRANSPOSED (L, H, 3H) so that the 3H threads read consecutive addresses (L2-resident: 1.9 MB shared by all
// workgroups); the H-vectors live in LDS.
// ------------------------------------------------------------------------------------------------
#define GRU_MAXL 8
// weight loads of each matrix in flight per thread: the gate loops are L2-latency bound (16 in flight: eight round trips per layer, 40 us for
// five layers; 32: four).  The summation order stays k ascending per accumulator, so the results do not depend on it.
constexpr int GRU_INFLIGHT = 32;
__global__ __launch_bounds__(384) void gru_step_kernel(int b, int layers, int hidden, const float *__restrict__ x,
                                                       const float *__restrict__ h_in, const float *__restrict__ w_ih,
                                                       const float *__restrict__ w_hh, const float *__restrict__ b_ih,
                                                       const float *__restrict__ b_hh, float *__restrict__ h_out,
                                                       float *__restrict__ y, const float *__restrict__ head_wt,
                                                       const float *__restrict__ head_bias, float *__restrict__ head_out, int head_cout) {
    __shared__ float s_x[128], s_h[128], s_gi[384], s_gh[384];
    const int s = blockIdx.x, t = threadIdx.x, H = hidden;
    if (t < H) s_x[t] = x[(long)s * H + t];
    for (int l = 0; l < layers; ++l) {
        if (t < H) s_h[t] = h_in[((long)l * b + s) * H + t];
        __syncthreads();
        if (t < 3 * H) {
            // transposed weights (L, H, 3H): consecutive threads read consecutive addresses
            const float *wi = w_ih + (long)l * H * 3 * H + t;
            const float *wh = w_hh + (long)l * H * 3 * H + t;
            // 16 weight loads of each matrix in flight per thread (the loop is L2-latency bound otherwise); the summation
            // order is still k ascending per accumulator pair, combined once at the end
            float ai = b_ih[l * 3 * H + t], ah = b_hh[l * 3 * H + t];
            for (int k0 = 0; k0 < H; k0 += GRU_INFLIGHT) {      // (H % GRU_INFLIGHT == 0, checked by the launcher)
                float wv[GRU_INFLIGHT], uv[GRU_INFLIGHT];
#pragma unroll
                for (int q = 0; q < GRU_INFLIGHT; ++q) {
                    wv[q] = wi[(long)(k0 + q) * 3 * H];
                    uv[q] = wh[(long)(k0 + q) * 3 * H];
                }
#pragma unroll
                for (int q = 0; q < GRU_INFLIGHT; ++q) {
                    ai = fmaf(wv[q], s_x[k0 + q], ai);
                    ah = fmaf(uv[q], s_h[k0 + q], ah);
                }
            }
            s_gi[t] = ai;
            s_gh[t] = ah;
        }
        __syncthreads();
        if (t < H) {
            const float r = 1.f / (1.f + expf(-(s_gi[t] + s_gh[t])));
            const float z = 1.f / (1.f + expf(-(s_gi[H + t] + s_gh[H + t])));
            const float nn = tanhf(s_gi[2 * H + t] + r * s_gh[2 * H + t]);
            const float hn = (1.f - z) * nn + z * s_h[t];
            h_out[((long)l * b + s) * H + t] = hn;
            s_x[t] = hn;                      // input of the next layer
            if (l == layers - 1) y[(long)s * H + t] = hn;
        }
        __syncthreads();
    }
    // optional epilogue: head_out[s] = head_w y[s] + head_bias (head_wt = the weight transposed, (H, head_cout)) -- the per-sample bias
    // the flow head's first layer takes from the GRU output (model_utils.py:297-300: cat of the broadcast GRU feature), a launch of
    // its own for 64 rows until round 6.  s_x holds y.
    if (head_wt) {
        for (int c = t; c < head_cout; c += blockDim.x) {
            float acc = head_bias ? head_bias[c] : 0.f;
            for (int k0 = 0; k0 < H; k0 += 16) {
                float wv[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) wv[q] = head_wt[(long)(k0 + q) * head_cout + c];
#pragma unroll
                for (int q = 0; q < 16; ++q) acc = fmaf(wv[q], s_x[k0 + q], acc);
            }
            head_out[(long)s * head_cout + c] = acc;
        }
    }
}

extern "C" int rtk_gru_step(int b, int layers, int hidden, const float *x, const float *h_in, const float *w_ih,
                            const float *w_hh, const float *b_ih, const float *b_hh, float *h_out, float *y,
                            rtk_stream_t stream) {
    RTK_REQUIRE(b > 0 && layers > 0 && layers <= GRU_MAXL && hidden > 0 && hidden <= 128 && hidden % GRU_INFLIGHT == 0 && x && h_in && w_ih &&
                w_hh && b_ih && b_hh && h_out && y, "gru_step: bad arguments (hidden=%d, layers=%d)", hidden, layers);
    gru_step_kernel<<<b, 384, 0, (hipStream_t)stream>>>(b, layers, hidden, x, h_in, w_ih, w_hh, b_ih, b_hh, h_out, y, nullptr, nullptr, nullptr, 0);
    RTK_CHECK_LAUNCH("gru_step");
    return RTK_OK;
}

extern "C" int rtk_gru_step_head(int b, int layers, int hidden, const float *x, const float *h_in, const float *w_ih,
                                 const float *w_hh, const float *b_ih, const float *b_hh, float *h_out, float *y, const float *head_wt,
                                 const float *head_bias, float *head_out, int head_cout, rtk_stream_t stream) {
    RTK_REQUIRE(b > 0 && layers > 0 && layers <= GRU_MAXL && hidden > 0 && hidden <= 128 && hidden % GRU_INFLIGHT == 0 && x && h_in && w_ih &&
                w_hh && b_ih && b_hh && h_out && y && head_wt && head_out && head_cout > 0,
                "gru_step_head: bad arguments (hidden=%d, layers=%d, head_cout=%d)", hidden, layers, head_cout);
    gru_step_kernel<<<b, 384, 0, (hipStream_t)stream>>>(b, layers, hidden, x, h_in, w_ih, w_hh, b_ih, b_hh, h_out, y, head_wt, head_bias,
                                                        head_out, head_cout);
    RTK_CHECK_LAUNCH("gru_step_head");
    return RTK_OK;
}

// ------------------------------------------------------------------------------------------------
// rtk_global_terms: everything that is a function of a sample's global (max-pooled) feature g (samples, cin), in one launch
// (models/track4d.py:89-95 broadcasts it over the points and concatenates; here a concatenated global half of a layer's input is a
// per-sample bias of that layer): up to RTK_GT_MAX_JOBS linear maps out_j[s - s0_j] = W_j g[s] + b_j over sample ranges, and the
// broadcast of g[s] over the n rows of sample s of a point-major tensor (the global half of pc{1,2}_features).  One workgroup per
// sample.  (Until round 6: three 64-row launches of the per-point MLP kernel + a framework broadcast copy.)
// ------------------------------------------------------------------------------------------------
struct GtParams {
    int samples, cin, njobs, n, bcast_pitch;
    const float *g;
    float *bcast;
    rtk_gterm_job_t job[RTK_GT_MAX_JOBS];
};

__global__ __launch_bounds__(256) void global_terms_kernel(const GtParams P) {
    __shared__ __attribute__((aligned(16))) float s_g[512];
    const int s = blockIdx.x, t = threadIdx.x;
    for (int k = t; k < P.cin; k += 256) s_g[k] = P.g[(long)s * P.cin + k];
    __syncthreads();
    // the outputs of all of this sample's jobs as one list, spread over the sample's workgroups, a thread per output (k ascending), 64 weight
    // loads in flight per thread: the launch is a few L2 round trips long
    int total = 0;
    for (int j = 0; j < P.njobs; ++j)
        if (s >= P.job[j].s0 && s < P.job[j].s0 + P.job[j].count) total += P.job[j].cout;
    for (int o = blockIdx.y * 256 + t; o < total; o += 256 * gridDim.y) {
        int c = o, j = 0;
        for (; j < P.njobs; ++j) {
            if (s < P.job[j].s0 || s >= P.job[j].s0 + P.job[j].count) continue;
            if (c < P.job[j].cout) break;
            c -= P.job[j].cout;
        }
        const rtk_gterm_job_t &J = P.job[j];
        float acc = J.bias ? J.bias[c] : 0.f;
        for (int k0 = 0; k0 < P.cin; k0 += 64) {          // cin % 32 == 0
            float wv[64];
#pragma unroll
            for (int q = 0; q < 64; ++q) wv[q] = J.wt[(long)min(k0 + q, P.cin - 1) * J.cout + c];
#pragma unroll
            for (int q = 0; q < 64; ++q)
                if (k0 + q < P.cin) acc = fmaf(wv[q], s_g[k0 + q], acc);
        }
        J.out[(long)(s - J.s0) * J.out_pitch + c] = acc;
    }
    if (P.bcast) {      // a thread owns one float4 of the row, rows strided by 1024 / cin
        const int per_row = P.cin >> 2, rows_per_iter = 256 / per_row, q = t % per_row;
        if (t < rows_per_iter * per_row) {
            const float4 v = *reinterpret_cast<const float4 *>(s_g + 4 * q);
            for (int r = blockIdx.y * rows_per_iter + t / per_row; r < P.n; r += rows_per_iter * gridDim.y)
                *reinterpret_cast<float4 *>(P.bcast + ((long)s * P.n + r) * P.bcast_pitch + 4 * q) = v;
        }
    }
}

extern "C" int rtk_global_terms(int samples, int cin, const float *g, int njobs, const rtk_gterm_job_t *jobs, float *bcast, int bcast_pitch,
                                int n, rtk_stream_t stream) {
    RTK_REQUIRE(samples > 0 && cin > 0 && cin <= 512 && cin % 32 == 0 && g && njobs >= 0 && njobs <= RTK_GT_MAX_JOBS && (njobs == 0 || jobs),
                "global_terms: bad arguments (samples=%d cin=%d njobs=%d)", samples, cin, njobs);
    RTK_REQUIRE(!bcast || (n > 0 && bcast_pitch >= cin && bcast_pitch % 4 == 0 && cin <= 1024), "global_terms: bad broadcast target");
    GtParams P;
    P.samples = samples; P.cin = cin; P.njobs = njobs; P.n = n; P.bcast_pitch = bcast_pitch; P.g = g; P.bcast = bcast;
    for (int j = 0; j < njobs; ++j) {
        RTK_REQUIRE(jobs[j].wt && jobs[j].out && jobs[j].cout > 0 && jobs[j].s0 >= 0 && jobs[j].count >= 0 && jobs[j].s0 + jobs[j].count <= samples &&
                    jobs[j].out_pitch >= jobs[j].cout, "global_terms: bad job %d", j);
        P.job[j] = jobs[j];
    }
    global_terms_kernel<<<dim3(samples, bcast ? 8 : 2), 256, 0, (hipStream_t)stream>>>(P);
    RTK_CHECK_LAUNCH("global_terms");
    return RTK_OK;
}

// ------------------------------------------------------------------------------------------------
// rtk_copy_multi: up to RTK_COPY_MAX_JOBS device-to-device copies in one launch (the inputs of a captured step into its static
// buffers: the framework's multi-tensor copy was 18 us for 0.8 MB).  Sizes in bytes, multiples of 4; 16-byte path when aligned.
// ------------------------------------------------------------------------------------------------
struct CopyParams {
    int njobs;
    rtk_copy_job_t job[RTK_COPY_MAX_JOBS];
    long first_block[RTK_COPY_MAX_JOBS + 1];
};
constexpr int COPY_BLOCK_BYTES = 256 * 16 * 4;      // 16 KiB per workgroup

__global__ __launch_bounds__(256) void copy_multi_kernel(const CopyParams P) {
    const long blk = blockIdx.x;
    int j = 0;
#pragma unroll
    for (int q = 1; q < RTK_COPY_MAX_JOBS; ++q) j += (q < P.njobs && blk >= P.first_block[q]) ? 1 : 0;
    const rtk_copy_job_t &J = P.job[j];
    const long off = (blk - P.first_block[j]) * COPY_BLOCK_BYTES;
    const char *src = reinterpret_cast<const char *>(J.src) + off;
    char *dst = reinterpret_cast<char *>(J.dst) + off;
    const long left = J.bytes - off;
    if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long o = ((long)u * 256 + threadIdx.x) * 16;
            if (o + 16 <= left) *reinterpret_cast<float4 *>(dst + o) = *reinterpret_cast<const float4 *>(src + o);
            else for (long b = o; b < left && b < o + 16; b += 4) *reinterpret_cast<float *>(dst + b) = *reinterpret_cast<const float *>(src + b);
        }
    } else {
        for (long o = (long)threadIdx.x * 4; o < left && o < COPY_BLOCK_BYTES; o += 256 * 4)
            *reinterpret_cast<float *>(dst + o) = *reinterpret_cast<const float *>(src + o);
    }
}

extern "C" int rtk_copy_multi(int njobs, const rtk_copy_job_t *jobs, rtk_stream_t stream) {
    RTK_REQUIRE(njobs > 0 && njobs <= RTK_COPY_MAX_JOBS && jobs, "copy_multi: %d jobs (1..%d)", njobs, RTK_COPY_MAX_JOBS);
    CopyParams P;
    P.njobs = njobs;
    long blocks = 0;
    for (int j = 0; j < njobs; ++j) {
        RTK_REQUIRE(jobs[j].src && jobs[j].dst && jobs[j].bytes > 0 && jobs[j].bytes % 4 == 0 &&
                    (((uintptr_t)jobs[j].src | (uintptr_t)jobs[j].dst) & 3) == 0, "copy_multi: bad job %d", j);
        P.job[j] = jobs[j];
        P.first_block[j] = blocks;
        blocks += (jobs[j].bytes + COPY_BLOCK_BYTES - 1) / COPY_BLOCK_BYTES;
    }
    P.first_block[njobs] = blocks;
    RTK_REQUIRE(blocks < 0x7fffffffL, "copy_multi: too large");
    copy_multi_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(P);
    RTK_CHECK_LAUNCH("copy_multi");
    return RTK_OK;
}

// Backward of rtk_gru_step.  Samples are independent, so one workgroup walks one sample down the layer stack: gates are
// recomputed from (x_l, h_in_l) exactly as in the forward, the gate gradients dgi / dgh (L,B,3H) are stored for the weight
// gradients (two batched GEMMs on the host side: dW_ih[l] = dgi[l]^T x_l, dW_hh[l] = dgh[l]^T h_in[l]) and propagated through
// W_ih^T / W_hh^T (original (L,3H,H) layout: consecutive threads read consecutive addresses).
//   r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r gh_n), h' = (1 - z) n + z h
__global__ __launch_bounds__(384) void gru_step_bwd_kernel(int b, int layers, int hidden, const float *__restrict__ x,
                                                           const float *__restrict__ h_in, const float *__restrict__ h_out,
                                                           const float *__restrict__ w_ih_t, const float *__restrict__ w_hh_t,
                                                           const float *__restrict__ w_ih, const float *__restrict__ w_hh,
                                                           const float *__restrict__ b_ih, const float *__restrict__ b_hh,
                                                           const float *__restrict__ dy, const float *__restrict__ dh_out,
                                                           float *__restrict__ dx, float *__restrict__ dh_in, float *__restrict__ dgi,
                                                           float *__restrict__ dgh) {
    __shared__ float s_x[128], s_h[128], s_gi[384], s_gh[384], s_dx[3][128], s_dh[3][128], s_carry[128], s_direct[128];
    const int s = blockIdx.x, t = threadIdx.x, H = hidden;
    if (t < H) s_carry[t] = dy[(long)s * H + t];
    for (int l = layers - 1; l >= 0; --l) {
        if (t < H) {
            s_x[t] = l == 0 ? x[(long)s * H + t] : h_out[((long)(l - 1) * b + s) * H + t];
            s_h[t] = h_in[((long)l * b + s) * H + t];
        }
        __syncthreads();
        if (t < 3 * H) {
            const float *wi = w_ih_t + (long)l * H * 3 * H + t;
            const float *wh = w_hh_t + (long)l * H * 3 * H + t;
            // 16 weight loads of each matrix in flight per thread (the loop is L2-latency bound otherwise); the summation
            // order is still k ascending per accumulator pair, combined once at the end
            float ai = b_ih[l * 3 * H + t], ah = b_hh[l * 3 * H + t];
            for (int k0 = 0; k0 < H; k0 += GRU_INFLIGHT) {      // (H % GRU_INFLIGHT == 0, checked by the launcher)
                float wv[GRU_INFLIGHT], uv[GRU_INFLIGHT];
#pragma unroll
                for (int q = 0; q < GRU_INFLIGHT; ++q) {
                    wv[q] = wi[(long)(k0 + q) * 3 * H];
                    uv[q] = wh[(long)(k0 + q) * 3 * H];
                }
#pragma unroll
                for (int q = 0; q < GRU_INFLIGHT; ++q) {
                    ai = fmaf(wv[q], s_x[k0 + q], ai);
                    ah = fmaf(uv[q], s_h[k0 + q], ah);
                }
            }
            s_gi[t] = ai;
            s_gh[t] = ah;
        }
        __syncthreads();
        float g_r = 0.f, g_z = 0.f, g_n = 0.f, g_nr = 0.f;
        if (t < H) {
            const float r = 1.f / (1.f + expf(-(s_gi[t] + s_gh[t])));
            const float z = 1.f / (1.f + expf(-(s_gi[H + t] + s_gh[H + t])));
            const float hn = s_gh[2 * H + t];
            const float nn = tanhf(s_gi[2 * H + t] + r * hn);
            const float dhp = s_carry[t] + (dh_out ? dh_out[((long)l * b + s) * H + t] : 0.f);
            const float dn = dhp * (1.f - z) * (1.f - nn * nn);
            g_n = dn;
            g_nr = dn * r;
            g_r = dn * hn * r * (1.f - r);
            g_z = dhp * (s_h[t] - nn) * z * (1.f - z);
            s_direct[t] = dhp * z;
        }
        __syncthreads();        // every thread has read s_gi / s_gh
        if (t < H) {
            s_gi[t] = g_r; s_gi[H + t] = g_z; s_gi[2 * H + t] = g_n;
            s_gh[t] = g_r; s_gh[H + t] = g_z; s_gh[2 * H + t] = g_nr;
        }
        __syncthreads();
        if (t < 3 * H) {
            dgi[((long)l * b + s) * 3 * H + t] = s_gi[t];
            dgh[((long)l * b + s) * 3 * H + t] = s_gh[t];
        }
        {   // dx = W_ih^T dgi, dh = W_hh^T dgh: thread (part, j) sums a third of the 3H rows
            const int part = t / H, j = t % H;
            if (part < 3) {
                const float *wi = w_ih + ((long)l * 3 * H + (long)part * H) * H + j;
                const float *wh = w_hh + ((long)l * 3 * H + (long)part * H) * H + j;
                float ax = 0.f, ah = 0.f;
                for (int i0 = 0; i0 < H; i0 += 16) {
                    float wv[16], uv[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        wv[q] = wi[(long)(i0 + q) * H];
                        uv[q] = wh[(long)(i0 + q) * H];
                    }
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        ax = fmaf(wv[q], s_gi[part * H + i0 + q], ax);
                        ah = fmaf(uv[q], s_gh[part * H + i0 + q], ah);
                    }
                }
                s_dx[part][j] = ax;
                s_dh[part][j] = ah;
            }
        }
        __syncthreads();
        if (t < H) {
            const float gx = s_dx[0][t] + s_dx[1][t] + s_dx[2][t];
            dh_in[((long)l * b + s) * H + t] = s_direct[t] + (s_dh[0][t] + s_dh[1][t] + s_dh[2][t]);
            s_carry[t] = gx;
            if (l == 0) dx[(long)s * H + t] = gx;
        }
        __syncthreads();
    }
}

extern "C" int rtk_gru_step_bwd(int b, int layers, int hidden, const float *x, const float *h_in, const float *h_out,
                                const float *w_ih_t, const float *w_hh_t, const float *w_ih, const float *w_hh, const float *b_ih,
                                const float *b_hh, const float *dy, const float *dh_out, float *dx, float *dh_in, float *dgi, float *dgh,
                                rtk_stream_t stream) {
    RTK_REQUIRE(b > 0 && layers > 0 && layers <= GRU_MAXL && hidden == 128 && x && h_in && h_out && w_ih_t && w_hh_t && w_ih && w_hh &&
                b_ih && b_hh && dy && dx && dh_in && dgi && dgh, "gru_step_bwd: bad arguments (hidden=%d must be 128, layers=%d)", hidden, layers);
    gru_step_bwd_kernel<<<b, 384, 0, (hipStream_t)stream>>>(b, layers, hidden, x, h_in, h_out, w_ih_t, w_hh_t, w_ih, w_hh, b_ih, b_hh, dy,
                                                            dh_out, dx, dh_in, dgi, dgh);
    RTK_CHECK_LAUNCH("gru_step_bwd");
    return RTK_OK;
}

// ------------------------------------------------------------------------------------------------
// rtk_log_sinkhorn: the 500-iteration log-space Sinkhorn normalisation of the object association
// (log_optimal_transport / log_sinkhorn_iterations, models/utils/track4d_utils.py:405-434) in ONE launch.
// The reference issues ~8 framework kernels per iteration on an (m+1) x (n+1) matrix of a few hundred entries: 4000
// launches, 40-90 ms per radar frame -- of a 100 ms frame period.  Here one workgroup keeps the score matrix with its
// dustbin row / column in LDS and alternates row and column log-sum-exp passes (thread = row, then thread = column).
//   couplings = [[scores, alpha], [alpha, alpha]],  norm = -log(m + n),
//   log_mu = (norm, ..., norm, log n + norm),  log_nu = (norm, ..., norm, log m + norm),  u = v = 0
//   iters x { u = log_mu - lse_j(Z + v);  v = log_nu - lse_i(Z + u) };   out = Z + u + v - norm
// logsumexp is evaluated as torch does: max + log(sum(exp(x - max))) (tree-summed across the lanes of a wave).
// ------------------------------------------------------------------------------------------------
// max / sum over the 16 lanes of a DPP row, result in every lane (s_nop 1: a VGPR written by VALU needs 2 wait states
// before a DPP read)
__device__ __forceinline__ float row16_max(float v) {
    asm volatile("s_nop 1\n v_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                 "v_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                 "v_max_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                 "v_max_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                 : "+v"(v));
    return v;
}

__device__ __forceinline__ float row16_sum(float v) {
    asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                 "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                 "v_add_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                 "v_add_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                 : "+v"(v));
    return v;
}

__global__ __launch_bounds__(256) void log_sinkhorn_kernel(int m, int n, const float *__restrict__ scores, float alpha, int iters,
                                                           float *__restrict__ out) {
    extern __shared__ float s_mem[];
    const int R = m + 1, C = n + 1, ld = C | 1;           // odd row stride: the column pass walks down a column conflict-free
    float *Z = s_mem, *u = Z + R * ld, *v = u + R;
    const int t = threadIdx.x;
    for (int e = t; e < R * C; e += 256) {
        const int i = e / C, j = e % C;
        Z[i * ld + j] = (i < m && j < n) ? scores[i * n + j] : alpha;
    }
    for (int e = t; e < R; e += 256) u[e] = 0.f;
    for (int e = t; e < C; e += 256) v[e] = 0.f;
    const float norm = -logf((float)m + (float)n);
    __syncthreads();
    // One 16-lane DPP row per matrix row (then per column), its lanes across the other axis: max and sum are 4 rotate-and-
    // combine DPP steps (a ds_bpermute shuffle chain costs ~60 cycles per step; 500 iterations x 2 phases x 12 steps of it
    // were 3 ms).  16 rows per pass over the workgroup.
    const int grp = t >> 4, c = t & 15;
    const float lmu_last = logf((float)n) + norm, lnu_last = logf((float)m) + norm;
    for (int it = 0; it < iters; ++it) {
        for (int i0 = 0; i0 < R; i0 += 16) {
            const int i = i0 + grp;
            const bool live = i < R;
            float mx = -INFINITY;
            if (live) for (int j = c; j < C; j += 16) mx = fmaxf(mx, Z[i * ld + j] + v[j]);
            mx = row16_max(mx);
            float sum = 0.f;
            if (live) for (int j = c; j < C; j += 16) sum += expf(Z[i * ld + j] + v[j] - mx);
            sum = row16_sum(sum);
            if (live && c == 0) u[i] = (i < m ? norm : lmu_last) - (logf(sum) + mx);
        }
        __syncthreads();
        for (int j0 = 0; j0 < C; j0 += 16) {
            const int j = j0 + grp;
            const bool live = j < C;
            float mx = -INFINITY;
            if (live) for (int i = c; i < R; i += 16) mx = fmaxf(mx, Z[i * ld + j] + u[i]);
            mx = row16_max(mx);
            float sum = 0.f;
            if (live) for (int i = c; i < R; i += 16) sum += expf(Z[i * ld + j] + u[i] - mx);
            sum = row16_sum(sum);
            if (live && c == 0) v[j] = (j < n ? norm : lnu_last) - (logf(sum) + mx);
        }
        __syncthreads();
    }
    for (int e = t; e < R * C; e += 256) {
        const int i = e / C, j = e % C;
        out[e] = Z[i * ld + j] + u[i] + v[j] - norm;
    }
}

extern "C" int rtk_log_sinkhorn(int m, int n, const float *scores, float alpha, int iters, float *out, rtk_stream_t stream) {
    RTK_REQUIRE(m > 0 && n > 0 && iters >= 0 && scores && out, "log_sinkhorn: bad arguments");
    const size_t lds = ((size_t)(m + 1) * ((n + 1) | 1) + (m + 1) + (n + 1)) * sizeof(float);
    RTK_REQUIRE(lds <= 64 * 1024, "log_sinkhorn: %d x %d objects exceed the single-workgroup LDS budget", m, n);
    log_sinkhorn_kernel<<<1, 256, lds, (hipStream_t)stream>>>(m, n, scores, alpha, iters, out);
    RTK_CHECK_LAUNCH("log_sinkhorn");
    return RTK_OK;
}

// ------------------------------------------------------------------------------------------------
// rtk_dbscan: clustering of the moving points (models/track4d.py:108-126: sklearn.cluster.DBSCAN(eps, min_samples) on the
// host, behind a device->host copy of the features and a boolean-mask gather).  One workgroup:
//   1. ordered compaction of the movers (score > threshold) and their D feature channels into LDS;
//   2. core points: closed eps-ball (itself included) holds >= min_samples movers.  Distances in float64 with numpy's
//      pairwise summation order, sqrt(d2) <= eps -- the arithmetic of ratrack_amd/association.dbscan;
//   3. connected components of the core points under the eps-graph by min-label propagation with pointer jumping;
//   4. sklearn numbers clusters in order of their first core point and fully expands one cluster before starting the next,
//      so: cluster id = rank of the component's smallest core index; a border point (non-core, within eps of a core
//      point; only possible for min_samples > 2) joins the lowest-numbered cluster among its core neighbours; the rest is
//      noise (-1).
// labels (n) int32: cluster id of every INPUT point, -1 for noise and for non-movers.
// ------------------------------------------------------------------------------------------------
#define DB_D 8

__device__ __forceinline__ bool db_adjacent(const float *f, int i, int j, double eps) {
    double q[DB_D];
#pragma unroll
    for (int c = 0; c < DB_D; ++c) {
        const double d = (double)f[i * DB_D + c] - (double)f[j * DB_D + c];
        q[c] = d * d;
    }
    const double d2 = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));      // numpy's 8-wide pairwise sum
    return __dsqrt_rn(d2) <= eps;
}

// WORK = false: the tables live in LDS (clouds up to ~2900 points: every real frame).  WORK = true: the same single workgroup on a
// global-memory workspace -- larger clouds are clustered on the device too (slower: O(m^2) distance tests from L2 instead of LDS),
// nothing falls back to the host.
template <bool WORK>
__global__ __launch_bounds__(256) void dbscan_kernel(int n, const float *__restrict__ feat, int pitch, const int *__restrict__ chan,
                                                     const float *__restrict__ score, float thr, double eps, int min_samples,
                                                     int *__restrict__ labels, unsigned char *__restrict__ work) {
    extern __shared__ __attribute__((aligned(16))) unsigned char db_smem[];
    float *f = WORK ? reinterpret_cast<float *>(work) : reinterpret_cast<float *>(db_smem);                 // (m, 8) compacted features
    int *src = reinterpret_cast<int *>(f + (size_t)n * DB_D);      // mover -> input index
    int *lab = src + n;                                            // component label (smallest core index) or INT_MAX
    int *aux = lab + n;                                            // core flag, then cluster number of a representative
    __shared__ int s_m, s_changed, s_wave[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // ---- 1. ordered compaction ----------------------------------------------------------------------------------------
    if (t == 0) s_m = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 256) {
        const int i = base + t;
        const bool mv = i < n && score[i] > thr;
        const unsigned long long bal = __ballot(mv);
        if (lane == 0) s_wave[wave] = __popcll(bal);
        __syncthreads();
        int off = s_m;
        for (int w = 0; w < wave; ++w) off += s_wave[w];
        if (mv) {
            const int k = off + __popcll(bal & ((1ull << lane) - 1ull));
            src[k] = i;
#pragma unroll
            for (int c = 0; c < DB_D; ++c) f[k * DB_D + c] = feat[(size_t)chan[c] * pitch + i];
        }
        __syncthreads();
        if (t == 0) s_m += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
    const int m = s_m;
    for (int i = t; i < n; i += 256) labels[i] = -1;
    // ---- 2. core points -----------------------------------------------------------------------------------------------
    for (int i = t; i < m; i += 256) {
        int cnt = 0;
        for (int j = 0; j < m; ++j) cnt += db_adjacent(f, i, j, eps) ? 1 : 0;
        aux[i] = cnt >= min_samples;
        lab[i] = cnt >= min_samples ? i : 0x7fffffff;
    }
    __syncthreads();
    // ---- 3. components of the core graph ----------------------------------------------------------------------------------
    for (;;) {
        if (t == 0) s_changed = 0;
        __syncthreads();
        for (int i = t; i < m; i += 256) {
            if (!aux[i]) continue;
            int best = lab[i];
            for (int j = 0; j < m; ++j)
                if (aux[j] && lab[j] < best && db_adjacent(f, i, j, eps)) best = lab[j];
            if (best < lab[i]) { lab[i] = best; s_changed = 1; }       // racy reads of lab[j] only ever see smaller, valid labels
        }
        __syncthreads();
        for (int i = t; i < m; i += 256)                               // pointer jumping: label of my label
            if (aux[i]) { const int l = lab[lab[i]]; if (l < lab[i]) lab[i] = l; }
        __syncthreads();
        if (!s_changed) break;
        __syncthreads();
    }
    // ---- 4. cluster numbers, border points, output -------------------------------------------------------------------------
    for (int i = t; i < m; i += 256) {           // representative i (lab[i] == i): its number = representatives before it
        if (aux[i] && lab[i] == i) {
            int r = 0;
            for (int j = 0; j < i; ++j) r += (aux[j] && lab[j] == j) ? 1 : 0;
            aux[i] = 2 + r;                      // >= 2 marks "core + number"; plain core points keep 1
        }
    }
    __syncthreads();
    for (int i = t; i < m; i += 256) {
        int out = -1;
        if (aux[i]) {
            out = aux[lab[i]] - 2;
        } else if (min_samples > 2) {
            for (int j = 0; j < m; ++j)
                if (aux[j] && db_adjacent(f, i, j, eps)) {
                    const int c = aux[lab[j]] - 2;
                    out = (out < 0 || c < out) ? c : out;
                }
        }
        labels[src[i]] = out;
    }
}

extern "C" int rtk_dbscan(int n, const float *feat, int pitch, const int *channels, const float *score, float threshold, double eps,
                          int min_samples, int *labels, rtk_stream_t stream) {
    RTK_REQUIRE(n > 0 && feat && channels && score && labels && pitch >= n && min_samples >= 1, "dbscan: bad arguments");
    RTK_REQUIRE(n <= 65536, "dbscan: n=%d > 65536 points (one workgroup tests all pairs)", n);
    const size_t bytes = (size_t)n * (DB_D * sizeof(float) + 3 * sizeof(int));
    hipStream_t st = (hipStream_t)stream;
    if (bytes <= 128 * 1024) {
        (void)hipFuncSetAttribute((const void *)dbscan_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);      // per device and cheap: every call
        dbscan_kernel<false><<<1, 256, bytes, st>>>(n, feat, pitch, channels, score, threshold, eps, min_samples, labels, nullptr);
    } else {      // beyond one workgroup's LDS: a stream-ordered workspace
        void *work = nullptr;
        if (hipMallocAsync(&work, bytes, st) != hipSuccess) {
            (void)hipGetLastError();
            rtk_set_error("dbscan: no %zu-byte workspace for n=%d points", bytes, n);
            return RTK_ERR_LAUNCH;
        }
        dbscan_kernel<true><<<1, 256, 0, st>>>(n, feat, pitch, channels, score, threshold, eps, min_samples, labels, (unsigned char *)work);
        (void)hipFreeAsync(work, st);
    }
    RTK_CHECK_LAUNCH("dbscan");
    return RTK_OK;
}

// ------------------------------------------------------------------------------------------------
// rtk_to_channel_major: 32x32 tiles through LDS so that both the point-major reads and the
// channel-major writes are coalesced.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void to_channel_major_kernel(int n, int channels, const float *__restrict__ src, int src_pitch,
                                                               int per_sample, float *__restrict__ dst, int dst_channels,
                                                               int dst_off) {
    __shared__ float tile[32][33];
    const int s = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        float v = 0.f;
        if (p < n && c < channels) v = src[(per_sample ? (long)s : (long)s * n + p) * src_pitch + c];
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        if (p < n && c < channels) dst[((long)s * dst_channels + dst_off + c) * n + p] = tile[tx][r];
    }
}

extern "C" int rtk_to_channel_major(int samples, int n, int channels, const float *src, int src_pitch, int per_sample,
                                    float *dst, int dst_channels, int dst_channel_offset, rtk_stream_t stream) {
    RTK_REQUIRE(samples > 0 && n > 0 && channels > 0 && src && dst && src_pitch >= channels && dst_channel_offset >= 0 &&
                dst_channel_offset + channels <= dst_channels && samples <= 65535, "to_channel_major: bad arguments");
    dim3 grid(rtk_divup(n, 32), rtk_divup(channels, 32), samples);
    to_channel_major_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(n, channels, src, src_pitch, per_sample, dst, dst_channels,
                                                                    dst_channel_offset);
    RTK_CHECK_LAUNCH("to_channel_major");
    return RTK_OK;
}

// ------------------------------------------------------------------------------------------------
// rtk_to_channel_major_multi: the six layout conversions of one backbone() in a single launch (job = blockIdx.z / samples)
// ------------------------------------------------------------------------------------------------
#define LAYOUT_MAX_JOBS 8
struct LayoutJobs {
    rtk_layout_job_t job[LAYOUT_MAX_JOBS];
    int samples, n;
};

__global__ __launch_bounds__(256) void to_channel_major_multi_kernel(const LayoutJobs J) {
    __shared__ float tile[32][33];
    const int jid = blockIdx.z / J.samples, s = blockIdx.z % J.samples;
    const rtk_layout_job_t job = J.job[jid];
    const int c0 = blockIdx.y * 32, p0 = blockIdx.x * 32, n = J.n;
    if (c0 >= job.channels) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        float v = 0.f;
        if (p < n && c < job.channels) v = job.src[(job.per_sample ? (long)s : (long)s * n + p) * job.src_pitch + c];
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        if (p < n && c < job.channels) job.dst[((long)s * job.dst_channels + job.dst_channel_offset + c) * n + p] = tile[tx][r];
    }
}

extern "C" int rtk_to_channel_major_multi(int samples, int n, int njobs, const rtk_layout_job_t *jobs, rtk_stream_t stream) {
    RTK_REQUIRE(samples > 0 && n > 0 && njobs > 0 && njobs <= LAYOUT_MAX_JOBS && jobs && (long)samples * njobs <= 65535,
                "to_channel_major_multi: bad arguments (njobs=%d)", njobs);
    LayoutJobs J;
    int cmax = 0;
    for (int i = 0; i < njobs; ++i) {
        RTK_REQUIRE(jobs[i].src && jobs[i].dst && jobs[i].channels > 0 && jobs[i].src_pitch >= jobs[i].channels &&
                    jobs[i].dst_channel_offset + jobs[i].channels <= jobs[i].dst_channels, "to_channel_major_multi: bad job %d", i);
        J.job[i] = jobs[i];
        if (jobs[i].channels > cmax) cmax = jobs[i].channels;
    }
    J.samples = samples; J.n = n;
    dim3 grid(rtk_divup(n, 32), rtk_divup(cmax, 32), samples * njobs);
    to_channel_major_multi_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(J);
    RTK_CHECK_LAUNCH("to_channel_major_multi");
    return RTK_OK;
}
