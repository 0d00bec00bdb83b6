// train_misc.hip -- small operators of the training path that exist to keep framework glue out of the captured step.
//
// rtk_gmax_cat: the backbone's "append the cloud's global feature to every point" (models/track4d.py:92-95 of the reference:
//     g = torch.max(f, -1)[0].unsqueeze(2).expand(-1, -1, N);  features = torch.cat((f, g), dim=1))
// as one kernel forward (copy + max + broadcast) and one backward (sum over the points of the broadcast half, routed to the first
// arg-max, added to the gradient of the copied half).  The framework formulation is a reduction, a concatenation, and in the backward a
// sum, a zero fill, a scatter and an accumulation -- per frame.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rtk_common.h"
#include "rtk_train.h"

namespace {

// one wave per (sample, channel) row of n points
__global__ __launch_bounds__(256) void gmax_cat_fwd_kernel(int rows, int channels, int n, const float *__restrict__ f, float *__restrict__ out,
                                                           int *__restrict__ arg) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int b = row / channels, c = row - b * channels;
    const float *src = f + (size_t)row * n;
    float *lo = out + ((size_t)b * 2 * channels + c) * n, *hi = lo + (size_t)channels * n;
    float best = -INFINITY;
    int at = 0x7fffffff;
    for (int p = lane; p < n; p += 64) {
        const float v = src[p];
        lo[p] = v;
        if (v > best || (v == best && p < at)) { best = v; at = p; }      // (a NaN never wins: torch.max would propagate it; the path has none)
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float v2 = __shfl_xor(best, o, 64);
        const int a2 = __shfl_xor(at, o, 64);
        if (v2 > best || (v2 == best && a2 < at)) { best = v2; at = a2; }
    }
    for (int p = lane; p < n; p += 64) hi[p] = best;
    if (lane == 0) arg[row] = at;
}

__global__ __launch_bounds__(256) void gmax_cat_bwd_kernel(int rows, int channels, int n, const float *__restrict__ dout, const int *__restrict__ arg,
                                                           float *__restrict__ df) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int b = row / channels, c = row - b * channels;
    const float *lo = dout + ((size_t)b * 2 * channels + c) * n, *hi = lo + (size_t)channels * n;
    float s = 0.f;
    for (int p = lane; p < n; p += 64) s += hi[p];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const int at = arg[row];
    float *dst = df + (size_t)row * n;
    for (int p = lane; p < n; p += 64) dst[p] = lo[p] + (p == at ? s : 0.f);
}

}  // namespace

extern "C" int rtk_gmax_cat_fwd(int samples, int channels, int n, const float *f, float *out, int *arg, rtk_stream_t stream) {
    RTK_REQUIRE(samples > 0 && channels > 0 && n > 0 && f && out && arg, "rtk_gmax_cat_fwd: bad arguments");
    const int rows = samples * channels;
    gmax_cat_fwd_kernel<<<rtk_divup(rows, 4), 256, 0, (hipStream_t)stream>>>(rows, channels, n, f, out, arg);
    RTK_CHECK_LAUNCH("rtk_gmax_cat_fwd");
    return RTK_OK;
}

extern "C" int rtk_gmax_cat_bwd(int samples, int channels, int n, const float *dout, const int *arg, float *df, rtk_stream_t stream) {
    RTK_REQUIRE(samples > 0 && channels > 0 && n > 0 && dout && arg && df, "rtk_gmax_cat_bwd: bad arguments");
    const int rows = samples * channels;
    gmax_cat_bwd_kernel<<<rtk_divup(rows, 4), 256, 0, (hipStream_t)stream>>>(rows, channels, n, dout, arg, df);
    RTK_CHECK_LAUNCH("rtk_gmax_cat_bwd");
    return RTK_OK;
}

// ---- rtk_gru_pack_params -----------------------------------------------------------------------------------------------------------
// The GRU step kernels (rtk_gru_step / rtk_gru_step_bwd) take the L layers' weights stacked, plain and transposed.  In training the
// weights change every step: four torch.stack and two transposed copies per step, six launches -- this is one.
namespace {

constexpr int GP_MAXL = 8;
struct GruPack {
    const float *w_ih[GP_MAXL], *w_hh[GP_MAXL], *b_ih[GP_MAXL], *b_hh[GP_MAXL];
    float *o_ih, *o_ih_t, *o_hh, *o_hh_t, *o_bih, *o_bhh;
    int L, H;
};

__global__ __launch_bounds__(256) void gru_pack_kernel(const GruPack Q) {
    const int H = Q.H, G = 3 * H;
    const int per = G * H;                                     // elements of one matrix
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    const long nmat = (long)Q.L * 2 * per;
    if (e < nmat) {
        const int l = (int)(e / (2 * per)), rest = (int)(e - (long)l * 2 * per), m = rest / per, i = rest - m * per;
        const int r = i / H, c = i - r * H;                    // source (3H, H) row-major
        const float v = (m ? Q.w_hh[l] : Q.w_ih[l])[i];
        float *plain = m ? Q.o_hh : Q.o_ih, *tr = m ? Q.o_hh_t : Q.o_ih_t;
        plain[(size_t)l * per + i] = v;
        tr[(size_t)l * per + (size_t)c * G + r] = v;            // (L, H, 3H)
    } else {
        const long b = e - nmat;
        if (b < (long)Q.L * 2 * G) {
            const int l = (int)(b / (2 * G)), rest = (int)(b - (long)l * 2 * G), m = rest / G, i = rest - m * G;
            (m ? Q.o_bhh : Q.o_bih)[(size_t)l * G + i] = (m ? Q.b_hh[l] : Q.b_ih[l])[i];
        }
    }
}

}  // namespace

extern "C" int rtk_gru_pack_params(int layers, int hidden, const float *const *params, float *w_ih, float *w_ih_t, float *w_hh, float *w_hh_t,
                                   float *b_ih, float *b_hh, rtk_stream_t stream) {
    RTK_REQUIRE(layers >= 1 && layers <= GP_MAXL && hidden > 0 && params && w_ih && w_ih_t && w_hh && w_hh_t && b_ih && b_hh,
                "rtk_gru_pack_params: bad arguments (1..%d layers)", GP_MAXL);
    GruPack Q = {};
    Q.L = layers; Q.H = hidden;
    for (int l = 0; l < layers; ++l) {
        RTK_REQUIRE(params[4 * l] && params[4 * l + 1] && params[4 * l + 2] && params[4 * l + 3], "rtk_gru_pack_params: null parameter of layer %d", l);
        Q.w_ih[l] = params[4 * l]; Q.w_hh[l] = params[4 * l + 1]; Q.b_ih[l] = params[4 * l + 2]; Q.b_hh[l] = params[4 * l + 3];
    }
    Q.o_ih = w_ih; Q.o_ih_t = w_ih_t; Q.o_hh = w_hh; Q.o_hh_t = w_hh_t; Q.o_bih = b_ih; Q.o_bhh = b_hh;
    const long total = (long)layers * 2 * 3 * hidden * hidden + (long)layers * 2 * 3 * hidden;
    gru_pack_kernel<<<rtk_divup(total, 256), 256, 0, (hipStream_t)stream>>>(Q);
    RTK_CHECK_LAUNCH("rtk_gru_pack_params");
    return RTK_OK;
}

// ---- rtk_gru_wgrad ---------------------------------------------------------------------------------------------------------------
// Weight and bias gradients of the GRU step from the gate gradients rtk_gru_step_bwd emits:
//     dW_ih[l] = dgi[l]^T x_l,  dW_hh[l] = dgh[l]^T h_in[l],  db_ih[l] = column sums of dgi[l],  db_hh[l] = of dgh[l]
// (x_0 = x, x_l = h_out[l-1]).  2 L products of (3H x H) outputs over a contraction of only B rows: one thread per output element,
// the B-deep dot product straight from L2 -- a concatenation, two batched library GEMMs and a reduction otherwise.
namespace {

__global__ __launch_bounds__(256) void gru_wgrad_kernel(int B, int L, int H, const float *__restrict__ x, const float *__restrict__ h_in,
                                                        const float *__restrict__ h_out, const float *__restrict__ dgi,
                                                        const float *__restrict__ dgh, float *__restrict__ dw_ih, float *__restrict__ dw_hh,
                                                        float *__restrict__ db_ih, float *__restrict__ db_hh) {
    const int G = 3 * H;
    const long per = (long)G * (H + 1);                        // one matrix + its bias column
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)L * 2 * per) return;
    const int l = (int)(e / (2 * per)), m = (int)((e - (long)l * 2 * per) / per);
    const long i = e - (long)l * 2 * per - (long)m * per;
    const int c = (int)(i % (H + 1)), r = (int)(i / (H + 1));      // c == H: the bias
    const float *dg = (m ? dgh : dgi) + (size_t)l * B * G + r;    // (B, 3H): column r
    const float *in = m ? h_in + (size_t)l * B * H : (l == 0 ? x : h_out + (size_t)(l - 1) * B * H);
    float a0 = 0.f, a1 = 0.f;
    int b = 0;
    for (; b + 1 < B; b += 2) {
        a0 += dg[(size_t)b * G] * (c < H ? in[(size_t)b * H + c] : 1.f);
        a1 += dg[(size_t)(b + 1) * G] * (c < H ? in[(size_t)(b + 1) * H + c] : 1.f);
    }
    if (b < B) a0 += dg[(size_t)b * G] * (c < H ? in[(size_t)b * H + c] : 1.f);
    const float v = a0 + a1;
    if (c < H) (m ? dw_hh : dw_ih)[((size_t)l * G + r) * H + c] = v;
    else (m ? db_hh : db_ih)[(size_t)l * G + r] = v;
}

}  // namespace

extern "C" int rtk_gru_wgrad(int b, int layers, int hidden, const float *x, const float *h_in, const float *h_out, const float *dgi,
                             const float *dgh, float *dw_ih, float *dw_hh, float *db_ih, float *db_hh, rtk_stream_t stream) {
    RTK_REQUIRE(b > 0 && layers > 0 && hidden > 0 && x && h_in && h_out && dgi && dgh && dw_ih && dw_hh && db_ih && db_hh, "rtk_gru_wgrad: bad arguments");
    const long total = (long)layers * 2 * 3 * hidden * (hidden + 1);
    gru_wgrad_kernel<<<rtk_divup(total, 256), 256, 0, (hipStream_t)stream>>>(b, layers, hidden, x, h_in, h_out, dgi, dgh, dw_ih, dw_hh, db_ih, db_hh);
    RTK_CHECK_LAUNCH("rtk_gru_wgrad");
    return RTK_OK;
}
