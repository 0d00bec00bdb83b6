/*
 * rtk_train.h -- C ABI of the training-mode kernels of librtk_hip.so.
 *
 * Training-mode BatchNorm2d + ReLU (+ max-pool over the neighbourhood axis) of a SharedMLP layer
 * (lib/pytorch_utils.py:20-32,104-123 = Conv2d(1x1, no bias) -> nn.BatchNorm2d -> ReLU, followed in a
 * set-abstraction scale by F.max_pool2d over nsample, lib/pointnet2_modules.py:44-47) as ONE weighted
 * operator on the DE-DUPLICATED activation tensor:
 *
 *   z (samples, C, rows, ns) fp32, NCHW-contiguous, holds one row per UNIQUE centroid.  The reference tensor
 *   has npoint rows of which rows >= nuniq[b] are exact copies of row 0 (over-sampling FPS, see
 *   rtk_fps_centroids); their contribution to the batch statistics is carried by a per-row weight
 *   row_weight (samples, rows):  w[b][0] = 1 + npoint - nuniq[b],  w[b][r] = 1 for 0 < r < nuniq[b],  0 beyond.
 *   With count = samples/groups * npoint * ns (the reference element count per channel):
 *       mean = sum w z / count,   var = sum w (z - mean)^2 / count   (biased; running_var gets count/(count-1))
 *       y = relu(gamma (z - mean) rstd + beta)
 *       dz_i = gamma rstd (dy_i [y_i > 0] - w_i (sum_j dy_j [y_j>0]) / count - w_i xhat_i (sum_j dy_j [y_j>0] xhat_j) / count)
 *   which is the exact gradient of the reference computation w.r.t. the de-duplicated rows (dy_i already holds
 *   the sum of the gradients of all copies of row i).
 *
 * groups: the reference runs the same module on frame 1 and then on frame 2 (models/track4d.py:88-92), i.e. two
 * BatchNorm calls with separate batch statistics and two running-stat updates; with groups = 2 both halves
 * of a stacked batch (samples/2 each) go through one launch with per-group statistics, and the running statistics
 * are updated group 0 first, then group 1 -- identical to the two sequential calls.
 *
 * Same conventions as rtk_pointnet2.h: caller-allocated device buffers, explicit stream, 0 / negative status.
 */
#ifndef RTK_TRAIN_H
#define RTK_TRAIN_H

#include "rtk_fused.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Batch-statistic buffers ("sums", "sums2" below) are (slots, groups, C, 2) float64 arrays, ZERO-INITIALISED by the caller and opaque
 * to it.  BIT 0 OF THE POINTER handed to a producer / consumer selects how the sums are kept (the same tag for every call that
 * touches one buffer):
 *   0: slots = RTK_STAT_SLOTS.  float64 atomics into eight replicas picked by the producer's sample index (one replica made every
 *      statistics epilogue a chain of 64..256 serialised atomics on the same address: 8..22 us per launch, more than the layers
 *      themselves at these sizes); the consumers add the replicas up.  The sum depends on the order of arrival in its last bits.
 *   1: slots = RTK_STAT_SLOTS_ORDERED.  ORDER-INDEPENDENT sums -- 90-bit fixed point in three 30-bit limbs kept as integers in
 *      float64 words, so that every atomic addition is exact (csrc/rtk_common.h rtk_stat_add), five replicas + a word of flags: with
 *      rtk_sa_first_layer_bwd's dwx_ws (the other operators have no order-dependent sums) a train step is reproducible bit for
 *      bit, at 2-3 % of its time. */
#define RTK_STAT_SLOTS 8
#define RTK_STAT_SLOTS_ORDERED 16

/* Weighted per-(group, channel) sums.  sums (RTK_STAT_SLOTS, groups, C, 2) float64, ZERO-INITIALISED by the caller:
 * [..,0] += sum w z, [..,1] += sum w z^2.  ns must be a power of two; row_weight may be NULL (all ones). */
RTK_EXPORT int rtk_bn_train_stats(int samples, int channels, int rows, int ns, int groups, const float *z,
                                  const float *row_weight, double *sums, rtk_stream_t stream);

/* y = relu(z scale + shift).  pool == 0: y has the shape of z.  pool != 0: y (samples, C, rows) = max over ns. */
RTK_EXPORT int rtk_bn_relu_fwd(int samples, int channels, int rows, int ns, int groups, const float *z, const float *par,
                               int pool, float *y, rtk_stream_t stream);

/* BatchNorm finalisation folded into the consumer: instead of reading par, the kernel derives (mean, rstd, scale, shift) of its
 * channels from the batch sums -- par (4, groups, C) fp32 = [mean | rstd | scale = gamma rstd | shift = beta - mean scale], count =
 * reference elements per channel and group -- and its first workgroups also store them into par_out (for the backward) and update
 * running_mean / running_var (momentum, unbiased variance) group by group, adding `groups` to num_batches_tracked.  One launch less per BatchNorm layer and step. */
typedef struct {
    const double *sums;             /* (RTK_STAT_SLOTS, groups, C, 2), complete */
    double count;
    const float *gamma, *beta;
    float eps, momentum;
    float *running_mean, *running_var;      /* optional */
    int64_t *num_batches_tracked;           /* optional */
    const double *group_counts;             /* optional DEVICE array (groups): per-group element counts that replace `count` -- padded
                                             * batches of clouds of different sizes (rtk_train_point_weights writes it) */
} rtk_bn_fin_t;
RTK_EXPORT int rtk_bn_relu_fwd_fin(int samples, int channels, int rows, int ns, int groups, const float *z, const rtk_bn_fin_t *fin,
                                   float *par_out, int pool, float *y, rtk_stream_t stream);
/* The pooled form that also records, per (sample, channel, row), the element the row's gradient will go to: zarg = its z value,
 * karg = its index inside the row (first arg-max of y; 255 when the whole row is clipped by the ReLU and receives no gradient).
 * With them the backward of the pooled layer needs no pass of its own over z:
 *   rtk_pool_bwd_stats_arg: sums2 (RTK_STAT_SLOTS, groups, C, 2) float64 += (sum d, sum d xhat(zarg)) over the rows, d = dy [karg != 255];
 *   the layer's dz = scale ((k == karg ? d : 0) - w (sums2[0] + xhat sums2[1]) / count) is formed ON LOAD by the consumers
 *   (rtk_conv_wgrad_stats / rtk_conv_bn_bwd_apply with a pooled source) from z itself. */
RTK_EXPORT int rtk_bn_relu_pool_fwd_fin_arg(int samples, int channels, int rows, int ns, int groups, const float *z, const rtk_bn_fin_t *fin,
                                            float *par_out, float *y, float *zarg_out, unsigned char *karg_out, rtk_stream_t stream);
RTK_EXPORT int rtk_pool_bwd_stats_arg(int samples, int channels, int rows, int groups, const float *dy, const float *zarg,
                                      const unsigned char *karg, const float *par, double *sums2, rtk_stream_t stream);

/* Backward, pass 1.  dy has the shape of y.  sums2 (groups, C, 2) float64 zero-initialised:
 * [..,0] += sum dy [y>0], [..,1] += sum dy [y>0] xhat   (pool: only the first arg-max position of each row counts). */
RTK_EXPORT int rtk_bn_relu_bwd_stats(int samples, int channels, int rows, int ns, int groups, const float *z,
                                     const float *dy, const float *par, int pool, double *sums2, rtk_stream_t stream);

/* Backward, pass 2: dz (shape of z); dgamma_dbeta (2, C) fp32 = [sum_g sums2[g][c][1] | sum_g sums2[g][c][0]].
 * group_counts: optional device array of per-group element counts replacing `count` (see rtk_bn_fin_t). */
RTK_EXPORT int rtk_bn_relu_bwd_apply(int samples, int channels, int rows, int ns, int groups, const float *z,
                                     const float *dy, const float *par, const float *row_weight, const double *sums2,
                                     double count, const double *group_counts, int pool, float *dz, float *dgamma_dbeta, rtk_stream_t stream);

/* Both backward passes in ONE launch for per-point layers (ns = 1; z, dy, dz (samples, C, positions), positions % 4 == 0, at most
 * 65536 elements per channel): one workgroup owns a channel, sums it, then applies -- no float64 sums buffer, no second launch.
 * dgamma_dbeta (2, C) as rtk_bn_relu_bwd_apply; row_weight (samples, positions) or NULL; count / group_counts as there. */
RTK_EXPORT int rtk_bn_relu_bwd_small(int samples, int channels, int positions, int groups, const float *z, const float *dy, const float *par,
                                     const float *row_weight, double count, const double *group_counts, float *dz, float *dgamma_dbeta,
                                     int split_groups, rtk_stream_t stream);
/* (split_groups != 0 with groups == 2: one workgroup per (channel, group) instead of per channel; dgamma_dbeta must then be
 * ZERO-INITIALISED -- the two groups' sums are added to it.) */

/* First layer of a set-abstraction SharedMLP from the per-point projection (conv([d_xyz || feats[idx]]) =
 * Wx.d_xyz + (Wf.feats)[idx]):  z[b][c][row][k] = proj[b][c][idx[b][row][k]] + wx[c] . dxyz[b][:, row, k], and the weighted
 * batch sums of z accumulated into sums (groups, C, 2) float64 (zero-initialised; as rtk_bn_train_stats).
 * proj (samples, C, n_src), idx (samples, rows, ns) int32 in [0, n_src), dxyz (samples, 3, rows, ns), wx: C rows of wx_pitch >= 3
 * floats (the first three columns of the layer's weight), z (samples, C, rows, ns); ns a power of two >= 4. */
RTK_EXPORT int rtk_sa_first_layer(int samples, int channels, int rows, int ns, int groups, int n_src, const float *proj, const int *idx,
                                  const float *dxyz, const float *wx, int wx_pitch, const float *row_weight, float *z, double *sums,
                                  rtk_stream_t stream);

/* Weight gradients that are contractions over positions, on the split matrix path (two fp16 pieces per operand, three products;
 * fp32 in, fp32 out, the error of an fp32 fmaf chain; csrc/train_gemm.hip):  out[i][j] = sum_{r < m} x[r][i] * y[r][j]  for up to
 * four (x, y, out) jobs in one launch.  The contraction runs over the rows, so an operand takes ONE power-of-two scale: x_amax /
 * y_amax point at a float >= max |element| of the operand (the cost-volume training kernels emit them; rtk_absmax computes one for
 * any tensor).  Elements far below the maximum keep an absolute precision of 2^-39 of it -- below one fp32 rounding of the sums they
 * enter.  x, y (m, 256) fp32 row-major, 16-byte aligned; out 256 rows of out_pitch >= 256 floats, fully written.  workspace: at least
 * njobs * 65536 floats; with njobs * 65536 * (256 / njobs) floats the grid is one slab per CU (the slabs' partial blocks are summed
 * by a second kernel: deterministic).  RANGE CONTRACT (one scale per tensor, round-5 advice): an element below 2^-25 of its tensor's
 * largest |element| loses relative precision (its low piece is an fp16 subnormal), one below 2^-39 contributes zero; a non-finite
 * element makes the scale, and with it the whole product, non-finite.  Callers whose rows span more than that (a gradient tensor with
 * one outlier position 10^8 above the rest) see the small rows' weight-gradient rows at that absolute floor -- below one fp32 rounding
 * of the sum the outlier dominates, but not below the RELATIVE precision a per-channel optimiser such as Adam normalises to
 * (tests/test_train_gpu.py::test_tn_gemm256_split_outlier_rows pins the floor).  Replaces the batched library GEMM of the cost volume's backward
 * (utils/model_utils/model_utils.py:177-183,226-231: the two 256 x 256 convolutions over N x 16 positions). */
typedef struct {
    const float *x, *y;
    float *out;
    int out_pitch;
    const float *x_amax, *y_amax;
} rtk_tn_job_t;
RTK_EXPORT int rtk_tn_gemm256_split(int njobs, const rtk_tn_job_t *jobs, long m, float *workspace, long workspace_floats, rtk_stream_t stream);
/* *amax = max(*amax, max |x[0 .. n)|): an unsigned maximum on the float bits (*amax zero-initialised, or a previous maximum). */
RTK_EXPORT int rtk_absmax(const float *x, long n, float *amax, rtk_stream_t stream);

/* Kernel images of live (trained) weights, all of an operator's matrices in ONE launch (the training path re-packs every step).
 * kind 0: fragment-major MFMA A-operand image of a (rows, cols) matrix, packed[u][v][16g+i][r] = W[16v+i][16u+4g+r], zero padded to
 *         multiples of 16 (fused.pack_layer); transpose != 0 packs W^T (element (o,k) = src[k*pitch + o]);
 * kind 1: offset-layer image of [W(:, 0:3) | b]: img[v][16g+i] = (g < 3 ? W[16v+i][g] : b[16v+i]) (b = src2, may be NULL = 0);
 * kind 2: vector copied and zero padded to a multiple of 16 (biases). */
typedef struct {
    const float *src;
    const float *src2;
    float *dst;
    int rows, cols, pitch, transpose, kind;
} rtk_pack_job_t;
RTK_EXPORT int rtk_pack_weights(int njobs, const rtk_pack_job_t *jobs, rtk_stream_t stream);

/* Backward of rtk_sa_first_layer.  rtk_group_inverse_index sorts the positions p = (row, k) of every sample by the source point
 * idx[s][p] they gather (once per geometry table and step): off (samples, n_src + 1) int32, inv (samples, positions) uint16 with
 * the positions referencing point q at inv[s][off[s][q] .. off[s][q+1]), ascending.  rtk_sa_first_layer_bwd reads dz
 * (samples, C, rows, ns) once: dproj (samples, C, n_src) = gather-sum of dz over each point's positions (fully written);
 * dwx (C rows of dwx_pitch >= 3 floats, e.g. the first three columns of the layer's full weight gradient), ZERO-INITIALISED by
 * the caller, += sum dz . dxyz (dxyz (samples, 3, rows, ns)).  ns % 4 == 0.  dproj is reproducible bit for bit (the segmented sums
 * are accumulated in per-plane fixed point, csrc/train_group.hip fx_plane).  dwx_ws (optional): workspace of samples * C * 3 floats --
 * every sample's share of dwx, added up in a fixed order by a second launch (reproducible, +5 us); NULL: the shares are added
 * to dwx with float atomics in the order of arrival. */
RTK_EXPORT int rtk_group_inverse_index(int samples, int n_src, int positions, const int *idx, int *off, unsigned short *inv,
                                       rtk_stream_t stream);
/* The same for several tables in ONE launch (up to 12; every table over the same `samples` clouds): a table is one workgroup per
 * cloud, so at small batches ten launches cost ten times the latency of one. */
typedef struct {
    int n_src, positions;
    const int *idx;
    int *off;
    unsigned short *inv;
    const int *live;      /* optional (samples) int32 on the device: only positions < live[s] * live_mult of sample s enter the table --  */
    int live_mult;        /* for rows nothing downstream reads (duplicate centroid rows of a de-duplicated level); NULL: all positions      */
} rtk_inverse_index_job_t;
RTK_EXPORT int rtk_group_inverse_index_multi(int samples, int njobs, const rtk_inverse_index_job_t *jobs, rtk_stream_t stream);

/* Backward of rtk_three_interpolate (lib/src/interpolate_gpu.cu:192-214, one atomicAdd per term) in gather form:
 * grad_points (b, c, m) = for every known point the sum of weight * grad_out over the (unknown point, slot) positions that
 * reference it, read off the inverse table (off (b, m+1), inv (b, 3n)) that rtk_group_inverse_index builds from the
 * interpolation indices idx (b, n, 3) taken as b lists of 3n positions.  Writes every element (no zero-fill needed), no
 * atomics, deterministic. */
/* n_valid (b) int32 or NULL: padded clouds -- the unknown points from n_valid[s] on are copies of point 0 (same neighbours and weights):
 * their gradient is folded into point 0's inside the kernel, and the table must have been built WITHOUT their positions
 * (rtk_inverse_index_job_t.live = n_valid, live_mult = 3). */
RTK_EXPORT int rtk_three_interpolate_grad_gather(int b, int c, int n, int m, const float *grad_out, const float *weight, const int *off,
                                                 const unsigned short *inv, float *grad_points, const int *n_valid, rtk_stream_t stream);
RTK_EXPORT int rtk_sa_first_layer_bwd(int samples, int channels, int rows, int ns, int n_src, const float *dz, const float *dxyz,
                                      const int *off, const unsigned short *inv, float *dproj, float *dwx, int dwx_pitch,
                                      float *dwx_ws, rtk_stream_t stream);

/* ---- 1x1 convolution fused with the BatchNorm work around it (set-abstraction SharedMLP layers 2, 3) ----------------
 * Tensors are NCHW planes (samples, C, rows*ns), ns a power of two >= 4, channel counts 16, 32 or 64.
 *
 * rtk_conv_bn_fwd:  a = relu(scale x + shift) with (scale, shift) of the PREVIOUS layer's BatchNorm read from pre_par
 * ((4, groups, cin) as written by the *_fin kernels; NULL: a = x), z = W a with w the plain row-major (cout, cin)
 * weight of the convolution, and this layer's weighted batch sums accumulated into sums (groups, cout, 2) float64,
 * zero-initialised by the caller (same meaning as rtk_bn_train_stats).  act_out (optional, shape of x) receives a. */
RTK_EXPORT int rtk_conv_bn_fwd(int samples, int cin, int cout, int rows, int ns, int groups, const float *x, const float *pre_par,
                               const float *w, float *z, float *act_out, const float *row_weight, double *sums,
                               rtk_stream_t stream);

/* rtk_conv_bn_fwd with the PREVIOUS layer's BatchNorm finalised on the fly (rtk_bn_fin_t above; pre_par_out (4, groups, cin)). */
RTK_EXPORT int rtk_conv_bn_fwd_fin(int samples, int cin, int cout, int rows, int ns, int groups, const float *x, const rtk_bn_fin_t *pre_fin,
                                   float *pre_par_out, const float *w, float *z, float *act_out, const float *row_weight, double *sums,
                                   rtk_stream_t stream);

/* rtk_conv_bn_bwd: backward through z = W relu(BatchNorm(zprev)) down to zprev.  dz (samples, cout, P) is the gradient of z,
 * w the SAME row-major (cout, cprev) weight as in the forward, pre_par the (4, groups, cprev) constants of the BatchNorm applied to
 * zprev.  apply == 0: sums2 (groups, cprev, 2) float64 (zero-initialised) += (sum dy m, sum dy m xhat) with dy = W^T dz and
 * m the ReLU mask.  apply != 0: dzprev = scale (dy m - w (sums2[0] + xhat sums2[1]) / count) is written, and
 * dgamma_dbeta (2, cprev) (optional) = (sum_g sums2[g][c][1] | sum_g sums2[g][c][0]). */
RTK_EXPORT int rtk_conv_bn_bwd(int samples, int cprev, int cout, int rows, int ns, int groups, const float *dz, const float *w,
                               const float *zprev, const float *pre_par, const float *row_weight, double *sums2, double count, int apply,
                               float *dzprev, float *dgamma_dbeta, rtk_stream_t stream);

/* Weight gradient of the same layer: dw (cout, cprev) row-major fp32 += sum over samples and positions of
 * dz (x) relu(BatchNorm(zprev)); the normalised activation is recomputed from zprev and pre_par on load.  One partial block per
 * workgroup goes through `workspace` (uninitialised, >= samples * cout * cprev floats; 1024 blocks = full parallelism) and a
 * second kernel adds them in a fixed order: no atomics, deterministic. */
RTK_EXPORT int rtk_conv_wgrad(int samples, int cprev, int cout, int rows, int ns, int groups, const float *dz, const float *zprev,
                              const float *pre_par, float *dw, float *workspace, long workspace_floats, rtk_stream_t stream);

/* One pass instead of rtk_conv_wgrad + the statistics pass of rtk_conv_bn_bwd (both read dz and zprev):  dw (cout, cprev) +=
 * sum dz a^T with a = relu(BatchNorm(zprev)), and sums2_out (RTK_STAT_SLOTS, groups, cprev, 2) float64, zero-initialised, +=
 * (sum dy m, sum dy m xhat) with dy = W^T dz -- what rtk_conv_bn_bwd_apply then takes as `sums2`.  Computed as two position
 * contractions G = dz m^T, H = dz (m xhat)^T per statistics group: dw = gamma H + beta G, sums2 = column sums of W o G, W o H.
 * gamma_prev / beta_prev: affine parameters of zprev's BatchNorm; w: the layer's forward weight (cout, cprev).
 * workspace: >= 2 * samples * cprev * cout floats of scratch (per-workgroup partials, added in a fixed order).
 *
 * Pooled source (`pool` != NULL, both functions): the first tensor is not dz but z of the pooled last layer of the chain, whose
 * gradient is formed on load from the layer's output gradient and the arg-max recorded by rtk_bn_relu_pool_fwd_fin_arg:
 *   dz = scale ((k == karg ? dout : 0) - w (c1 + xhat c2)),  (c1, c2) = pool.sums2 / count  (rtk_pool_bwd_stats_arg)
 * -- the separate apply pass over z (read z, write dz) disappears.  pool.dgamma_dbeta (2, cout), optional, receives the pooled
 * BatchNorm's parameter gradients from the apply kernel. */
typedef struct {
    const float *dout;              /* (samples, cout, rows) gradient of the pooled output */
    const unsigned char *karg;      /* (samples, cout, rows) */
    const float *par;               /* (4, groups, cout) of the pooled layer's BatchNorm */
    const double *sums2;            /* (RTK_STAT_SLOTS, groups, cout, 2) */
    float *dgamma_dbeta;            /* (2, cout) or NULL */
} rtk_pool_src_t;
RTK_EXPORT int rtk_conv_wgrad_stats(int samples, int cprev, int cout, int rows, int ns, int groups, const float *dz_or_z,
                                    const rtk_pool_src_t *pool, const float *zprev, const float *pre_par, const float *row_weight, double count,
                                    const float *w, const float *gamma_prev, const float *beta_prev, float *dw, double *sums2_out,
                                    float *workspace, long workspace_floats, rtk_stream_t stream);
RTK_EXPORT int rtk_conv_bn_bwd_apply(int samples, int cprev, int cout, int rows, int ns, int groups, const float *dz_or_z,
                                     const rtk_pool_src_t *pool, const float *w, const float *zprev, const float *pre_par,
                                     const float *row_weight, const double *sums2, double count, float *dzprev, float *dgamma_dbeta,
                                     rtk_stream_t stream);

/* ---- per-point layers: 1x1 convolutions on one row per point / centroid (feature propagation, nn.Linear bottlenecks, layer-1
 * feature projections, predictor heads: lib/pointnet2_modules.py:140-158, utils/model_utils/model_utils.py:308-357,393-424) -----
 * An operand of a VIRTUAL channel concatenation: element (sample s, channel c, position p) lives at
 *   layout 0 (channel-major planes):  ptr[s*sample_stride + c*pitch + p]
 *   layout 1 (point-major rows):      ptr[s*sample_stride + p*pitch + c]
 * col0 = the operand's first column (input side) or row (output side) in the weight matrix. */
typedef struct {
    const float *ptr;
    long sample_stride;
    int pitch;
    int channels;
    int layout;
    int col0;
} rtk_pw_operand_t;

/* [dst_0 ; dst_1 ; ...] = W . [src_0 ; src_1 ; ...] (+ bias): weight element (o, k) = w[o*w_pitch + k], or w[k*w_pitch + o] with
 * transpose_w (the input gradient W^T dz of the same layer, written into each source's gradient tensor; accumulate != 0 adds
 * to the destinations).  bias (optional) and sums (optional; (groups, stat_channels, 2) float64, zero-initialised:
 * += (sum w z, sum w z^2) over the positions, w = row_weight (samples, positions) or 1) are indexed by the weight row of the
 * output channel.  Up to 4 sources and 4 destinations; any channel and position counts. */
RTK_EXPORT int rtk_pw_conv(int samples, int positions, int nsrc, const rtk_pw_operand_t *srcs, int ndst, const rtk_pw_operand_t *dsts,
                           const float *w, int w_pitch, int transpose_w, const float *bias, int accumulate, const float *row_weight,
                           int groups, double *sums, int stat_channels, rtk_stream_t stream);

/* dw[o][src_i.col0 + k] += sum over samples and positions of dz[o] * src_i[k]; dbias[o] += sum dz[o] (optional).  dw (and dbias)
 * hold the values to add to (zeros for a plain gradient).  The position axis is split over workgroups whose 64 x 64 partial
 * blocks go through `workspace` (workspace_floats floats, uninitialised, >= 4096 per split and block; 4 Mi floats serve every
 * shape at full parallelism; NULL: no split) and are added by a second kernel in a fixed order: no atomics, deterministic. */
RTK_EXPORT int rtk_pw_wgrad(int samples, int positions, const rtk_pw_operand_t *dz, int nsrc, const rtk_pw_operand_t *srcs, float *dw,
                            int w_pitch, float *dbias, float *workspace, long workspace_floats, rtk_stream_t stream);
/* Any number of such weight gradients with as few launches as possible (eight jobs per launch: their parameters travel in the kernel
 * argument).  The weight gradients of a training step are leaves -- nothing but the optimizer reads them -- so the training path
 * queues them during the backward and issues them together at its end.  The workspace is shared out equally among a launch's jobs. */
typedef struct {
    int samples, positions;
    const rtk_pw_operand_t *dz;
    int nsrc;
    const rtk_pw_operand_t *srcs;
    float *dw;
    int w_pitch;
    float *dbias;
} rtk_pw_wgrad_job_t;
RTK_EXPORT int rtk_pw_wgrad_multi(int njobs, const rtk_pw_wgrad_job_t *jobs, float *workspace, long workspace_floats, rtk_stream_t stream);


/* ---- cost volume (utils/model_utils/model_utils.py:216-236) ------------------------------------------------------
 * Training forward: rtk_cost_volume (rtk_fused.h; same arguments and result) that also keeps the three activations
 * a1, a2, a3 (M, 256), M = samples*n1*16 positions (query-major, neighbour minor), for the backward (a1, a2: operands of its
 * weight-gradient GEMMs; a3: read by rtk_cost_volume_bwd), and mask1, mask2 (M, 32 bytes each): the sign bits of a1, a2 in
 * the kernel's own lane order (opaque; 1/32 of the bytes the backward would otherwise re-read for the leaky-ReLU slopes). */
RTK_EXPORT int rtk_cost_volume_train(int samples, int n1, int n2, const float *xyz1, const float *xyz2, const int64_t *knn_idx,
                                     const float *p1, const float *p2, const float *wd_packed, const rtk_layer_t *layers,
                                     const rtk_layer_t *wn, float *out, int out_pitch, float *a1, float *a2, float *a3,
                                     void *mask1, void *mask2, rtk_stream_t stream);

/* Backward of rtk_cost_volume_train.  layers_t[0..1] = the packed transposed 256x256 layers W3^T, W2^T, contiguous in
 * memory; wn = the WeightNet images of the forward; wct_packed = the packed transpose of the WeightNet's last layer
 * (8 x 256 -> [16][1] fragments).  dout (samples*n1, dout_pitch) is the gradient of the forward output; a3, mask1, mask2 what
 * the forward saved.  Outputs, all caller-allocated:
 *   dz1, dz2, dz3 (M,256)  gradients of the three pre-activations (dW2 = dz2^T a1, dW3 = dz3^T a2: GEMMs on the host side)
 *   dq3 (M,256)      gradient of the WeightNet's last pre-activation  (dWc|dbc = dq3^T [t2 | 1], rtk_weightnet_bwd)
 *   dt2 (M,8)        Wc^T dq3: gradient of the WeightNet's second hidden activation (before its ReLU mask)
 *   d4 (M,4)         (neighbour - query, 1) of every position
 *   dp1 (samples*n1, 256)     gradient of p1 = sum of dz1 over the 16 neighbours
 *   dpd (samples*n1, 3, 256)  per-query partial sums of dz1 x direction; dWd[c][k] = sum over queries of dpd[q][k][c].
 *   dbias_rows (samples*n1, 2, 256), optional (NULL: not computed)  per-query sums over the 16 neighbours of dz3 | dz2:
 *                    db3 | db2 = their column sums (a 16x smaller reduction than the column sums of dz3, dz2 themselves).
 * The gradient of p2 is rtk_scatter_add_rows(knn_idx, dz1). */
RTK_EXPORT int rtk_cost_volume_bwd(int samples, int n1, int n2, const float *xyz1, const float *xyz2, const int64_t *knn_idx,
                                   const rtk_layer_t *layers_t, const rtk_layer_t *wn, const float *wct_packed, const float *dout,
                                   int dout_pitch, const float *a3, const void *mask1, const void *mask2, float *dz1, float *dz2, float *dz3,
                                   float *dq3, float *d4, float *dp1, float *dpd, float *dt2, float *dbias_rows, rtk_stream_t stream);

/* The same two operators with their 256 x 256 products on the split matrix path (rtk_fused.h, csrc/split_mfma.h): identical
 * arguments and tensor formats (activations, sign masks, gradients), except that the two layers arrive as split images with their
 * inverse scales -- forward: W2, W3 (rtk_pack_split_layer) + fp32 biases; backward: W3^T, W2^T (rtk_pack_split_layer with
 * transposed = 1 on the forward's weights) -- and that the backward builds its Wc^T operand itself (no wct_packed).  act_amax / dz_amax (optional, [2] floats,
 * ZERO-INITIALISED by the caller): the largest |element| of (a1, a2) / (dz3, dz2), folded in by the kernels for rtk_tn_gemm256_split. */
RTK_EXPORT int rtk_cost_volume_split_train(int samples, int n1, int n2, const float *xyz1, const float *xyz2,
                                           const int64_t *knn_idx, const float *p1, const float *p2, const float *wd_packed,
                                           const void *split_images, const float *image_scales, const float *bias2,
                                           const float *bias3, const rtk_layer_t *wn, float *out, int out_pitch, float *a1, float *a2,
                                           float *a3, void *mask1, void *mask2, float *act_amax, rtk_stream_t stream);
RTK_EXPORT int rtk_cost_volume_bwd_split(int samples, int n1, int n2, const float *xyz1, const float *xyz2,
                                         const int64_t *knn_idx, const void *split_images_t, const float *image_scales_t,
                                         const rtk_layer_t *wn, const float *dout, int dout_pitch, const float *a3, const void *mask1, const void *mask2,
                                         float *dz1, float *dz2, float *dz3, float *dq3, float *d4, float *dp1, float *dpd,
                                         float *dt2, float *dbias_rows, float *dz_amax, rtk_stream_t stream);

/* Backward of rtk_patch_cost (rtk_fused.h; same forward arguments; feat point-major).  dout (samples*n, dout_pitch).
 * Outputs over the M = samples*n*16 positions: dxg (M,256) = dout * wn (optional; scatter it onto feat's rows with
 * rtk_scatter_add_rows(knn_idx, dxg), or pass NULL and use rtk_patch_dfeat_gather), dq3 (M,256), dt2 (M,8), d4 (M,4) as in
 * rtk_cost_volume_bwd, t2 (M,8) (optional). */
RTK_EXPORT int rtk_patch_cost_bwd(int samples, int n, const float *xyz, const int64_t *knn_idx, const float *feat, int feat_pitch,
                                  const rtk_layer_t *wn, const float *wct_packed, const float *dout, int dout_pitch, float *dxg,
                                  float *dq3, float *dt2, float *d4, float *t2, rtk_stream_t stream);

/* The feature gradient of rtk_patch_cost without dxg: rtk_patch_cost_bwd with dxg = NULL and t2 (M, 8) (the WeightNet's hidden
 * activation per position), then  dfeat (samples*n, 256)[m] = sum over the positions (i,k) with knn[i,k] = m of
 * relu(wc t2 + bc) * dout[i]  over the inverse table (off (samples, n+1), inv (samples, 16n)) that rtk_group_inverse_index
 * builds from the kNN table (as int32, n_src = n, positions = 16 n).  wc (256, 8), bc (256): the live last-layer parameters.
 * Fully written, no atomics, deterministic. */
/* n_valid (samples) int32 + row0 (samples, 256) workspace, or both NULL: padded clouds -- the query points from n_valid[s] on are
 * copies of point 0 (same neighbours, same WeightNet output): their dout rows are added to row 0's (into row0, one more small launch)
 * and the table must have been built WITHOUT their positions (live = n_valid, live_mult = 16). */
RTK_EXPORT int rtk_patch_dfeat_gather(int samples, int n, const int *off, const unsigned short *inv, const float *t2, const float *wc,
                                      const float *bc, const float *dout, int dout_pitch, float *dfeat, const int *n_valid, float *row0,
                                      rtk_stream_t stream);

/* ---- de-duplicated geometry tables (ratrack_amd/train_path.py) ---------------------------------------------------------
 * rtk_train_group_geometry: for the first `rows` centroids of every sample, idx_out (samples, rows, ns) = ball_idx
 * (samples, npoint, ns) with entries >= src_live_rows -- source rows the de-duplicated level tensor does not hold: copies of row 0
 * -- redirected to 0 (rows in [nuniq, src_live_rows) exist as computed copies of row 0 and are used as they are); centroid rows
 * >= dst_nuniq[b] (copies of centroid 0 whose ball lists the query skipped; dst_nuniq may be NULL) take centroid 0's list; and
 * dxyz (samples, 3, rows, ns) = src_xyz[idx_out] - dst_xyz (neighbour - centroid, lib/pointnet2_utils.py:279-285);
 * src_xyz (samples, n_src_rows, 3), dst_xyz (samples, npoint, 3). */
RTK_EXPORT int rtk_train_group_geometry(int samples, int n_src_rows, int npoint, int rows, int ns, const float *src_xyz,
                                        const float *dst_xyz, const int *ball_idx, int src_live_rows, const int *dst_nuniq, int *idx_out,
                                        float *dxyz, rtk_stream_t stream);
/* three-NN tables -> interpolation weights (lib/pointnet2_modules.py:143-146: 1/(sqrt(d2)+1e-8), normalised) and indices with
 * entries >= known_nuniq[b] redirected to 0, for the first `rows` of `rows_total` unknown rows. */
RTK_EXPORT int rtk_train_interp_weights(int samples, int rows_total, int rows, const float *dist2, const int *idx, const int *known_nuniq,
                                        int *idx_out, float *weight_out, rtk_stream_t stream);
/* BatchNorm row weights of a level: w[b][r] = [r < nuniq[b]] + [r == 0] (npoint - nuniq[b]). */
RTK_EXPORT int rtk_train_row_weights(int samples, int rows, int npoint, const int *nuniq, float *weights, rtk_stream_t stream);
/* Level-0 statistics weights of a padded batch (clouds of n_valid[s] <= rows points, the rest copies of their point 0):
 * weights (samples, rows) = [r < n_valid[s]]; group_counts (groups) float64 = sum of n_valid over each group's samples -- the
 * per-group element counts of the per-point BatchNorm layers (rtk_bn_fin_t.group_counts). */
RTK_EXPORT int rtk_train_point_weights(int samples, int rows, int groups, const int *n_valid, float *weights, double *group_counts,
                                       rtk_stream_t stream);

/* The stacked weight images rtk_gru_step / rtk_gru_step_bwd take, from the nn.GRU's live per-layer parameters, in one launch:
 * params = host array of 4 * layers device pointers (w_ih_l, w_hh_l (3H,H), b_ih_l, b_hh_l (3H) for l = 0 ..); w_ih / w_hh (L,3H,H),
 * w_ih_t / w_hh_t (L,H,3H), b_ih / b_hh (L,3H). */
RTK_EXPORT int rtk_gru_pack_params(int layers, int hidden, const float *const *params, float *w_ih, float *w_ih_t, float *w_hh, float *w_hh_t,
                                   float *b_ih, float *b_hh, rtk_stream_t stream);

/* ---- GRU step (fd_layer.torchGRU on a length-1 sequence, utils/model_utils/model_utils.py:279,296) -------------------
 * Backward of rtk_gru_step (rtk_fused.h).  x (B,H), h_in / h_out (L,B,H) as in the forward; w_ih_t, w_hh_t the TRANSPOSED
 * weights (L,H,3H) of the forward, w_ih, w_hh the original (L,3H,H), b_ih, b_hh (L,3H); dy (B,H) gradient of y = h_out[L-1],
 * dh_out (L,B,H) gradient of h_out (may be NULL).  Outputs: dx (B,H), dh_in (L,B,H), and the gate gradients dgi, dgh
 * (L,B,3H) from which the caller forms dW_ih[l] = dgi[l]^T x_l, dW_hh[l] = dgh[l]^T h_in[l], db = column sums
 * (x_0 = x, x_l = h_out[l-1]).  H = 128. */
RTK_EXPORT int rtk_gru_step_bwd(int b, int layers, int hidden, const float *x, const float *h_in, const float *h_out,
                                const float *w_ih_t, const float *w_hh_t, const float *w_ih, const float *w_hh, const float *b_ih,
                                const float *b_hh, const float *dy, const float *dh_out, float *dx, float *dh_in, float *dgi, float *dgh,
                                rtk_stream_t stream);

/* The GRU's parameter gradients from those gate gradients, one launch: dw_ih, dw_hh (L,3H,H), db_ih, db_hh (L,3H), fully written. */
RTK_EXPORT int rtk_gru_wgrad(int b, int layers, int hidden, const float *x, const float *h_in, const float *h_out, const float *dgi,
                             const float *dgh, float *dw_ih, float *dw_hh, float *db_ih, float *db_hh, rtk_stream_t stream);

/* dst[b][idx[b][r]][:] += src[b][r][:] for r < m, dst (samples, n, channels) fully written (no zero-fill needed):
 * the scatter half of the backward of a row gather.  idx (samples, m) int64 in [0, n); channels % 32 == 0.
 * channels == 256: partitioned by destination rows, deterministic, any n.  Otherwise: 32-channel slabs accumulated with
 * LDS atomics, n * 128 bytes of LDS per workgroup (n <= 1024). */
RTK_EXPORT int rtk_scatter_add_rows(int samples, int m, int n, int channels, const int64_t *idx, const float *src, float *dst,
                                    rtk_stream_t stream);

/* Parameter gradients of a WeightNet(3 -> 8 -> 8 -> C) (utils/model_utils/model_utils.py:359-390) from what rtk_cost_volume_bwd /
 * rtk_patch_cost_bwd emit per position: d4 (M,4), dq3 (M,C), dt2 (M,8).  wa (8,3) ba (8) wb (8,8) bb (8) row-major live weights;
 * dwa (8,3) dba (8) dwb (8,8) dbb (8) dwc (C,8) dbc (C) are added to (zeros for a plain gradient).  Workgroup partial vectors of
 * (9C + 104) floats go through `workspace` (uninitialised, >= one vector; 1024 vectors = full parallelism) and are added up in a
 * fixed order by a second kernel: no atomics, deterministic. */
RTK_EXPORT int rtk_weightnet_bwd(long positions, int channels, const float *d4, const float *dq3, const float *dt2, const float *wa,
                                 const float *ba, const float *wb, const float *bb, float *dwa, float *dba, float *dwb, float *dbb,
                                 float *dwc, float *dbc, float *workspace, long workspace_floats, rtk_stream_t stream);

/* "Append the cloud's global feature to every point" (models/track4d.py:92-95: torch.max over the points, expand, cat) as one kernel:
 * f (samples, C, n) -> out (samples, 2 C, n) = [f ; max_p f broadcast]; arg (samples, C) int32 = first arg-max of every row, for the
 * backward: df[s][c][p] = dout[s][c][p] + (p == arg[s][c] ? sum_p' dout[s][C + c][p'] : 0)  (what autograd computes for
 * cat + expand + max, in one pass).  All tensors contiguous. */
RTK_EXPORT int rtk_gmax_cat_fwd(int samples, int channels, int n, const float *f, float *out, int *arg, rtk_stream_t stream);
RTK_EXPORT int rtk_gmax_cat_bwd(int samples, int channels, int n, const float *dout, const int *arg, float *df, rtk_stream_t stream);

/* Multi-task loss of the backbone trainer (losses/loss.py:8-31,85-89,124-146, batch mean) and its gradients in one launch.
 * pc1, flow, gt_warp (B,3,N) contiguous; cls (B,N) probabilities; gt_cls uint8/bool, sample b's row at gt_cls + b*gt_cls_stride
 * (stride 0: one label vector for the whole batch).  items (5 + 2 B) fp32 ZERO-INITIALISED: [0..3] = [Loss, SceneFlowLoss,
 * TrackingLoss (left 0), SegLoss], [4] an arrival counter, [5..] the samples' shares -- summed sample 0 first by the last workgroup
 * to arrive (reproducible bit for bit).  dflow (B,3,N) / dcls (B,N) (optional): d Loss / d flow (not written while pre-training: the loss does not depend
 * on the flow then) and d Loss / d cls.  n_valid (B) int32, optional: padded batch -- sample b consists of its first n_valid[b]
 * points; the padding columns enter no mean and get zero gradients. */
RTK_EXPORT int rtk_backbone_loss(int b, int n, const float *pc1, const float *flow, const float *gt_warp, const float *cls,
                                 const unsigned char *gt_cls, int gt_cls_stride, int pretrain, float *items, float *dflow, float *dcls,
                                 const int *n_valid, rtk_stream_t stream);

/* Adam (torch.optim.Adam semantics: L2 weight decay in the gradient, bias-corrected, no amsgrad; main.py:61) over n_tensors fp32
 * tensors in one launch.  table: n_tensors rows of seven 64-bit words {param, grad, exp_avg, exp_avg_sq, numel, first workgroup,
 * step}, 4096 elements per workgroup, total_blocks = sum over tensors of ceil(numel / 4096); step: the parameter's own device
 * scalar (float), the number of steps it has taken -- the kernel uses step + 1 and advances it.  lr_ptr (device scalar)
 * overrides lr when not NULL.  ticket: device int, zero (the kernel leaves it zero). */
RTK_EXPORT int rtk_adam_multi(int n_tensors, const void *table, long total_blocks, const float *lr_ptr, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int *ticket, rtk_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RTK_TRAIN_H */
