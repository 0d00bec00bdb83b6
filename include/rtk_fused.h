/*
 * rtk_fused.h -- C ABI of the fused inference stages of librtk_hip.so.
 *
 * These entry points have no single counterpart in the reference: each one replaces a SEQUENCE of
 * framework ops + pointnet2_cuda calls that the reference issues from Python, fused
 * into one gfx950 kernel working on point-major (row = point, contiguous channels) fp32 tensors:
 *
 *   rtk_pointwise_mlp   [three_nn weights + three_interpolate + cat skip] -> [Conv 1x1 + BN + act] x L
 *                       (lib/pointnet2_modules.py:140-158, lib/pytorch_utils.py:20-32,
 *                        utils/model_utils/model_utils.py:308-357, nn.Linear bottlenecks :414-418)
 *   rtk_sa_scale        group (xyz - centroid || features) -> SharedMLP -> max over the ball
 *                       (lib/pointnet2_utils.py:269-292 + lib/pointnet2_modules.py:37-53)
 *   rtk_cost_volume     kNN gather -> 3-layer MLP -> WeightNet -> weighted sum over neighbours
 *                       (utils/model_utils/model_utils.py:216-236)
 *   rtk_patch_cost      kNN gather -> WeightNet -> weighted sum          (model_utils.py:238-248)
 *
 * BatchNorm (eval mode) is folded into the packed weights/bias on the host; weights use the
 * fragment-major packing documented in ratrack_amd/csrc/fused_common.h.  Same conventions as
 * rtk_pointnet2.h: caller-allocated device buffers, explicit stream, 0 / negative status.
 */
#ifndef RTK_FUSED_H
#define RTK_FUSED_H

#include "rtk_pointnet2.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RTK_MAX_SRC 4
#define RTK_MAX_LAYERS 4

/* One input segment of a concatenated per-point feature vector. */
typedef struct {
    const float *ptr;  /* (rows, pitch) point-major, or (samples, pitch) when per_sample != 0 */
    int pitch;         /* floats per row, multiple of 4, buffer readable up to ceil4(channels) */
    int channels;      /* valid channels; the segment occupies ceil16(channels) input slots */
    int per_sample;    /* 1: row index = sample (broadcast over the sample's points) */
} rtk_src_t;

/* One 1x1-conv layer with folded BN: y = act(W x + b).  w_packed is [cin16][cout16][64][4] floats -- or, with RTK_LAYER_SPLIT
 * or-ed into act (rtk_pointwise_mlp only; all layers of a chain alike), the layer's 16-position split image: one 1 KiB fragment of
 * 64 lanes x 8 fp16 per (pair of 16-channel input blocks, 16-channel output block, piece h / l) of 2^k W, see csrc/fused_common.h,
 * with inv_scale = 2^-k (k: max|W| 2^k in [2^14, 2^15)). */
#define RTK_LAYER_SPLIT 0x100
typedef struct {
    const float *w_packed;
    const float *bias; /* 16*cout16 floats (zero padded) */
    int cin16, cout16; /* channel counts in units of 16 */
    int act;           /* RTK_ACT_* (0 none, 1 relu, 2 leaky 0.1, 3 sigmoid) */
    float inv_scale;   /* split images only: the inverse of the image's power-of-two weight scale */
} rtk_layer_t;

/* Optional first segment produced by three-NN inverse-distance interpolation. */
typedef struct {
    const float *known_feats; /* (samples * m, pitch) point-major */
    int pitch, channels, m;
    const int *idx;       /* (rows, 3) int32, indices into the sample's m known points */
    const float *dist2;   /* (rows, 3) squared distances from rtk_three_nn */
    const int *nuniq;     /* optional (samples): known rows >= nuniq[b] are duplicates of row 0 and read as row 0 */
} rtk_interp_t;

/* rows = total points (samples * rows_per_sample).  Input vector = [interp segment (optional)] ||
 * srcs[0] || srcs[1] ...; each segment padded to a multiple of 16 channels.  sample_bias (optional,
 * (samples, 16*layers[0].cout16)) is added to layer 0's pre-activation.  Output: point-major
 * (rows, out_pitch) at channel offset 0, or channel-major (samples, out_channels, rows_per_sample)
 * when out_channel_major != 0; only out_channels channels are written.
 * row_nuniq (optional, (samples)): rows r >= row_nuniq[b] of sample b are duplicates of the sample's row 0
 * (centroids picked after furthest-point sampling exhausted the cloud, see rtk_fps_centroids); they are
 * neither read nor written -- consumers alias them to row 0.
 * colmax (optional, (samples, 16*last.cout16), ZERO-INITIALISED by the caller): per-sample maximum over the rows of
 * every output channel (torch.max(features, -1), models/track4d.py:89-92), accumulated with atomic max on the float
 * bits -- only valid when the last activation is non-negative (ReLU / sigmoid). */
RTK_EXPORT int rtk_pointwise_mlp(int rows, int rows_per_sample, const rtk_interp_t *interp, int nsrc,
                                 const rtk_src_t *srcs, const float *sample_bias, int nlayers,
                                 const rtk_layer_t *layers, float *out, int out_pitch, int out_channels,
                                 int out_channel_major, const int *row_nuniq, float *colmax, rtk_stream_t stream);

/* One scale of a set-abstraction level.  q (samples*n, q_pitch): per-point layer-1 projection of the
 * features (BN scale folded); layer 1 = relu(q[idx] + Wx.(xyz[idx] - centroid) + b1) with
 * w1xyz_packed the [1][c1_16][64][4] image of [Wx | b1]; then nlayers more packed layers; the last
 * one's bias+ReLU is applied after the max over the nsample neighbours.  idx from rtk_ball_query.
 * out (samples*npoint, out_pitch) receives 16*last.cout16 channels at out_offset.
 * src_nuniq / dst_nuniq (optional, (samples)): duplicate-row counters of the source level (q rows >= src_nuniq[b]
 * are read as row 0) and of the centroid level (centroids >= dst_nuniq[b] are skipped). */
RTK_EXPORT int rtk_sa_scale(int samples, int n, int npoint, int nsample, const float *xyz,
                            const float *new_xyz, const int *idx, const float *q, int q_pitch, int c1_16,
                            const float *w1xyz_packed, int nlayers, const rtk_layer_t *layers, float *out,
                            int out_pitch, int out_offset, const int *src_nuniq, const int *dst_nuniq,
                            rtk_stream_t stream);

/* Point-to-patch cost volume for k = 16 neighbours.  p1 (samples*n1, 256) / p2 (samples*n2, 256):
 * first-layer projections of the query / neighbour features (bias folded into p1);
 * layer 1 = leaky(p1[i] + p2[idx] + Wd.(xyz2[idx] - xyz1[i])), then layers[0..1] (256->256, leaky),
 * WeightNet wn[0..2] (3->8->8->256, ReLU) on the same direction vectors, out = sum_k wn * feat. */
RTK_EXPORT int rtk_cost_volume(int samples, int n1, int n2, const float *xyz1, const float *xyz2,
                               const int64_t *knn_idx, const float *p1, const float *p2,
                               const float *wd_packed, const rtk_layer_t *layers,
                               const rtk_layer_t *wn, float *out, int out_pitch, rtk_stream_t stream);

/* Patch-to-patch aggregation: out[i] = sum_k WeightNet(xyz[idx[i,k]] - xyz[i]) * feat[idx[i,k]]. */
RTK_EXPORT int rtk_patch_cost(int samples, int n, const float *xyz, const int64_t *knn_idx,
                              const float *feat, int feat_pitch, const rtk_layer_t *wn, float *out,
                              int out_pitch, int out_channel_major, rtk_stream_t stream);

/* ---- split matrix path (csrc/split_mfma.h): fp32 results from the fp16 matrix pipe --------------------------------------
 * Every fp32 operand, scaled by an exact power of two, is the sum of two fp16 pieces up to 2^-23 of itself; a product is the
 * three partial products l.h + h.l + h.h accumulated in fp32.  The error is that of an fp32 fmaf chain (2.7e-7 of max|y| on a
 * 256-deep product against float64; an fp32 GEMM: 6.4e-7); the matrix time is 3/16 of the fp32-input MFMA's, the only exact-fp32
 * matrix instruction of gfx950.  Scales: one power of two per weight matrix (chosen by the packer), one per position (chosen by
 * the kernels from the position's own largest activation) -- the path has no operand range to leave, inputs of any fp32
 * magnitude give fp32-accurate results (non-finite inputs give non-finite outputs, as in the reference).
 *
 * rtk_pack_split_layer: w (cout, cin) row-major fp32 (transposed != 0: the layer is w^T, w stored (cin, cout) row-major -- the
 * backward's W^T products from the forward's weights), both multiples of 32, cout * cin <= 2^22 -> image of cin/16 * cout/32 * 2
 * fragments of 1 KiB (4 * cout * cin bytes), fragment (s, v, p) = piece p (0: h, 1: l) of 2^k W, rows 32 v .. +31 against the 16
 * input channels of k-step s in the lane order of v_mfma_f32_32x32x16_f16; *inv_scale = 2^-k (k: max|W| 2^k in [2^14, 2^15)).
 * Consumers take the image and a pointer to its inverse scale (layers back to back: images back to back, scales back to back).
 * rtk_split_mlp2: y = leaky(W2 leaky(W1 x + b1) + b2), LeakyReLU(0.1), x / y (positions, 256) point-major; images = the split
 * images of W1 and W2 back to back, image_scales = their two inverse scales.  The inner layers of the cost volume
 * (utils/model_utils/model_utils.py:216-236) as a standalone operator: what tests and tools time the matrix path with. */
RTK_EXPORT int rtk_pack_split_layer(int cout, int cin, const float *w, int transposed, void *image, float *inv_scale,
                                    rtk_stream_t stream);
/* rtk_cost_volume with its two 256 x 256 layers on the split path: same arguments, the layers as their split images (W2, W3
 * back to back, 2 * 262144 bytes, with their two inverse scales) and fp32 biases instead of the packed rtk_layer_t pair.  samples * n2 <= 2^22 (the gathered
 * p2 rows are requested with 32-bit byte offsets; RTK_ERR_INVALID beyond: split the batch). */
RTK_EXPORT int rtk_cost_volume_split(int samples, int n1, int n2, const float *xyz1, const float *xyz2,
                                     const int64_t *knn_idx, const float *p1, const float *p2, const float *wd_packed,
                                     const void *split_images, const float *image_scales, const float *bias2,
                                     const float *bias3, const rtk_layer_t *wn, float *out, int out_pitch, rtk_stream_t stream);
/* ... on a share of the chip.  The kernel keeps a CU whole (486 registers per lane, 112 KiB of LDS): launched with one workgroup per
 * CU it stops every other kernel for its duration.  With several batches in flight (ratrack_amd.fused.GraphPipeline) the step is
 * shorter when it takes `workgroups` < the CU count -- 3/4 of them at B = 64: the kernel itself runs 16 % longer, the pipelined
 * forward 2 % faster, the other batches' kernels keep a quarter of every XCD.  workgroups = 0: all CUs (rtk_cost_volume_split);
 * rounded down to a multiple of 8 (one share per XCD: all tiles of sample s run on XCD s % 8); ignored unless samples % 8 == 0. */
RTK_EXPORT int rtk_cost_volume_split_shared(int samples, int n1, int n2, const float *xyz1, const float *xyz2,
                                            const int64_t *knn_idx, const float *p1, const float *p2, const float *wd_packed,
                                            const void *split_images, const float *image_scales, const float *bias2,
                                            const float *bias3, const rtk_layer_t *wn, float *out, int out_pitch, int workgroups,
                                            rtk_stream_t stream);
/* rtk_sa_scale for the scales whose MLP is offset layer (c1 = 32 or 64 channels) + ONE layer c1 -> 64, nsample 16 or 32 (sa2 scale 1,
 * sa3 scales 0 and 1 of the PNHead): same arguments, the layer as its split image + inverse scale (rtk_pack_split_layer(64, c1, ...)) + fp32 bias. */
RTK_EXPORT int rtk_sa_scale_split(int samples, int n, int npoint, int nsample, const float *xyz, const float *new_xyz,
                                  const int *idx, const float *q, int q_pitch, int c1, const float *w1xyz_packed,
                                  const void *split_image, const float *image_scale, const float *bias2, float *out, int out_pitch,
                                  int out_offset, const int *src_nuniq, const int *dst_nuniq, rtk_stream_t stream);
RTK_EXPORT int rtk_split_mlp2(int positions, const float *x, const void *images, const float *image_scales, const float *bias1,
                              const float *bias2, float *y, rtk_stream_t stream);

/* Layout glue of Track4D.backbone (models/track4d.py:104-105): (B,3,N)/(B,2,N) channel-major inputs of both
 * frames -> xyz (2B,N,3) and raw (2B,N,4) = (RCS, v_r, 0, 0) point-major, frame 1 first. */
RTK_EXPORT int rtk_prepare_inputs(int b, int n, const float *pc1, const float *pc2, const float *feature1,
                                  const float *feature2, float *xyz, float *raw, rtk_stream_t stream);

/* furthest_point_sample + gather_operation of a set-abstraction level in one launch
 * (lib/pointnet2_modules.py:30-35): same selection rule as rtk_furthest_point_sampling with the
 * min-distance scratch held on chip (initialised to 1e10).  idx (B,npoint) int32, new_xyz (B,npoint,3);
 * nuniq (B) int32 (optional) = number of picks made before the cloud was exhausted (every later pick is
 * point 0, i.e. a duplicate of centroid 0).  tie (B) int32 (optional) = the LAST round in which more than one point attained a
 * non-zero maximum (the pick then depended on the reference's tie rule), 0 if there was none.
 * n_valid (B) int32 (optional): padded batch -- sample b's cloud is its first n_valid[b] <= n points (row pitch n); the
 * selection, INCLUDING the size-dependent tie rule (block = 2^floor(log2 n_valid[b])), is that of the unpadded cloud.
 * snap (B,n) fp32 + first_tie (B) int32 (optional, together): at the first round with a tie, that round's number and the
 * min-distance state it started from (by point index) -- what rtk_fps_relevel needs to resume there.  Rows without a tie are not
 * written. */
RTK_EXPORT int rtk_fps_centroids(int b, int n, int npoint, const float *xyz, int *idx, float *new_xyz, int *nuniq,
                                 int *tie, const int *n_valid, float *snap, int *first_tie, rtk_stream_t stream);

/* rtk_knn_point over padded batches: only points[b][0 .. n_valid[b]) are candidates (n_valid (B) int32, >= k). */
RTK_EXPORT int rtk_knn_point_masked(int b, int s, int n, int k, const float *query, const float *points, int64_t *idx,
                                    const int *n_valid, rtk_stream_t stream);

/* Levels 2 .. 1+levels of a PNHead (model_utils.py:415-417): furthest point sampling of npoint out of the npoint
 * centroids of the previous level, starting from the level-1 centroids xyz1 (B,npoint,3) with their counters
 * nuniq1 / tie (B) from rtk_fps_centroids.  One launch per level; decided per cloud on the device: a cloud whose previous level
 * had no tie is the identity on the coordinates (proof at rtk_fps_relevel in csrc/ops_pointnet2.hip) and is only copied, a tied
 * cloud runs the selection of the level-1 kernel -- stopping as soon as, past the previous level's last tie, the picked set is a
 * prefix again (the rest is then the identity).  idx (levels,B,npoint) int32, new_xyz (levels,B,npoint,3), nuniq (levels,B);
 * tie_work (levels,B) int32: the levels' own tie counters (required for levels > 1).
 * idx1 (B,npoint) / snap1 (B,snap_pitch) / first_tie1 (B) from rtk_fps_centroids (optional, together; levels <= 2): tied clouds
 * resume at level 1's first tied round from the state saved there instead of at round 1. */
RTK_EXPORT int rtk_fps_relevel(int b, int npoint, int levels, const float *xyz1, const int *nuniq1, const int *tie,
                               int *idx, float *new_xyz, int *nuniq, int *tie_work, const int *idx1, const float *snap1,
                               int snap_pitch, const int *first_tie1, rtk_stream_t stream);

/* One time step of nn.GRU(hidden, hidden, layers) on a length-1 sequence (model_utils.py:279,296).
 * x (B,H); h_in, h_out (L,B,H); w_ih, w_hh TRANSPOSED (L,H,3H) = weight_{ih,hh}_l{l}.T, gate order (r,z,n);
 * b_ih, b_hh (L,3H); y (B,H) = h_out[L-1].  H <= 128. */
RTK_EXPORT int rtk_gru_step(int b, int layers, int hidden, const float *x, const float *h_in, const float *w_ih,
                            const float *w_hh, const float *b_ih, const float *b_hh, float *h_out, float *y,
                            rtk_stream_t stream);

/* rtk_gru_step with an epilogue: head_out (B, head_cout) = head_w y + head_bias, head_wt = head_w TRANSPOSED (H, head_cout) -- the
 * per-sample bias the flow head's first layer takes from the GRU output (model_utils.py:297-300).  H % 16 == 0. */
RTK_EXPORT int rtk_gru_step_head(int b, int layers, int hidden, const float *x, const float *h_in, const float *w_ih,
                                 const float *w_hh, const float *b_ih, const float *b_hh, float *h_out, float *y, const float *head_wt,
                                 const float *head_bias, float *head_out, int head_cout, rtk_stream_t stream);

/* Everything that is a function of a sample's global (max-pooled) feature g (samples, cin) in one launch (models/track4d.py:89-95
 * broadcasts it over the points and concatenates it to per-point inputs; a concatenated global half of a layer's input is a
 * per-sample bias of that layer): jobs[j]: out[(s - s0), :cout] = W g[s] + bias for s in [s0, s0 + count), wt = W TRANSPOSED
 * (cin, cout) fp32; and, with bcast, bcast[(s n + r) bcast_pitch + c] = g[s][c] for every row r < n of every sample.  cin % 32 == 0. */
#define RTK_GT_MAX_JOBS 4
typedef struct {
    const float *wt, *bias;      /* (cin, cout), (cout) or NULL */
    float *out;                  /* (count, out_pitch) */
    int cout, s0, count, out_pitch;
} rtk_gterm_job_t;
RTK_EXPORT int rtk_global_terms(int samples, int cin, const float *g, int njobs, const rtk_gterm_job_t *jobs, float *bcast, int bcast_pitch,
                                int n, rtk_stream_t stream);

/* Up to RTK_COPY_MAX_JOBS device-to-device copies in one launch (sizes in bytes, multiples of 4; pointers 4-byte aligned). */
#define RTK_COPY_MAX_JOBS 8
typedef struct {
    const void *src;
    void *dst;
    long bytes;
} rtk_copy_job_t;
RTK_EXPORT int rtk_copy_multi(int njobs, const rtk_copy_job_t *jobs, rtk_stream_t stream);

/* (rows = samples*n, pitch) point-major -> (samples, channels, n) channel-major at channel offset
 * dst_channel_offset of a (samples, dst_channels, n) tensor; per_sample != 0 broadcasts a (samples, pitch)
 * source over the n points (the global-feature halves of pc{1,2}_features, models/track4d.py:89-95). */
RTK_EXPORT int rtk_to_channel_major(int samples, int n, int channels, const float *src, int src_pitch, int per_sample,
                                    float *dst, int dst_channels, int dst_channel_offset, rtk_stream_t stream);

/* Both ball queries of one MSG level in a single scan of the source cloud (lib/pointnet2_modules.py:37-38 issues
 * one ball_query per scale over the same centroids): identical results to two rtk_ball_query calls with
 * (radius1, nsample1) and (radius2, nsample2), radius1 <= radius2.  idx1/idx2 zero-initialised by the caller.
 * nuniq (optional, (B)): centroids >= nuniq[b] (duplicates of centroid 0, see rtk_fps_centroids) are skipped -- their rows keep the
 * caller's zeros, except that the rows up to the next multiple of 32 are WRITTEN as zeros (round 6: a consumer's last tile may load
 * them before it masks them; rtk_geometry_tables relies on this instead of a zero-filled workspace).  An empty ball's row is written
 * as zeros as well (what the caller's initialisation leaves there in the reference). */
RTK_EXPORT int rtk_ball_query_pair(int b, int n, int npoint, float radius1, int nsample1, float radius2, int nsample2,
                                   const float *new_xyz, const float *xyz, int *idx1, int *idx2, const int *nuniq,
                                   rtk_stream_t stream);

/* rtk_three_nn restricted to the non-duplicate unknown rows (rows >= unknown_nuniq[b] are not computed) and aware of
 * duplicate known rows (known rows >= known_nuniq[b] are copies of known row 0: only the unique prefix is scanned,
 * the result is identical to the full scan).  Either counter may be NULL. */
RTK_EXPORT int rtk_three_nn_masked(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx,
                                   const int *unknown_nuniq, const int *known_nuniq, rtk_stream_t stream);

/* The geometry of a batch of frame pairs in two launches (round 6) -- bit for bit the tables of the eleven launches they replace
 * (sampling_gpu.cu:94-209 x 3 levels, ball_query_gpu.cu:9-45 x 6, interpolate_gpu.cu:81-124 x 3, model_utils.py:85-99 x 2).
 *
 * rtk_geometry_front: everything that depends on the input clouds only.  frame1 / frame2: the b clouds of each frame,
 * channel_major != 0: the API's (B,3,n) tensors, else point-major (B,n,3); clouds = 2 b, or b (frame 1 only, frame2 unused, no kNN; S below
 * stands for `clouds`).  With xyz / raw (together, with the (B,2,n) features):
 * rtk_prepare_inputs' outputs.  The three PNHead levels of rtk_fps_centroids + rtk_fps_relevel: fps_idx (3,S,npoint) int32,
 * new_xyz (3,S,npoint,3), nuniq (3,S), tie (3,S), first_tie (S) (zero-initialised), snap (S,n).  n_valid (S) optional.
 * knn12 / knn11 (B,n,16) int64 (optional, together): rtk_knn_point_masked of frame 1 in frame 2 / in frame 1, k = 16.
 * q1_w (q1_cout, 2) / q1_out (S n, q1_cout) (optional, together, with xyz / raw): q1_out = q1_w . (feature row) -- the encoder's first
 * per-point projection (lib/pointnet2_modules.py:37-53: layer 1 of sa1's MLPs restricted to the two feature channels), no bias.
 * n <= 2048, npoint <= 512. */
RTK_EXPORT int rtk_geometry_front(int b, int clouds, int n, int npoint, const float *frame1, const float *frame2, int channel_major,
                                  const float *feature1, const float *feature2, float *xyz, float *raw, int *fps_idx, float *new_xyz,
                                  int *nuniq, int *tie, int *first_tie, float *snap, const int *n_valid, int64_t *knn12, int64_t *knn11,
                                  const float *q1_w, float *q1_out, int q1_cout, rtk_stream_t stream);
/* rtk_geometry_tables: the six ball queries (rtk_ball_query_pair per level) and the three three-NN tables (rtk_three_nn_masked) of a
 * PNHead.  xyz0 (S,n,3) point-major clouds; new_xyz / nuniq as written by rtk_geometry_front.  radii, nsamples: HOST arrays of six
 * (level-major, scale 0 then 1, radii ascending within a level); ball: host array of six device tables (S,npoint,nsample) int32,
 * zero-initialised by the caller; nn_idx / nn_dist2: host arrays of three device tables [fp3, fp2, fp1]: (S,npoint,3) x 2, (S,n,3). */
RTK_EXPORT int rtk_geometry_tables(int samples, int n, int npoint, const float *xyz0, const float *new_xyz, const int *nuniq,
                                   const float *radii, const int *nsamples, int *const *ball, int *const *nn_idx, float *const *nn_dist2,
                                   rtk_stream_t stream);

/* Several rtk_to_channel_major jobs in one launch. */
typedef struct {
    const float *src; float *dst;
    int channels, src_pitch, per_sample, dst_channels, dst_channel_offset;
} rtk_layout_job_t;
RTK_EXPORT int rtk_to_channel_major_multi(int samples, int n, int njobs, const rtk_layout_job_t *jobs, rtk_stream_t stream);

/* Object association (models/track4d.py:166-180, models/utils/track4d_utils.py:405-434): log-space Sinkhorn normalisation
 * with a dustbin row and column, all `iters` iterations in one launch.  scores (m,n) fp32 (previous x current object
 * affinities), alpha the dustbin score; out (m+1, n+1) = log_optimal_transport(scores, alpha, iters).  One workgroup:
 * (m+1)(n+2) floats must fit 64 KiB of LDS (m, n up to ~120 objects). */
RTK_EXPORT int rtk_log_sinkhorn(int m, int n, const float *scores, float alpha, int iters, float *out, rtk_stream_t stream);

/* Moving-point clustering (models/track4d.py:108-126; sklearn.cluster.DBSCAN(eps, min_samples) in the reference, on the host).
 * feat: channel-major (C, pitch) per-point tensor of ONE frame; channels (8) int32 DEVICE array = the feature channels
 * entering the distance; score (n): a point takes part iff score > threshold (the motion-segmentation probability).
 * labels (n) int32: sklearn's cluster ids restricted to the participating points (numbered by first core point), -1 = noise or
 * not participating.  One workgroup; its tables live in LDS up to ~2900 points and in a stream-ordered global workspace beyond
 * (n <= 65536; all pairs are tested by that one workgroup). */
RTK_EXPORT int rtk_dbscan(int n, const float *feat, int pitch, const int *channels, const float *score, float threshold, double eps,
                          int min_samples, int *labels, rtk_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RTK_FUSED_H */
