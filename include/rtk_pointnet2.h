/*
 * rtk_pointnet2.h -- C ABI of librtk_hip.so, the MI355X (gfx950) replacement for RaTrack's
 * `pointnet2_cuda` extension and the fused stages built on top of it.
 *
 * Drop-in boundary.  Each entry point below replaces one pybind export of the
 * reference, /root/reference/src/lib/src/pointnet2_api.cpp:10-25, and keeps its conventions:
 *   - sizes first (int), then device pointers; fp32 data, int32 indices, dense row-major tensors;
 *   - OUTPUTS ARE CALLER-ALLOCATED device buffers (the reference's Python allocates them,
 *     lib/pointnet2_utils.py:25-26,55,93-94,122-123,156,200,246);
 *   - asynchronous on the stream handed in (the reference uses the framework's current stream,
 *     e.g. ball_query.cpp:27), no host synchronisation, no allocation.
 * Differences: the stream is an explicit last argument (a hipStream_t passed as void*), and launch
 * failures are RETURNED (0 = ok, <0 = rtk_status) instead of fprintf + exit(-1)
 * (sampling_gpu.cu:39-43 and every other launcher).  No torch types appear in any signature.
 *
 * Arithmetic contract (bit-exact with oracle/pointnet2_ref.c): squared distances are
 * fmaf(dz,dz, fmaf(dy,dy, dx*dx)); three_interpolate is fmaf(w2,p2, fmaf(w1,p1, w0*p0)).
 */
#ifndef RTK_POINTNET2_H
#define RTK_POINTNET2_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *rtk_stream_t; /* hipStream_t */

#define RTK_EXPORT __attribute__((visibility("default")))

enum rtk_status {
    RTK_OK = 0,
    RTK_ERR_INVALID = -1,     /* bad size / null pointer / unsupported shape */
    RTK_ERR_LAUNCH = -2,      /* hipGetLastError() != hipSuccess after the launch */
    RTK_ERR_UNSUPPORTED = -3  /* shape outside what the kernel was built for */
};

/* Human-readable text of the last error raised on the calling thread ("" if none). */
RTK_EXPORT const char *rtk_last_error(void);
/* Library/ABI version: (major << 16) | minor. */
RTK_EXPORT int rtk_version(void);

/* replaces furthest_point_sampling_wrapper   sampling.cpp:37-47 / sampling_gpu.cu:94-253
 * xyz (B,N,3); temp (B,N) scratch the caller pre-fills with 1e10 (lib/pointnet2_utils.py:26) --
 * read as the initial min-distance and left holding the final one; idx int32 (B,npoint).
 * The selection rule of the reference block reduction is reproduced exactly: each of the
 * block = 2^floor(log2 N) (capped at 1024, cuda_utils.h:10-14) threads keeps the first maximum of its
 * strided scan, and the shared-memory halving tree (sampling_gpu.cu:86-91,143-203) keeps slot t over
 * slot t+s on equal values, so ties go to the smallest (bitrev(k mod block), k div block). */
RTK_EXPORT int rtk_furthest_point_sampling(int b, int n, int npoint, const float *xyz, float *temp, int *idx,
                                rtk_stream_t stream);

/* replaces gather_points_wrapper / gather_points_grad_wrapper   sampling.cpp:12-34
 * points (B,C,N), idx (B,npoint) -> out (B,C,npoint); grad: grad_points (B,C,N) zero-initialised by
 * the caller, accumulated with fp32 atomics. */
RTK_EXPORT int rtk_gather_points(int b, int c, int n, int npoint, const float *points, const int *idx, float *out,
                      rtk_stream_t stream);
RTK_EXPORT int rtk_gather_points_grad(int b, int c, int n, int npoint, const float *grad_out, const int *idx,
                           float *grad_points, rtk_stream_t stream);

/* replaces ball_query_wrapper   ball_query.cpp:18-29 / ball_query_gpu.cu:9-45
 * new_xyz (B,npoint,3), xyz (B,N,3) -> idx int32 (B,npoint,nsample), ZERO-INITIALISED BY THE CALLER
 * (lib/pointnet2_utils.py:246): first `nsample` points in index order with d2 < radius^2, unfilled
 * slots repeat the first hit, an empty ball leaves the row untouched. */
RTK_EXPORT int rtk_ball_query(int b, int n, int npoint, float radius, int nsample, const float *new_xyz,
                   const float *xyz, int *idx, rtk_stream_t stream);

/* replaces group_points_wrapper / group_points_grad_wrapper   group_points.cpp:13-38
 * points (B,C,N), idx (B,npoint,nsample) -> out (B,C,npoint,nsample); grad accumulates into the
 * zero-initialised grad_points (B,C,N). */
RTK_EXPORT int rtk_group_points(int b, int c, int n, int npoint, int nsample, const float *points, const int *idx,
                     float *out, rtk_stream_t stream);
RTK_EXPORT int rtk_group_points_grad(int b, int c, int n, int npoint, int nsample, const float *grad_out,
                          const int *idx, float *grad_points, rtk_stream_t stream);
/* same result written (not accumulated) into an UNINITIALISED grad_points: saves the caller's zero-fill */
RTK_EXPORT int rtk_group_points_grad_set(int b, int c, int n, int npoint, int nsample, const float *grad_out,
                          const int *idx, float *grad_points, rtk_stream_t stream);

/* replaces three_nn_wrapper   interpolate.cpp:16-25 / interpolate_gpu.cu:81-124
 * unknown (B,n,3), known (B,m,3) -> dist2 (B,n,3) SQUARED distances ascending, idx int32 (B,n,3);
 * ties keep the earlier index; with m < 3 the unfilled slots are (+inf, 0). */
RTK_EXPORT int rtk_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx,
                 rtk_stream_t stream);

/* replaces knn_wrapper   interpolate.cpp:27-36 / interpolate_gpu.cu:9-57   (k <= 200 as there) */
RTK_EXPORT int rtk_knn(int b, int n, int m, int k, const float *unknown, const float *known, float *dist2, int *idx,
            rtk_stream_t stream);

/* replaces three_interpolate_wrapper / three_interpolate_grad_wrapper   interpolate.cpp:39-67
 * points (B,c,m), idx (B,n,3), weight (B,n,3) -> out (B,c,n); grad accumulates into the
 * zero-initialised grad_points (B,c,m). */
RTK_EXPORT int rtk_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                          const float *weight, float *out, rtk_stream_t stream);
RTK_EXPORT int rtk_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out, const int *idx,
                               const float *weight, float *grad_points, rtk_stream_t stream);
/* same result written (not accumulated) into an UNINITIALISED grad_points */
RTK_EXPORT int rtk_three_interpolate_grad_set(int b, int c, int n, int m, const float *grad_out, const int *idx,
                               const float *weight, float *grad_points, rtk_stream_t stream);

/* replaces knn_point()   utils/model_utils/model_utils.py:17-39,85-99  (square_distance + torch.topk)
 * query (B,S,3) = `new_xyz`, points (B,N,3) = `xyz` -> idx int64 (B,S,k): the k nearest under
 * d = max(((-2*dot) + |q|^2) + |p|^2, 0), ordered by (d, index) ascending.  1 <= k <= N (k <= 32: sorting networks, 4 lanes per
 * query; larger k: one wave per query, k selection rounds). */
RTK_EXPORT int rtk_knn_point(int b, int s, int n, int k, const float *query, const float *points, int64_t *idx,
                  rtk_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RTK_POINTNET2_H */
