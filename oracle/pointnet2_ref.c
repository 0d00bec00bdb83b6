/*
 * oracle/pointnet2_ref.c -- CPU restatement of RaTrack's `pointnet2_cuda` native ops.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under ratrack_amd/ may import, link or call this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it (as the checker /
 * the timed CPU baseline, never as the product path).
 *
 * The reference ships these ops as CUDA only (src/lib/src/ *.cu -- needs nvcc, unbuildable here),
 * so there is no reference CPU build to link against ("parity unpinned" by the reference's own
 * tests).  Every function below follows the cited .cu kernel statement by
 * statement; thread/block structure is restated only where it decides the result (the FPS block
 * reduction, run as the literal halving tree).
 *
 * Arithmetic contract (mirrored bit-for-bit by the HIP kernels in ratrack_amd/csrc):
 *   squared distance  d2 = fmaf(dz,dz, fmaf(dy,dy, dx*dx))       -- the contraction nvcc's default
 *   --fmad=true applies to  dx*dx + dy*dy + dz*dz  (reference builds with plain -O2,
 *   src/lib/setup.py:19-20).  This file is compiled with -ffp-contract=off so that every fused
 *   operation is an explicit fmaf() and nothing else is contracted.
 *   knn_point distance  d = max(((-2*dot) + |s|^2) + |t|^2, 0) with dot = fmaf(z,z',fmaf(y,y',x*x'))
 *   and |p|^2 = (x*x + y*y) + z*z   (the op sequence of utils/model_utils/model_utils.py:17-39 as
 *   PyTorch-CPU evaluates it).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define RTK_API __attribute__((visibility("default")))

static inline float sqdist3(float ax, float ay, float az, float bx, float by, float bz) {
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

/* cuda_utils.h:10-14  opt_n_threads(): block = 2^floor(log2 n) clamped to [1,1024]; the float
 * log ratio is reproduced literally because it is host code in the reference as well. */
RTK_API int rtk_ref_fps_block_size(int n) {
    const int pow_2 = (int)(log((double)n) / log(2.0));
    int t = 1 << pow_2;
    if (t > 1024) t = 1024;
    if (t < 1) t = 1;
    return t;
}

/* sampling_gpu.cu:94-209 furthest_point_sampling_kernel<block_size>
 * dataset (B,N,3), temp (B,N) pre-filled by the caller (1e10, lib/pointnet2_utils.py:26), idxs (B,M).
 * Tie rule: thread tid scans k = tid, tid+block, ... keeping the FIRST maximum (strict >, :136-137);
 * the per-thread results then go through the shared-memory halving tree of :143-203, run literally
 * below: at stride s = block/2 ... 1 slot t < s absorbs slot t+s and KEEPS ITS OWN entry on equal values
 * (__update, :86-91: `v2 > v1 ? i2 : i1`).  Slot 0 therefore ends up with the maximum whose thread has the
 * smallest BIT-REVERSED tid (the last level prefers even tids over odd ones, the level before tids = 0 mod 4
 * over 2 mod 4, ...): e.g. tids 1 and 2 tied -> tid 2 wins.  Pinned by the literal SIMT emulation
 * tools/emulate_cu.py (tests/test_emulator_cpu.py). */
RTK_API int rtk_ref_furthest_point_sampling(int b, int n, int m, const float *dataset, float *temp,
                                            int *idxs) {
    if (m <= 0) return 0;
    const int block = rtk_ref_fps_block_size(n);
    float *best_v = (float *)malloc(sizeof(float) * (size_t)block);
    int *best_i = (int *)malloc(sizeof(int) * (size_t)block);
    for (int bi = 0; bi < b; ++bi) {
        const float *pts = dataset + (size_t)bi * n * 3;
        float *tmp = temp + (size_t)bi * n;
        int *out = idxs + (size_t)bi * m;
        int old = 0;
        out[0] = old;
        for (int j = 1; j < m; ++j) {
            const float x1 = pts[old * 3 + 0], y1 = pts[old * 3 + 1], z1 = pts[old * 3 + 2];
            for (int tid = 0; tid < block; ++tid) {
                int besti = 0;
                float best = -1.f;
                for (int k = tid; k < n; k += block) {
                    const float d = sqdist3(pts[k * 3 + 0], pts[k * 3 + 1], pts[k * 3 + 2], x1, y1, z1);
                    const float d2 = fminf(d, tmp[k]);
                    tmp[k] = d2;
                    besti = d2 > best ? k : besti;
                    best = d2 > best ? d2 : best;
                }
                best_v[tid] = best;
                best_i[tid] = besti;
            }
            /* :143-203 the halving tree; __update(dists, dists_i, t, t + s) for t < s at every level */
            for (int s = block >> 1; s >= 1; s >>= 1)
                for (int t = 0; t < s; ++t) {
                    const float v1 = best_v[t], v2 = best_v[t + s];
                    const int i1 = best_i[t], i2 = best_i[t + s];
                    best_v[t] = v1 > v2 ? v1 : v2;      /* max(v1, v2), :89 (no NaNs: distances of finite points) */
                    best_i[t] = v2 > v1 ? i2 : i1;      /* :90 -- equal values keep slot t */
                }
            old = best_i[0];
            out[j] = old;
        }
    }
    free(best_v);
    free(best_i);
    return 0;
}

/* sampling_gpu.cu:8-24 gather_points_kernel_fast: out[b,c,j] = points[b,c,idx[b,j]] */
RTK_API int rtk_ref_gather_points(int b, int c, int n, int m, const float *points, const int *idx,
                                  float *out) {
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci) {
            const float *p = points + ((size_t)bi * c + ci) * n;
            float *o = out + ((size_t)bi * c + ci) * m;
            const int *id = idx + (size_t)bi * m;
            for (int j = 0; j < m; ++j) o[j] = p[id[j]];
        }
    return 0;
}

/* sampling_gpu.cu:46-63 gather_points_grad_kernel_fast: grad_points[b,c,idx[b,j]] += grad_out[b,c,j]
 * (atomicAdd in the reference; sequential j order here). grad_points is zero-initialised by the caller. */
RTK_API int rtk_ref_gather_points_grad(int b, int c, int n, int m, const float *grad_out,
                                       const int *idx, float *grad_points) {
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci) {
            float *g = grad_points + ((size_t)bi * c + ci) * n;
            const float *go = grad_out + ((size_t)bi * c + ci) * m;
            const int *id = idx + (size_t)bi * m;
            for (int j = 0; j < m; ++j) g[id[j]] += go[j];
        }
    return 0;
}

/* ball_query_gpu.cu:9-45 ball_query_kernel_fast.  idx (B,M,nsample) is ZERO-INITIALISED BY THE
 * CALLER (lib/pointnet2_utils.py:246): an empty ball leaves the row untouched (all zeros). */
RTK_API int rtk_ref_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                               const float *xyz, int *idx) {
    const float radius2 = radius * radius;
    for (int bi = 0; bi < b; ++bi)
        for (int pt = 0; pt < m; ++pt) {
            const float *q = new_xyz + ((size_t)bi * m + pt) * 3;
            const float *p = xyz + (size_t)bi * n * 3;
            int *o = idx + ((size_t)bi * m + pt) * nsample;
            int cnt = 0;
            for (int k = 0; k < n; ++k) {
                const float d2 = sqdist3(q[0], q[1], q[2], p[k * 3 + 0], p[k * 3 + 1], p[k * 3 + 2]);
                if (d2 < radius2) {
                    if (cnt == 0)
                        for (int l = 0; l < nsample; ++l) o[l] = k;
                    o[cnt] = k;
                    ++cnt;
                    if (cnt >= nsample) break;
                }
            }
        }
    return 0;
}

/* group_points_gpu.cu:47-66 group_points_kernel_fast: out[b,c,j,l] = points[b,c,idx[b,j,l]] */
RTK_API int rtk_ref_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                                 const int *idx, float *out) {
    const size_t sn = (size_t)npoints * nsample;
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci) {
            const float *p = points + ((size_t)bi * c + ci) * n;
            float *o = out + ((size_t)bi * c + ci) * sn;
            const int *id = idx + (size_t)bi * sn;
            for (size_t t = 0; t < sn; ++t) o[t] = p[id[t]];
        }
    return 0;
}

/* group_points_gpu.cu:8-25 group_points_grad_kernel_fast (atomicAdd -> sequential order). */
RTK_API int rtk_ref_group_points_grad(int b, int c, int n, int npoints, int nsample,
                                      const float *grad_out, const int *idx, float *grad_points) {
    const size_t sn = (size_t)npoints * nsample;
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci) {
            float *g = grad_points + ((size_t)bi * c + ci) * n;
            const float *go = grad_out + ((size_t)bi * c + ci) * sn;
            const int *id = idx + (size_t)bi * sn;
            for (size_t t = 0; t < sn; ++t) g[id[t]] += go[t];
        }
    return 0;
}

/* interpolate_gpu.cu:81-124 three_nn_kernel_fast.  best* are doubles initialised to 1e40, the
 * float distance is compared against them with strict < (earliest index wins ties) and the result
 * is narrowed back to float on store (:122) -- with fewer than 3 known points the unfilled slots
 * become +inf with index 0. */
RTK_API int rtk_ref_three_nn(int b, int n, int m, const float *unknown, const float *known,
                             float *dist2, int *idx) {
    for (int bi = 0; bi < b; ++bi)
        for (int pt = 0; pt < n; ++pt) {
            const float *u = unknown + ((size_t)bi * n + pt) * 3;
            const float *kn = known + (size_t)bi * m * 3;
            double best1 = 1e40, best2 = 1e40, best3 = 1e40;
            int besti1 = 0, besti2 = 0, besti3 = 0;
            for (int k = 0; k < m; ++k) {
                const float d = sqdist3(u[0], u[1], u[2], kn[k * 3 + 0], kn[k * 3 + 1], kn[k * 3 + 2]);
                if (d < best1) {
                    best3 = best2; besti3 = besti2;
                    best2 = best1; besti2 = besti1;
                    best1 = d; besti1 = k;
                } else if (d < best2) {
                    best3 = best2; besti3 = besti2;
                    best2 = d; besti2 = k;
                } else if (d < best3) {
                    best3 = d; besti3 = k;
                }
            }
            float *od = dist2 + ((size_t)bi * n + pt) * 3;
            int *oi = idx + ((size_t)bi * n + pt) * 3;
            od[0] = (float)best1; od[1] = (float)best2; od[2] = (float)best3;
            oi[0] = besti1; oi[1] = besti2; oi[2] = besti3;
        }
    return 0;
}

/* interpolate_gpu.cu:9-57 knn_kernel_fast: k smallest by insertion into an ascending list, strict <
 * (earliest index wins ties); k <= 200 is silently assumed by the reference (double best[200]). */
RTK_API int rtk_ref_knn(int b, int n, int m, int k, const float *unknown, const float *known,
                        float *dist2, int *idx) {
    if (k > 200 || k < 1) return -1;
    for (int bi = 0; bi < b; ++bi)
        for (int pt = 0; pt < n; ++pt) {
            const float *u = unknown + ((size_t)bi * n + pt) * 3;
            const float *kn = known + (size_t)bi * m * 3;
            double best[200];
            int besti[200];
            for (int i = 0; i < k; ++i) { best[i] = 1e40; besti[i] = 0; }
            for (int i = 0; i < m; ++i) {
                const float d = sqdist3(u[0], u[1], u[2], kn[i * 3 + 0], kn[i * 3 + 1], kn[i * 3 + 2]);
                for (int j = 0; j < k; ++j) {
                    if (d < best[j]) {
                        for (int l = k - 1; l > j; --l) { best[l] = best[l - 1]; besti[l] = besti[l - 1]; }
                        best[j] = d;
                        besti[j] = i;
                        break;
                    }
                }
            }
            float *od = dist2 + ((size_t)bi * n + pt) * k;
            int *oi = idx + ((size_t)bi * n + pt) * k;
            for (int i = 0; i < k; ++i) { oi[i] = besti[i]; od[i] = (float)best[i]; }
        }
    return 0;
}

/* interpolate_gpu.cu:149-169 three_interpolate_kernel_fast:
 * out[b,c,j] = w0*p[i0] + w1*p[i1] + w2*p[i2], evaluated left to right; nvcc contracts it to
 * fmaf(w2,p2, fmaf(w1,p1, w0*p0)). */
RTK_API int rtk_ref_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                                      const float *weight, float *out) {
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci) {
            const float *p = points + ((size_t)bi * c + ci) * m;
            float *o = out + ((size_t)bi * c + ci) * n;
            for (int j = 0; j < n; ++j) {
                const float *w = weight + ((size_t)bi * n + j) * 3;
                const int *id = idx + ((size_t)bi * n + j) * 3;
                o[j] = fmaf(w[2], p[id[2]], fmaf(w[1], p[id[1]], w[0] * p[id[0]]));
            }
        }
    return 0;
}

/* interpolate_gpu.cu:192-214 three_interpolate_grad_kernel_fast (3 atomicAdds -> sequential). */
RTK_API int rtk_ref_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                           const int *idx, const float *weight, float *grad_points) {
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci) {
            float *g = grad_points + ((size_t)bi * c + ci) * m;
            const float *go = grad_out + ((size_t)bi * c + ci) * n;
            for (int j = 0; j < n; ++j) {
                const float *w = weight + ((size_t)bi * n + j) * 3;
                const int *id = idx + ((size_t)bi * n + j) * 3;
                g[id[0]] += go[j] * w[0];
                g[id[1]] += go[j] * w[1];
                g[id[2]] += go[j] * w[2];
            }
        }
    return 0;
}

/* utils/model_utils/model_utils.py:17-39,85-99  knn_point = square_distance (expansion formula)
 * + torch.topk(largest=False, sorted=False).  query (B,S,3) = `new_xyz`, points (B,N,3) = `xyz`.
 * Output: for every query the k nearest indices ordered by (distance, index) ascending -- torch's
 * own order is unspecified, consumers only sum over the neighbour axis (model_utils.py:236,248),
 * so parity is defined on the index SET.  dist_out (optional) receives the k distances. */
RTK_API int rtk_ref_knn_point(int b, int s, int n, int k, const float *query, const float *points,
                              int64_t *idx_out, float *dist_out) {
    if (k < 1 || k > n) return -1;
    float *d = (float *)malloc(sizeof(float) * (size_t)n);
    float *pn = (float *)malloc(sizeof(float) * (size_t)n);
    int *sel = (int *)malloc(sizeof(int) * (size_t)k);
    for (int bi = 0; bi < b; ++bi) {
        const float *P = points + (size_t)bi * n * 3;
        for (int j = 0; j < n; ++j)
            pn[j] = (P[j * 3] * P[j * 3] + P[j * 3 + 1] * P[j * 3 + 1]) + P[j * 3 + 2] * P[j * 3 + 2];
        for (int qi = 0; qi < s; ++qi) {
            const float *q = query + ((size_t)bi * s + qi) * 3;
            const float qn = (q[0] * q[0] + q[1] * q[1]) + q[2] * q[2];
            for (int j = 0; j < n; ++j) {
                const float dot = fmaf(q[2], P[j * 3 + 2], fmaf(q[1], P[j * 3 + 1], q[0] * P[j * 3]));
                float v = (-2.f * dot + qn) + pn[j];
                d[j] = v > 0.f ? v : 0.f; /* torch.maximum(dist, 0) */
            }
            /* k-smallest by (d, index): insertion into an ascending list */
            int cnt = 0;
            for (int j = 0; j < n; ++j) {
                if (cnt == k && !(d[j] < d[sel[k - 1]])) continue;
                int pos = cnt < k ? cnt : k - 1;
                while (pos > 0 && d[j] < d[sel[pos - 1]]) { sel[pos] = sel[pos - 1]; --pos; }
                sel[pos] = j;
                if (cnt < k) ++cnt;
            }
            int64_t *oi = idx_out + ((size_t)bi * s + qi) * k;
            for (int t = 0; t < k; ++t) oi[t] = sel[t];
            if (dist_out) {
                float *od = dist_out + ((size_t)bi * s + qi) * k;
                for (int t = 0; t < k; ++t) od[t] = d[sel[t]];
            }
        }
    }
    free(d);
    free(pn);
    free(sel);
    return 0;
}
