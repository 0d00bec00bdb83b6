"""ctypes front-end of the CPU oracle (oracle/pointnet2_ref.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, tools/make_golden.py, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by anything under ratrack_amd/.

The ten `*_wrapper` functions keep the names, argument order and caller-allocates convention of the
reference's pybind module `pointnet2_cuda` (/root/reference/src/lib/src/pointnet2_api.cpp:10-25) but
take CPU tensors, so this module can stand in for `pointnet2_cuda` when the reference's Python graph
is imported in the build container (tools/make_golden.py).
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libpointnet2_ref.so")


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (idempotent)."""
    src = os.path.join(_HERE, "pointnet2_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _fp(t):
    assert t.dtype == torch.float32 and t.is_contiguous() and t.device.type == "cpu", (t.dtype, t.device)
    return ctypes.cast(t.data_ptr(), ctypes.POINTER(ctypes.c_float))


def _ip(t):
    assert t.dtype == torch.int32 and t.is_contiguous() and t.device.type == "cpu", (t.dtype, t.device)
    return ctypes.cast(t.data_ptr(), ctypes.POINTER(ctypes.c_int))


def _lp(t):
    assert t.dtype == torch.int64 and t.is_contiguous() and t.device.type == "cpu", (t.dtype, t.device)
    return ctypes.cast(t.data_ptr(), ctypes.POINTER(ctypes.c_int64))


# ---- the pybind surface of pointnet2_cuda (pointnet2_api.cpp:10-25) ------------------------------

def furthest_point_sampling_wrapper(b, n, m, points, temp, idx):
    return lib().rtk_ref_furthest_point_sampling(b, n, m, _fp(points), _fp(temp), _ip(idx))


def gather_points_wrapper(b, c, n, npoints, points, idx, out):
    return lib().rtk_ref_gather_points(b, c, n, npoints, _fp(points), _ip(idx), _fp(out))


def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
    return lib().rtk_ref_gather_points_grad(b, c, n, npoints, _fp(grad_out), _ip(idx), _fp(grad_points))


def ball_query_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx):
    return lib().rtk_ref_ball_query(b, n, m, ctypes.c_float(radius), nsample, _fp(new_xyz), _fp(xyz), _ip(idx))


def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
    return lib().rtk_ref_group_points(b, c, n, npoints, nsample, _fp(points), _ip(idx), _fp(out))


def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
    return lib().rtk_ref_group_points_grad(b, c, n, npoints, nsample, _fp(grad_out), _ip(idx), _fp(grad_points))


def three_nn_wrapper(b, n, m, unknown, known, dist2, idx):
    return lib().rtk_ref_three_nn(b, n, m, _fp(unknown), _fp(known), _fp(dist2), _ip(idx))


def knn_wrapper(b, n, m, k, unknown, known, dist2, idx):
    return lib().rtk_ref_knn(b, n, m, k, _fp(unknown), _fp(known), _fp(dist2), _ip(idx))


def three_interpolate_wrapper(b, c, m, n, points, idx, weight, out):
    return lib().rtk_ref_three_interpolate(b, c, m, n, _fp(points), _ip(idx), _fp(weight), _fp(out))


def three_interpolate_grad_wrapper(b, c, n, m, grad_out, idx, weight, grad_points):
    return lib().rtk_ref_three_interpolate_grad(b, c, n, m, _fp(grad_out), _ip(idx), _fp(weight), _fp(grad_points))


# ---- convenience (allocating) forms used by the oracle graph and the tests -------------------------

def fps(xyz, npoint):
    """xyz (B,N,3) -> idx int32 (B,npoint).  lib/pointnet2_utils.py:12-29."""
    B, N, _ = xyz.shape
    out = torch.empty(B, npoint, dtype=torch.int32)
    temp = torch.full((B, N), 1e10, dtype=torch.float32)
    furthest_point_sampling_wrapper(B, N, npoint, xyz.contiguous(), temp, out)
    return out


def gather(features, idx):
    """features (B,C,N), idx (B,M) -> (B,C,M)."""
    B, C, N = features.shape
    M = idx.shape[1]
    out = torch.empty(B, C, M, dtype=torch.float32)
    gather_points_wrapper(B, C, N, M, features.contiguous(), idx.contiguous(), out)
    return out


def ball_query(radius, nsample, xyz, new_xyz):
    """xyz (B,N,3), new_xyz (B,M,3) -> idx int32 (B,M,nsample), zero-initialised (pointnet2_utils.py:246)."""
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = torch.zeros(B, M, nsample, dtype=torch.int32)
    ball_query_wrapper(B, N, M, float(radius), nsample, new_xyz.contiguous(), xyz.contiguous(), idx)
    return idx


def group(features, idx):
    """features (B,C,N), idx (B,M,ns) -> (B,C,M,ns)."""
    B, C, N = features.shape
    _, M, ns = idx.shape
    out = torch.empty(B, C, M, ns, dtype=torch.float32)
    group_points_wrapper(B, C, N, M, ns, features.contiguous(), idx.contiguous(), out)
    return out


def three_nn(unknown, known):
    """-> (dist2 (B,n,3) SQUARED distances, idx int32 (B,n,3))."""
    B, n, _ = unknown.shape
    m = known.shape[1]
    d2 = torch.empty(B, n, 3, dtype=torch.float32)
    idx = torch.empty(B, n, 3, dtype=torch.int32)
    three_nn_wrapper(B, n, m, unknown.contiguous(), known.contiguous(), d2, idx)
    return d2, idx


def three_interpolate(features, idx, weight):
    B, c, m = features.shape
    n = idx.shape[1]
    out = torch.empty(B, c, n, dtype=torch.float32)
    three_interpolate_wrapper(B, c, m, n, features.contiguous(), idx.contiguous(), weight.contiguous(), out)
    return out


def knn(k, unknown, known):
    """interpolate_gpu.cu:9-57 (the exported, dead-on-the-live-path kernel): -> (dist2 (B,n,k), idx int32 (B,n,k))."""
    B, n, _ = unknown.shape
    m = known.shape[1]
    d2 = torch.empty(B, n, k, dtype=torch.float32)
    idx = torch.empty(B, n, k, dtype=torch.int32)
    knn_wrapper(B, n, m, k, unknown.contiguous(), known.contiguous(), d2, idx)
    return d2, idx


def knn_point(nsample, xyz, new_xyz, with_dist=False):
    """model_utils.py:85-99: k nearest of `xyz` (B,N,3) for every `new_xyz` (B,S,3) under the
    expansion-formula distance; indices int64 (B,S,k) ordered by (distance, index)."""
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    idx = torch.empty(B, S, nsample, dtype=torch.int64)
    dist = torch.empty(B, S, nsample, dtype=torch.float32) if with_dist else None
    rc = lib().rtk_ref_knn_point(B, S, N, nsample, _fp(new_xyz.contiguous()), _fp(xyz.contiguous()), _lp(idx),
                                 _fp(dist) if with_dist else None)
    assert rc == 0
    return (idx, dist) if with_dist else idx
