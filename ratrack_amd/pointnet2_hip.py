"""`pointnet2_cuda` replacement: the reference's ten pybind entry points, same names, argument order
and caller-allocates convention (/root/reference/src/lib/src/pointnet2_api.cpp:10-25), routed to the
gfx950 kernels in librtk_hip.so on torch's current HIP stream.

A reference checkout can switch with one line in lib/pointnet2_utils.py:7 --
    import ratrack_amd.pointnet2_hip as pointnet2
(see INTEGRATION.md).  Every tensor must be a contiguous device tensor; anything else raises instead
of silently computing elsewhere.
"""
import torch

from . import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(t, dtype, name):
    if not torch.is_tensor(t) or t.device.type != "cuda":
        raise _lib.RtkError("%s must be a device (HIP) tensor, got %s" % (name, getattr(t, "device", type(t))))
    if t.dtype != dtype:
        raise _lib.RtkError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise _lib.RtkError("%s must be contiguous" % name)
    return t.data_ptr()


_f = lambda t, n: _chk(t, torch.float32, n)
_i = lambda t, n: _chk(t, torch.int32, n)


def furthest_point_sampling_wrapper(b, n, m, points, temp, idx):
    _lib.call("rtk_furthest_point_sampling", b, n, m, _f(points, "points"), _f(temp, "temp"), _i(idx, "idx"), _stream())
    return 1


def gather_points_wrapper(b, c, n, npoints, points, idx, out):
    _lib.call("rtk_gather_points", b, c, n, npoints, _f(points, "points"), _i(idx, "idx"), _f(out, "out"), _stream())
    return 1


def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
    _lib.call("rtk_gather_points_grad", b, c, n, npoints, _f(grad_out, "grad_out"), _i(idx, "idx"),
              _f(grad_points, "grad_points"), _stream())
    return 1


def ball_query_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx):
    _lib.call("rtk_ball_query", b, n, m, float(radius), nsample, _f(new_xyz, "new_xyz"), _f(xyz, "xyz"), _i(idx, "idx"),
              _stream())
    return 1


def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
    _lib.call("rtk_group_points", b, c, n, npoints, nsample, _f(points, "points"), _i(idx, "idx"), _f(out, "out"), _stream())
    return 1


def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
    _lib.call("rtk_group_points_grad", b, c, n, npoints, nsample, _f(grad_out, "grad_out"), _i(idx, "idx"),
              _f(grad_points, "grad_points"), _stream())
    return 1


def three_nn_wrapper(b, n, m, unknown, known, dist2, idx):
    _lib.call("rtk_three_nn", b, n, m, _f(unknown, "unknown"), _f(known, "known"), _f(dist2, "dist2"), _i(idx, "idx"), _stream())


def knn_wrapper(b, n, m, k, unknown, known, dist2, idx):
    _lib.call("rtk_knn", b, n, m, k, _f(unknown, "unknown"), _f(known, "known"), _f(dist2, "dist2"), _i(idx, "idx"), _stream())


def three_interpolate_wrapper(b, c, m, n, points, idx, weight, out):
    _lib.call("rtk_three_interpolate", b, c, m, n, _f(points, "points"), _i(idx, "idx"), _f(weight, "weight"), _f(out, "out"),
              _stream())


def three_interpolate_grad_wrapper(b, c, n, m, grad_out, idx, weight, grad_points):
    _lib.call("rtk_three_interpolate_grad", b, c, n, m, _f(grad_out, "grad_out"), _i(idx, "idx"), _f(weight, "weight"),
              _f(grad_points, "grad_points"), _stream())


def knn_point_wrapper(b, s, n, k, query, points, idx):
    """Not in the reference's pybind module: replaces the torch.topk-based knn_point()
    (utils/model_utils/model_utils.py:85-99).  idx int64 (B,S,k), caller-allocated."""
    _lib.call("rtk_knn_point", b, s, n, k, _f(query, "query"), _f(points, "points"), _chk(idx, torch.int64, "idx"), _stream())
