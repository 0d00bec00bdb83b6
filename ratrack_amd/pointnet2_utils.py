"""Autograd front-ends of the native ops -- the counterpart of the reference's
lib/pointnet2_utils.py with the same public names and call signatures
(furthest_point_sample, gather_operation, three_nn, three_interpolate, grouping_operation,
ball_query, knn, QueryAndGroup, GroupAll), backed by the gfx950 kernels.

Autograd contract: sampling / query ops are non-differentiable; gather, group and
three_interpolate differentiate w.r.t. the features only (fp32 atomics in the scatter kernels).
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib
from . import pointnet2_hip as _native


def _new(like, shape, dtype):
    return torch.empty(shape, dtype=dtype, device=like.device)


class FurthestPointSampling(Function):
    """xyz (B,N,3) -> int32 (B,npoint).  lib/pointnet2_utils.py:10-36."""

    @staticmethod
    def forward(ctx, xyz, npoint):
        assert xyz.is_contiguous()
        B, N, _ = xyz.shape
        out = _new(xyz, (B, npoint), torch.int32)
        temp = torch.full((B, N), 1e10, dtype=torch.float32, device=xyz.device)
        _native.furthest_point_sampling_wrapper(B, N, npoint, xyz, temp, out)
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, grad=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    """features (B,C,N), idx (B,npoint) -> (B,C,npoint).  lib/pointnet2_utils.py:39-73."""

    @staticmethod
    def forward(ctx, features, idx):
        assert features.is_contiguous() and idx.is_contiguous()
        B, npoint = idx.shape
        _, C, N = features.shape
        out = _new(features, (B, C, npoint), torch.float32)
        _native.gather_points_wrapper(B, C, N, npoint, features, idx, out)
        ctx.save_for_backward(idx)
        ctx.dims = (C, N)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        C, N = ctx.dims
        B, npoint = idx.shape
        grad = torch.zeros((B, C, N), dtype=torch.float32, device=grad_out.device)
        _native.gather_points_grad_wrapper(B, C, N, npoint, grad_out.contiguous(), idx, grad)
        return grad, None


gather_operation = GatherOperation.apply


class KNN(Function):
    """k nearest of `known` for each `unknown` (direct-difference distance); returns (sqrt(d2), idx).
    lib/pointnet2_utils.py:75-102 -- exported by the reference, unused on its live path."""

    @staticmethod
    def forward(ctx, k, unknown, known):
        assert unknown.is_contiguous() and known.is_contiguous()
        B, N, _ = unknown.shape
        m = known.shape[1]
        dist2 = _new(unknown, (B, N, k), torch.float32)
        idx = _new(unknown, (B, N, k), torch.int32)
        _native.knn_wrapper(B, N, m, k, unknown, known, dist2, idx)
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None


knn = KNN.apply


class ThreeNN(Function):
    """-> (dist (B,n,3) = sqrt of squared distance, idx int32 (B,n,3)).  lib/pointnet2_utils.py:104-133."""

    @staticmethod
    def forward(ctx, unknown, known):
        assert unknown.is_contiguous() and known.is_contiguous()
        B, N, _ = unknown.shape
        m = known.shape[1]
        dist2 = _new(unknown, (B, N, 3), torch.float32)
        idx = _new(unknown, (B, N, 3), torch.int32)
        _native.three_nn_wrapper(B, N, m, unknown, known, dist2, idx)
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    """features (B,C,M), idx (B,n,3), weight (B,n,3) -> (B,C,n).  lib/pointnet2_utils.py:136-181."""

    @staticmethod
    def forward(ctx, features, idx, weight):
        assert features.is_contiguous() and idx.is_contiguous() and weight.is_contiguous()
        B, c, m = features.shape
        n = idx.shape[1]
        out = _new(features, (B, c, n), torch.float32)
        _native.three_interpolate_wrapper(B, c, m, n, features, idx, weight, out)
        ctx.save_for_backward(idx, weight)
        ctx.m = m
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        B, c, n = grad_out.shape
        # the reference zero-fills and lets the kernel accumulate (lib/pointnet2_utils.py:176-177); the *_set entry point
        # writes the same values into an uninitialised buffer
        grad = torch.empty((B, c, ctx.m), dtype=torch.float32, device=grad_out.device)
        _lib.call("rtk_three_interpolate_grad_set", B, c, n, ctx.m, grad_out.contiguous().data_ptr(), idx.data_ptr(), weight.data_ptr(),
                  grad.data_ptr(), torch.cuda.current_stream().cuda_stream)
        return grad, None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    """features (B,C,N), idx (B,npoint,nsample) -> (B,C,npoint,nsample).  lib/pointnet2_utils.py:184-225."""

    @staticmethod
    def forward(ctx, features, idx):
        assert features.is_contiguous() and idx.is_contiguous()
        idx = idx.int()
        B, npoint, nsample = idx.shape
        _, C, N = features.shape
        out = _new(features, (B, C, npoint, nsample), torch.float32)
        _native.group_points_wrapper(B, C, N, npoint, nsample, features, idx, out)
        ctx.save_for_backward(idx)
        ctx.N = N
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        B, C, npoint, nsample = grad_out.shape
        grad = torch.empty((B, C, ctx.N), dtype=torch.float32, device=grad_out.device)      # see ThreeInterpolate.backward
        _lib.call("rtk_group_points_grad_set", B, C, ctx.N, npoint, nsample, grad_out.contiguous().data_ptr(), idx.data_ptr(),
                  grad.data_ptr(), torch.cuda.current_stream().cuda_stream)
        return grad, None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    """-> int32 (B,npoint,nsample), zero-initialised here as in lib/pointnet2_utils.py:246."""

    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        assert xyz.is_contiguous() and new_xyz.is_contiguous()
        B, N, _ = xyz.shape
        npoint = new_xyz.shape[1]
        idx = torch.zeros((B, npoint, nsample), dtype=torch.int32, device=xyz.device)
        _native.ball_query_wrapper(B, N, npoint, radius, nsample, new_xyz, xyz, idx)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


def knn_point(nsample, xyz, new_xyz):
    """Drop-in for utils/model_utils/model_utils.py:85-99: indices int64 (B,S,nsample) of the
    `nsample` nearest `xyz` points of every `new_xyz` under the expansion-formula distance.
    The reference's neighbour order is unspecified (topk sorted=False); ours is (distance, index)."""
    xyz, new_xyz = xyz.contiguous(), new_xyz.contiguous()
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    idx = torch.empty((B, S, nsample), dtype=torch.int64, device=xyz.device)
    _native.knn_point_wrapper(B, S, N, nsample, new_xyz.detach(), xyz.detach(), idx)
    return idx


class QueryAndGroup(nn.Module):
    """ball query + grouping: (B,N,3),(B,npoint,3),(B,C,N) -> (B,3+C,npoint,nsample) with channel
    order [xyz - centroid || features].  lib/pointnet2_utils.py:259-292."""

    def __init__(self, radius, nsample, use_xyz=True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, new_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        grouped_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)
        grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            return grouped_xyz
        grouped = grouping_operation(features, idx)
        return torch.cat([grouped_xyz, grouped], dim=1) if self.use_xyz else grouped


class GroupAll(nn.Module):
    """lib/pointnet2_utils.py:295-318."""

    def __init__(self, use_xyz=True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return grouped_xyz
        grouped = features.unsqueeze(2)
        return torch.cat([grouped_xyz, grouped], dim=1) if self.use_xyz else grouped
