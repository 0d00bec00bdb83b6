"""Autograd bindings of the training-mode kernels (include/rtk_train.h).

`bn_relu(z, bn, ...)` = nn.BatchNorm2d in training mode followed by ReLU (and, with pool=True, by the max over the
neighbourhood axis) -- the tail of every SharedMLP layer of the reference (lib/pytorch_utils.py:20-32,
lib/pointnet2_modules.py:44-47) -- on a de-duplicated, row-weighted tensor.  No CPU / eager fallback.
"""
import ctypes

import torch

from . import _lib

_i, _f, _d, _p = ctypes.c_int, ctypes.c_float, ctypes.c_double, ctypes.c_void_p
_lib.SIGNATURES.update({
    "rtk_bn_train_stats": [_i] * 5 + [_p] * 3 + [_p],
    "rtk_bn_train_finalize": [_i, _i, _p, _d, _p, _p, _f, _f, _p, _p, _p, _p, _p],
    "rtk_bn_relu_fwd": [_i] * 5 + [_p, _p, _i, _p, _p],
    "rtk_bn_relu_bwd_stats": [_i] * 5 + [_p, _p, _p, _i, _p, _p],
    "rtk_bn_relu_bwd_apply": [_i] * 5 + [_p] * 5 + [_d, _i, _p, _p, _p],
})


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return t.data_ptr() if t is not None else None


class _BNReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, gamma, beta, running_mean, running_var, nbt, row_weight, count, groups, eps, momentum, pool):
        assert z.is_cuda and z.dtype == torch.float32 and z.dim() == 4, "bn_relu: z must be a CUDA fp32 (S,C,rows,ns) tensor"
        z = z.contiguous()
        S_, C, rows, ns = z.shape
        if row_weight is not None:
            assert row_weight.shape == (S_, rows) and row_weight.dtype == torch.float32 and row_weight.is_contiguous()
        dev = z.device
        sums = torch.zeros(groups, C, 2, dtype=torch.float64, device=dev)
        _lib.call("rtk_bn_train_stats", S_, C, rows, ns, groups, z.data_ptr(), _ptr(row_weight), sums.data_ptr(), _stream())
        par = torch.empty(4, groups, C, dtype=torch.float32, device=dev)
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        _lib.call("rtk_bn_train_finalize", C, groups, sums.data_ptr(), float(count), g.data_ptr(), b.data_ptr(), float(eps),
                  float(momentum), _ptr(running_mean), _ptr(running_var), _ptr(nbt), par.data_ptr(), _stream())
        y = torch.empty((S_, C, rows) if pool else (S_, C, rows, ns), dtype=torch.float32, device=dev)
        _lib.call("rtk_bn_relu_fwd", S_, C, rows, ns, groups, z.data_ptr(), par.data_ptr(), int(pool), y.data_ptr(), _stream())
        ctx.save_for_backward(z, par, row_weight)
        ctx.cfg = (count, groups, pool)
        return y

    @staticmethod
    def backward(ctx, dy):
        z, par, row_weight = ctx.saved_tensors
        count, groups, pool = ctx.cfg
        S_, C, rows, ns = z.shape
        dy = dy.contiguous()
        dev = z.device
        sums2 = torch.zeros(groups, C, 2, dtype=torch.float64, device=dev)
        _lib.call("rtk_bn_relu_bwd_stats", S_, C, rows, ns, groups, z.data_ptr(), dy.data_ptr(), par.data_ptr(), int(pool),
                  sums2.data_ptr(), _stream())
        dz = torch.empty_like(z)
        dgb = torch.empty(2, C, dtype=torch.float32, device=dev)
        _lib.call("rtk_bn_relu_bwd_apply", S_, C, rows, ns, groups, z.data_ptr(), dy.data_ptr(), par.data_ptr(), _ptr(row_weight),
                  sums2.data_ptr(), float(count), int(pool), dz.data_ptr(), dgb.data_ptr(), _stream())
        return dz, dgb[0], dgb[1], None, None, None, None, None, None, None, None, None


def bn_relu(z, bn, row_weight=None, count=None, groups=1, pool=False):
    """z (S,C,rows,ns) -> relu(batch_norm_train(z)) of shape z, or (S,C,rows) with pool=True (max over ns).
    bn: the nn.BatchNorm2d whose weight / bias / running statistics are used and updated exactly as its own
    training-mode forward would (momentum, unbiased running variance, num_batches_tracked).
    row_weight (S,rows) / count: statistics weights of de-duplicated rows and the reference element count per
    channel and group (default: all ones, S/groups * rows * ns).  groups > 1: separate statistics for consecutive
    batch slices, running statistics updated slice by slice (= the reference's sequential per-frame calls)."""
    S_, C, rows, ns = z.shape
    if count is None:
        count = (S_ // groups) * rows * ns
    momentum = bn.momentum if bn.momentum is not None else 0.1
    track = bn.track_running_stats
    return _BNReLU.apply(z, bn.weight, bn.bias, bn.running_mean if track else None, bn.running_var if track else None,
                         bn.num_batches_tracked if track else None, row_weight, float(count), int(groups), bn.eps, momentum,
                         bool(pool))
