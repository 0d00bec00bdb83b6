"""Fused eval-mode backbone: Track4D.backbone() as a short sequence of hand-written gfx950 kernels.

Host side of include/rtk_fused.h.  What happens here (once per weight set, not per step):
  * eval-mode BatchNorm is folded into the preceding 1x1 conv (scale into the weights, shift into the
    bias); consecutive linear maps are composed (nn.Linear bottleneck followed by the next level's
    first conv; conv2 followed by the cls head's Linear(3,1));
  * the first layer of every grouped MLP is split by linearity into a per-POINT projection of the
    features (computed once per point instead of once per (centroid, neighbour) pair) and a 3-channel
    xyz-offset term evaluated per pair -- same function, 8-32x fewer multiply-adds in that layer;
  * weights are packed into the MFMA fragment order documented in csrc/fused_common.h.
Per step: geometry kernels (FPS / ball query / three-NN / kNN) + fused stages on point-major tensors.
All results agree with the module path / CPU oracle within fp32 rounding (tests: 1e-4 rel-to-scale).
"""
import ctypes

import torch

from . import _lib
from . import pointnet2_hip as _native

ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_SIGMOID = 0, 1, 2, 3


# ---- ctypes mirrors of the structs in include/rtk_fused.h -------------------------------------------
class _Src(ctypes.Structure):
    _fields_ = [("ptr", ctypes.c_void_p), ("pitch", ctypes.c_int), ("channels", ctypes.c_int), ("per_sample", ctypes.c_int)]


class _Layer(ctypes.Structure):
    _fields_ = [("w_packed", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("cin16", ctypes.c_int), ("cout16", ctypes.c_int),
                ("act", ctypes.c_int)]


class _Interp(ctypes.Structure):
    _fields_ = [("known_feats", ctypes.c_void_p), ("pitch", ctypes.c_int), ("channels", ctypes.c_int), ("m", ctypes.c_int),
                ("idx", ctypes.c_void_p), ("dist2", ctypes.c_void_p)]


_vp, _ci = ctypes.c_void_p, ctypes.c_int
_lib.SIGNATURES.update({
    "rtk_pointwise_mlp": [_ci, _ci, ctypes.POINTER(_Interp), _ci, ctypes.POINTER(_Src), _vp, _ci, ctypes.POINTER(_Layer), _vp,
                          _ci, _ci, _ci, _vp],
    "rtk_sa_scale": [_ci] * 4 + [_vp] * 4 + [_ci, _ci, _vp, _ci, ctypes.POINTER(_Layer), _vp, _ci, _ci, _vp],
    "rtk_cost_volume": [_ci] * 3 + [_vp] * 7 + [ctypes.POINTER(_Layer), ctypes.POINTER(_Layer), _vp, _ci, _vp],
    "rtk_patch_cost": [_ci] * 2 + [_vp] * 3 + [_ci, ctypes.POINTER(_Layer), _vp, _ci, _ci, _vp],
})


def _stream():
    return torch.cuda.current_stream().cuda_stream


def ceil16(c):
    return (c + 15) // 16 * 16


# ---- weight preparation -----------------------------------------------------------------------------

def fold_bn(w, bn_prefix, sd, eps=1e-5):
    """Conv2d(no bias) + eval BatchNorm -> (W', b'):  y = s*(W x) + (beta - s*mean),  s = gamma/sqrt(var+eps)."""
    g, b = sd[bn_prefix + ".weight"].double(), sd[bn_prefix + ".bias"].double()
    m, v = sd[bn_prefix + ".running_mean"].double(), sd[bn_prefix + ".running_var"].double()
    s = g / torch.sqrt(v + eps)
    w = w.double().reshape(w.shape[0], -1)
    return (w * s[:, None]), (b - s * m)


def pack_layer(w):
    """(Cout, Cin) -> fragment-major image [U][V][64 lanes][4]: packed[u][v][16g+i][r] = W[16v+i][16u+4g+r]
    (zero padded to multiples of 16).  One (u, v) fragment = the A operands of 4 consecutive MFMA k-steps."""
    cout, cin = w.shape
    V, U = ceil16(cout) // 16, ceil16(cin) // 16
    wp = torch.zeros(V * 16, U * 16, dtype=torch.float32, device=w.device)
    wp[:cout, :cin] = w.float()
    return wp.reshape(V, 16, U, 4, 4).permute(2, 0, 3, 1, 4).contiguous().reshape(-1)   # (U, V, g, i, r)


def pad_bias(b, cout):
    out = torch.zeros(ceil16(cout), dtype=torch.float32, device=b.device)
    out[:b.numel()] = b.float()
    return out


class Chain:
    """A chain of packed layers living in ONE contiguous device blob (the kernels stream it through LDS)."""

    def __init__(self, layers, device):
        """layers: list of (W (Cout,Cin) float64/32 tensor, bias (Cout,), act)."""
        packs, biases, meta = [], [], []
        for w, b, act in layers:
            cout, cin = w.shape
            packs.append(pack_layer(w.to(device)))
            biases.append(pad_bias(b.to(device), cout))
            meta.append((ceil16(cin) // 16, ceil16(cout) // 16, act))
        self.blob = torch.cat(packs).contiguous()
        self.bias = torch.cat(biases).contiguous()
        arr = (_Layer * len(layers))()
        woff = boff = 0
        for i, (u, v, act) in enumerate(meta):
            arr[i].w_packed = self.blob.data_ptr() + 4 * woff
            arr[i].bias = self.bias.data_ptr() + 4 * boff
            arr[i].cin16, arr[i].cout16, arr[i].act = u, v, act
            woff += u * v * 256
            boff += v * 16
        self.arr = arr
        self.n = len(layers)
        self.cout = layers[-1][0].shape[0]
        self.cout16 = meta[-1][1]


def pointwise_mlp(rows, rows_per_sample, srcs, chain, out, out_channels=None, sample_bias=None, interp=None,
                  channel_major=False):
    """srcs: list of (tensor2d (rows_or_samples, pitch), channels, per_sample).  out: (rows, pitch) point-major or
    (samples, C, n) channel-major.  interp: (known_feats (samples*m, pitch), channels, m, idx (rows,3) int32, dist2 (rows,3))."""
    arr = (_Src * max(len(srcs), 1))()
    for i, (t, ch, per) in enumerate(srcs):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.device.type == "cuda"
        arr[i].ptr, arr[i].pitch, arr[i].channels, arr[i].per_sample = t.data_ptr(), t.shape[-1], ch, int(per)
    ip = None
    if interp is not None:
        kf, ch, m, idx, d2 = interp
        it = _Interp(kf.data_ptr(), kf.shape[-1], ch, m, idx.data_ptr(), d2.data_ptr())
        ip = ctypes.pointer(it)
    oc = out_channels if out_channels is not None else chain.cout
    pitch = 0 if channel_major else out.shape[-1]
    _lib.call("rtk_pointwise_mlp", rows, rows_per_sample, ip, len(srcs), arr,
              sample_bias.data_ptr() if sample_bias is not None else None, chain.n, chain.arr, out.data_ptr(), pitch, oc,
              int(channel_major), _stream())
    return out
