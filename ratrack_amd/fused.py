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
import os

import torch

from . import _lib
from . import pointnet2_hip as _native

ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_SIGMOID = 0, 1, 2, 3


# ---- ctypes mirrors of the structs in include/rtk_fused.h -------------------------------------------
class _Src(ctypes.Structure):
    _fields_ = [("ptr", ctypes.c_void_p), ("pitch", ctypes.c_int), ("channels", ctypes.c_int), ("per_sample", ctypes.c_int)]


class _Layer(ctypes.Structure):
    _fields_ = [("w_packed", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("cin16", ctypes.c_int), ("cout16", ctypes.c_int),
                ("act", ctypes.c_int), ("inv_scale", ctypes.c_float)]


class _Interp(ctypes.Structure):
    _fields_ = [("known_feats", ctypes.c_void_p), ("pitch", ctypes.c_int), ("channels", ctypes.c_int), ("m", ctypes.c_int),
                ("idx", ctypes.c_void_p), ("dist2", ctypes.c_void_p), ("nuniq", ctypes.c_void_p)]


_vp, _ci = ctypes.c_void_p, ctypes.c_int
_lib.SIGNATURES.update({
    "rtk_pointwise_mlp": [_ci, _ci, ctypes.POINTER(_Interp), _ci, ctypes.POINTER(_Src), _vp, _ci, ctypes.POINTER(_Layer), _vp,
                          _ci, _ci, _ci, _vp, _vp, _vp],
    "rtk_sa_scale": [_ci] * 4 + [_vp] * 4 + [_ci, _ci, _vp, _ci, ctypes.POINTER(_Layer), _vp, _ci, _ci, _vp, _vp, _vp],
    "rtk_cost_volume": [_ci] * 3 + [_vp] * 6 + [ctypes.POINTER(_Layer), ctypes.POINTER(_Layer), _vp, _ci, _vp],
    "rtk_patch_cost": [_ci] * 2 + [_vp] * 3 + [_ci, ctypes.POINTER(_Layer), _vp, _ci, _ci, _vp],
    "rtk_pack_split_layer": [_ci, _ci, _vp, _ci, _vp, _vp, _vp],
    "rtk_cost_volume_split": [_ci] * 3 + [_vp] * 10 + [ctypes.POINTER(_Layer), _vp, _ci, _vp],
    "rtk_cost_volume_split_shared": [_ci] * 3 + [_vp] * 10 + [ctypes.POINTER(_Layer), _vp, _ci, _ci, _vp],
    "rtk_split_mlp2": [_ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "rtk_sa_scale_split": [_ci] * 4 + [_vp] * 4 + [_ci, _ci, _vp, _vp, _vp, _vp, _vp, _ci, _ci, _vp, _vp, _vp],
    "rtk_prepare_inputs": [_ci] * 2 + [_vp] * 6 + [_vp],
    "rtk_fps_centroids": [_ci] * 3 + [_vp] * 8 + [_vp],
    "rtk_knn_point_masked": [_ci] * 4 + [_vp] * 4 + [_vp],
    "rtk_fps_relevel": [_ci] * 3 + [_vp] * 9 + [_ci, _vp, _vp],
    "rtk_gru_step": [_ci] * 3 + [_vp] * 8 + [_vp],
    "rtk_to_channel_major": [_ci] * 3 + [_vp, _ci, _ci, _vp, _ci, _ci, _vp],
    "rtk_ball_query_pair": [_ci] * 3 + [ctypes.c_float, _ci, ctypes.c_float, _ci] + [_vp] * 5 + [_vp],
    "rtk_geometry_front": [_ci] * 4 + [_vp, _vp, _ci] + [_vp] * 13 + [_vp, _vp, _ci, _vp],
    "rtk_geometry_tables": [_ci] * 3 + [_vp] * 3 + [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(_ci), ctypes.POINTER(_vp), ctypes.POINTER(_vp),
                            ctypes.POINTER(_vp), _vp],
    "rtk_three_nn_masked": [_ci] * 3 + [_vp] * 6 + [_vp],
    "rtk_to_channel_major_multi": [_ci] * 3 + [_vp, _vp],
    "rtk_log_sinkhorn": [_ci, _ci, _vp, ctypes.c_float, _ci, _vp, _vp],
    "rtk_dbscan": [_ci, _vp, _ci, _vp, _vp, ctypes.c_float, ctypes.c_double, _ci, _vp, _vp],
})


class _GtJob(ctypes.Structure):
    _fields_ = [("wt", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("out", ctypes.c_void_p), ("cout", ctypes.c_int), ("s0", ctypes.c_int),
                ("count", ctypes.c_int), ("out_pitch", ctypes.c_int)]


class _CopyJob(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("bytes", ctypes.c_long)]


_lib.SIGNATURES.update({
    "rtk_gru_step_head": [_ci] * 3 + [_vp] * 11 + [_ci, _vp],
    "rtk_global_terms": [_ci, _ci, _vp, _ci, ctypes.POINTER(_GtJob), _vp, _ci, _ci, _vp],
    "rtk_copy_multi": [_ci, ctypes.POINTER(_CopyJob), _vp],
})


def copy_multi(pairs):
    """[(dst, src), ...] contiguous same-shape tensors: one launch (rtk_copy_multi) instead of a framework multi-tensor copy."""
    for i in range(0, len(pairs), 8):
        part = pairs[i:i + 8]
        jobs = (_CopyJob * len(part))()
        for j, (d, s_) in enumerate(part):
            assert d.is_contiguous() and s_.is_contiguous() and d.shape == s_.shape and d.dtype == s_.dtype and d.device == s_.device
            jobs[j].src, jobs[j].dst, jobs[j].bytes = s_.data_ptr(), d.data_ptr(), d.numel() * d.element_size()
        _lib.call("rtk_copy_multi", len(part), jobs, _stream())


class _LayoutJob(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("channels", ctypes.c_int), ("src_pitch", ctypes.c_int),
                ("per_sample", ctypes.c_int), ("dst_channels", ctypes.c_int), ("dst_channel_offset", ctypes.c_int)]


def _stream():
    return torch.cuda.current_stream().cuda_stream


def ceil16(c):
    return (c + 15) // 16 * 16


# ---- work counter ------------------------------------------------------------------------------------------------
# The multiply-adds this design EXECUTES differ from the reference formulation's (per-point layer-1 projections instead of
# per-(centroid, neighbour) ones, duplicate centroids skipped, composed linear maps), and part of the count is data
# dependent (exhausted-cloud counters).  trace_work() records every fused launch of one backbone() call with its shapes;
# executed_macs() turns the records into multiply-adds (true channel counts, not the 16-padded MFMA tiles).
_TRACE = None


class trace_work:
    def __enter__(self):
        global _TRACE
        _TRACE = self.records = []
        return self

    def __exit__(self, *a):
        global _TRACE
        _TRACE = None
        return False

    def executed_macs(self):
        """-> (total multiply-adds, {kernel family: multiply-adds}).  Synchronises (reads the duplicate-row counters)."""
        per = {}
        for kind, rows, macs_per_row in self.records:
            r = float(rows.sum().item()) if torch.is_tensor(rows) else float(rows)
            per[kind] = per.get(kind, 0.0) + r * macs_per_row
        return sum(per.values()), per


def _live_rows(nuniq, rows_per_sample, samples):
    """Rows a kernel computes when rows >= nuniq[b] of every sample are skipped (device tensor: no sync here)."""
    if nuniq is None:
        return samples * rows_per_sample
    return torch.clamp(nuniq[:samples].to(torch.int64), max=rows_per_sample)


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
    if cout % 16 == 0 and cin % 16 == 0:      # no padding: one permuting copy (the training path re-packs every step)
        wp = w.float()
    else:
        wp = torch.zeros(V * 16, U * 16, dtype=torch.float32, device=w.device)
        wp[:cout, :cin] = w.float()
    return wp.reshape(V, 16, U, 4, 4).permute(2, 0, 3, 1, 4).contiguous().reshape(-1)   # (U, V, g, i, r)


SPLIT_IMAGE_256 = 2 * 256 * 256      # int16 elements of the split image of a 256 x 256 layer (two fp16 pieces)


def split3_bf16(w):
    """The three bf16 pieces of an fp32 tensor (truncation splits: p0 + p1 + p2 == w exactly), as the upper 16 bits of each
    piece (int16), stacked on a new first axis.  (Rounds 2-4's split; the 16-position per-point chains and the training kernels'
    position contractions still take it: csrc/fused_common.h.)"""
    x = w.float().contiguous()
    out = []
    for _ in range(3):
        top = (x.view(torch.int32) & -65536)
        out.append((top >> 16).to(torch.int16))
        x = x - top.view(torch.float32)
    return torch.stack(out)


def pow2_scale(amax):
    """The exact power of two s with amax * s in [2^14, 2^15) and its inverse, as csrc/fused_common.h lane_scale_of computes them from
    the exponent field of amax (clamped: s <= 2^126; amax = 0 -> 2^126)."""
    e = (torch.as_tensor(amax, dtype=torch.float32).reshape(1).view(torch.int32).item() >> 23) & 0xff
    sf = min(268 - e, 253)
    return 2.0 ** (sf - 127), 2.0 ** (127 - sf)


def split2_f16(w, scale):
    """The two fp16 pieces of w * scale (round to nearest: h = fp16(w s), l = fp16(w s - h)) as int16 bit patterns, stacked on a new
    first axis.  csrc/split_mfma.h."""
    x = w.float().contiguous() * scale                                    # exact: a power of two
    h = x.half()
    l = (x - h.float()).half()
    return torch.stack([h.view(torch.int16), l.view(torch.int16)])


def pack_layer_split(w):
    """(Cout, Cin) fp32, both multiples of 32 -> (split image of csrc/split_mfma.h as int16, inverse weight scale):
    frag[s][v][p][lane = 32 hh + i][t] = piece_p(2^k W[32 v + i][32 (s / 2) + 16 (s % 2) + 8 (t / 4) + 4 hh + t % 4]).
    (The product path packs on the device, rtk_pack_split_layer; this is the host restatement the tests compare it with.)"""
    cout, cin = w.shape
    assert cout % 32 == 0 and cin % 32 == 0
    scale, inv = pow2_scale(w.float().abs().max())
    p = split2_f16(w, scale)                                             # (2, cout, cin)
    p = p.reshape(2, cout // 32, 32, cin // 32, 2, 2, 2, 4)               # (p, v, i, a, e, d, hh, r): c = 32 a + 16 e + 8 d + 4 hh + r
    return p.permute(3, 4, 1, 0, 6, 2, 5, 7).contiguous().reshape(-1), inv     # (a, e, v, p, hh, i, d, r): s = 2 a + e, t = 4 d + r


def pack_split_device(w, transposed=False):
    """rtk_pack_split_layer on a (cout, cin) fp32 CUDA tensor (transposed: w is stored (cin, cout)) -> (image int16, inv_scale (1,) fp32)."""
    cout, cin = (w.shape[1], w.shape[0]) if transposed else w.shape
    image = torch.empty(2 * cout * cin, dtype=torch.int16, device=w.device)
    inv = torch.empty(1, dtype=torch.float32, device=w.device)
    _lib.call("rtk_pack_split_layer", cout, cin, w.data_ptr(), 1 if transposed else 0, image.data_ptr(), inv.data_ptr(), _stream())
    return image, inv


def pack_layer_split16(w):
    """(Cout, Cin) -> (16-position split image (csrc/fused_common.h) as int16, inverse weight scale): one fragment per (pair of
    16-channel input blocks, 16-channel output block, piece): frag[up][v][p][lane = 16 g + i][t] =
    piece_p(2^k W[16 v + i][16 (2 up + t // 4) + 4 g + t % 4]), zero beyond (Cout, Cin)."""
    cout, cin = w.shape
    V, U2 = ceil16(cout) // 16, (ceil16(cin) // 16 + 1) // 2
    wp = torch.zeros(V * 16, U2 * 32, dtype=torch.float32, device=w.device)
    wp[:cout, :cin] = w.float()
    scale, inv = pow2_scale(wp.abs().max())
    p = split2_f16(wp, scale).reshape(2, V, 16, U2, 2, 4, 4)               # (p, v, i, up, d, g, r): c = 32 up + 16 d + 4 g + r
    return p.permute(3, 1, 0, 5, 2, 4, 6).contiguous().reshape(-1), inv    # (up, v, p, g, i, d, r): lane = 16 g + i, t = 4 d + r


# The product path runs every wide layer on the split-bf16 matrix path (csrc/split_mfma.h).  The fp32-input MFMA kernels stay compiled as
# the second implementation the tests compare against: tests flip these module attributes / FusedBackbone(split=False).
PW_SPLIT = True
LAYER_SPLIT = 0x100      # RTK_LAYER_SPLIT (include/rtk_fused.h)


def pad_bias(b, cout):
    out = torch.zeros(ceil16(cout), dtype=torch.float32, device=b.device)
    out[:b.numel()] = b.float()
    return out


class Chain:
    """A chain of packed layers living in ONE contiguous device blob (the kernels stream it through LDS)."""

    def split_arr(self):
        """The same chain as split images (rtk_pointwise_mlp with RTK_LAYER_SPLIT), built on first use."""
        if self._split is None:
            packed = [pack_layer_split16(w.to(self._device)) for w, _ in self._layers]
            blob = torch.cat([im for im, _ in packed]).contiguous()
            arr = (_Layer * len(self._layers))()
            off = boff = 0
            for i, (w, act) in enumerate(self._layers):
                cout, cin = w.shape
                u, v = ceil16(cin) // 16, ceil16(cout) // 16
                arr[i].w_packed = blob.data_ptr() + 2 * off
                arr[i].bias = self.bias.data_ptr() + 4 * boff
                arr[i].cin16, arr[i].cout16, arr[i].act, arr[i].inv_scale = u, v, act | LAYER_SPLIT, packed[i][1]
                off += ((u + 1) // 2) * v * 2 * 512      # int16 elements per fragment: 64 lanes x 8
                boff += v * 16
            self._split = (arr, blob)
        return self._split[0]

    def __init__(self, layers, device):
        """layers: list of (W (Cout,Cin) float64/32 tensor, bias (Cout,), act)."""
        packs, biases, meta = [], [], []
        self.dims = [tuple(w.shape) for w, _, _ in layers]          # true (Cout, Cin) per layer, for the work counter
        for w, b, act in layers:
            cout, cin = w.shape
            packs.append(pack_layer(w.to(device)))
            biases.append(pad_bias(b.to(device), cout))
            meta.append((ceil16(cin) // 16, ceil16(cout) // 16, act))
        self.blob = torch.cat(packs).contiguous()
        self.bias = torch.cat(biases).contiguous()
        self._layers, self._device, self._split = [(w, act) for w, _, act in layers], device, None
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


def _colptr(t, col=0):
    """(data pointer, pitch) of a 2-D row-major view starting at column `col`."""
    assert t.dim() == 2 and t.stride(1) == 1 and t.dtype == torch.float32
    return t.data_ptr() + 4 * col, t.stride(0)


def pointwise(rows, rows_per_sample, srcs, chain, out, out_channels=None, sample_bias=None, interp=None, channel_major=False,
              row_nuniq=None, colmax=None):
    """srcs: list of (2-D tensor or column-sliced view, channels, per_sample).  out: (rows, pitch) point-major (may be a
    column-sliced view) or (samples, C, n) channel-major.  interp: (known_feats (samples*m, pitch), channels, m,
    idx (rows,3) int32, dist2 (rows,3)[, nuniq (samples) int32]).  row_nuniq: per-sample count of non-duplicate rows."""
    arr = (_Src * max(len(srcs), 1))()
    for i, (t, ch, per) in enumerate(srcs):
        ptr, pitch = _colptr(t)
        arr[i].ptr, arr[i].pitch, arr[i].channels, arr[i].per_sample = ptr, pitch, ch, int(per)
    ip = None
    if interp is not None:
        kf, ch, m, idx, d2 = interp[:5]
        nu = interp[5] if len(interp) > 5 else None
        ptr, pitch = _colptr(kf)
        ip = ctypes.pointer(_Interp(ptr, pitch, ch, m, idx.data_ptr(), d2.data_ptr(), nu.data_ptr() if nu is not None else None))
    oc = out_channels if out_channels is not None else chain.cout
    if _TRACE is not None:
        macs = sum(co * ci for co, ci in chain.dims) + (3 * interp[1] if interp is not None else 0)
        _TRACE.append(("pointwise", _live_rows(row_nuniq, rows_per_sample, rows // rows_per_sample), macs))
    if channel_major:
        optr, opitch = out.data_ptr(), 0
    else:
        optr, opitch = _colptr(out)
    _lib.call("rtk_pointwise_mlp", rows, rows_per_sample, ip, len(srcs), arr,
              sample_bias.data_ptr() if sample_bias is not None else None, chain.n, chain.split_arr() if PW_SPLIT else chain.arr, optr, opitch, oc,
              int(channel_major), row_nuniq.data_ptr() if row_nuniq is not None else None,
              colmax.data_ptr() if colmax is not None else None, _stream())
    return out


pointwise_mlp = pointwise


def offset_image(w4, device):
    """[Wx | b] (Cout, 4) -> A-operand image of the single k-step offset layer: img[v][16g+i] = W4[16v+i][g]."""
    cout = w4.shape[0]
    V = ceil16(cout) // 16
    wp = torch.zeros(V * 16, 4, dtype=torch.float32, device=device)
    wp[:cout] = w4.float().to(device)
    return wp.reshape(V, 16, 4).permute(0, 2, 1).contiguous().reshape(-1)


class _WeightNet:
    """WeightNet(3 -> 8 -> 8 -> 256), bn=False (model_utils.py:359-390) in kernel form."""

    def __init__(self, sd, prefix, device):
        g = lambda k: sd[prefix + k].double()
        wa, ba = g(".mlp_convs.0.weight").reshape(8, 3), g(".mlp_convs.0.bias")
        wb, bb = g(".mlp_convs.1.weight").reshape(8, 8), g(".mlp_convs.1.bias")
        wc, bc = g(".mlp_convs.2.weight").reshape(-1, 8), g(".mlp_convs.2.bias")
        self.wa = offset_image(torch.cat([wa, ba[:, None]], 1), device)
        self.wb, self.bb = pack_layer(wb.to(device)), pad_bias(bb.to(device), 8)
        self.wc, self.bc = pack_layer(wc.to(device)), pad_bias(bc.to(device), wc.shape[0])
        arr = (_Layer * 3)()
        arr[0].w_packed, arr[0].cin16, arr[0].cout16 = self.wa.data_ptr(), 1, 1
        arr[1].w_packed, arr[1].bias, arr[1].cin16, arr[1].cout16 = self.wb.data_ptr(), self.bb.data_ptr(), 1, 1
        arr[2].w_packed, arr[2].bias, arr[2].cin16, arr[2].cout16 = self.wc.data_ptr(), self.bc.data_ptr(), 1, ceil16(wc.shape[0]) // 16
        self.arr = arr


SA_SPLIT = True


class _SAScale:
    """One MSG scale: offset image + packed layers 2(,3) + the per-point projection matrix of layer 1."""

    def __init__(self, sd, prefix, nsample, radius, device):
        self.nsample, self.radius = nsample, float(radius)
        ws = []
        i = 0
        while (prefix + ".layer%d.conv.weight" % i) in sd:
            ws.append(fold_bn(sd[prefix + ".layer%d.conv.weight" % i], prefix + ".layer%d.bn.bn" % i, sd))
            i += 1
        w1, b1 = ws[0]
        self.c1 = w1.shape[0]
        self.wf = w1[:, 3:]                                         # (C1, Cf) acts on the features -> per-point projection
        self.w1img = offset_image(torch.cat([w1[:, :3], b1[:, None]], 1), device)
        self.chain = Chain([(w, b, ACT_RELU) for w, b in ws[1:]], device)
        self.cout = ws[-1][0].shape[0]
        # the wide two-layer scales also as a split image (csrc/split_mfma.h): rtk_sa_scale_split
        self.split_image = None
        if SA_SPLIT and len(ws) == 2 and self.cout == 64 and self.c1 in (32, 64) and nsample in (16, 32) and torch.device(device).type == "cuda":
            w2 = ws[1][0].float().to(device).contiguous()
            self.split_bias = ws[1][1].float().to(device).contiguous()
            self.split_image, self.split_scale = pack_split_device(w2)
            torch.cuda.current_stream().synchronize()


class _PNHeadWeights:
    """Folded / composed weights of one PNHead (model_utils.py:393-424)."""

    RADII = [[2, 4], [4, 8], [8, 16]]
    NSAMPLES = [[4, 8], [8, 16], [16, 32]]

    def __init__(self, sd, prefix, device):
        d = lambda k: sd[prefix + k].double()
        self.scales = [[_SAScale(sd, "%ssa%d.mlps.%d" % (prefix, l + 1, s), self.NSAMPLES[l][s], self.RADII[l][s], device)
                        for s in range(2)] for l in range(3)]
        # projection of the raw level-0 features for sa1 (both scales side by side); the caller supplies the split
        self.wq1 = torch.cat([self.scales[0][0].wf, self.scales[0][1].wf], 0)            # (32, Cf)
        # level transitions: nn.Linear bottleneck composed with the next level's layer-1 feature projection
        self.trans = []
        for l in (1, 2):
            wl, bl = d("linear%d.weight" % l), d("linear%d.bias" % l)
            rows, bias = [wl], [bl]
            for s in range(2):
                wf = self.scales[l][s].wf
                rows.append(wf @ wl)
                bias.append(wf @ bl)
            self.trans.append(Chain([(torch.cat(rows, 0), torch.cat(bias, 0), ACT_NONE)], device))
        self.lin3 = Chain([(d("linear3.weight"), d("linear3.bias"), ACT_NONE)], device)
        self.fp = {}
        for name in ("fp3", "fp2", "fp1"):
            w, b = fold_bn(sd[prefix + name + ".mlp.layer0.conv.weight"], prefix + name + ".mlp.layer0.bn.bn", sd)
            self.fp[name] = Chain([(w, b, ACT_RELU)], device)


FUSED_GEOMETRY = True          # the geometry of a batch in two launches (rtk_geometry_front / rtk_geometry_tables); False (tests): the eleven
                               # launches of the separate entry points they replace -- the same tables bit for bit
GEOMETRY_POISON = None         # tests: an int32 pattern for the (otherwise uninitialised) index workspace of the two-launch eval geometry
CHECK_FPS_RELEVEL = False      # debug: compare every re-levelling launch with the full selection (synchronises; tests set it)


def check_fps_relevel(xyz1, idx, new_xyz, nuniq):
    """Debug / test helper: levels 2.. as produced by rtk_fps_relevel (idx (L,S_,npoint), new_xyz (L,S_,npoint,3),
    nuniq (L,S_)) against the full selection kernel run level after level.  Synchronises."""
    S_, npoint, _ = xyz1.shape
    src = xyz1
    for l in range(idx.shape[0]):
        i = torch.empty(S_, npoint, dtype=torch.int32, device=src.device)
        out = torch.empty(S_, npoint, 3, dtype=torch.float32, device=src.device)
        cnt = torch.empty(S_, dtype=torch.int32, device=src.device)
        _lib.call("rtk_fps_centroids", S_, npoint, npoint, src.data_ptr(), i.data_ptr(), out.data_ptr(), cnt.data_ptr(), None, None, None, None,
                  _stream())
        assert torch.equal(i, idx[l].view(S_, npoint)), "fps_relevel: level %d indices differ from the full selection" % (l + 2)
        assert torch.equal(out, new_xyz[l].view(S_, npoint, 3)), "fps_relevel: level %d centroids differ" % (l + 2)
        assert torch.equal(cnt, nuniq[l].view(-1)), "fps_relevel: level %d exhausted-cloud counters differ" % (l + 2)
        src = out


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class Geometry:
    """FPS centroids, ball-query indices and three-NN tables of one batch of clouds (feature independent,
    so the decoder's PNHead over pc1 reuses the encoder's)."""

    def __init__(self, xyz, npoint, side=None, knn_frames=0, finite=False, n_valid=None, level_hook=None, tail_hook=None, zeros=None,
                 prepare=None, q1=None):
        """xyz (S_,n,3).  prepare = (pc1, pc2, feature1, feature2, raw): xyz and raw (S_*n, 4) are OUTPUTS -- the API's channel-major
        tensors of the two frames are converted on the way (rtk_prepare_inputs, inside rtk_geometry_front when the geometry is fused).
        q1 = (w (C, 2) fp32, out (S_*n, C)) with prepare: out = w . (the two raw features of every point), written by the same launch when
        the geometry is fused (self.q1_done says whether it was).  With `side` (a torch.cuda.Stream) every geometry kernel is enqueued on that stream, forked from
        the current one, and consumers call wait(stage) -- the feature kernels overlap the latency-bound FPS chain.
        knn_frames = B > 0: also the two kNN tables of the cost volume, frame 1 = xyz[:B], frame 2 = xyz[B:].
        finite: zero-fill the three-NN distances of the skipped (duplicate) rows instead of leaving them unwritten
        (the training path computes -- and ignores -- those rows, so they must hold finite numbers).
        n_valid (S_,) int32: padded batch (ratrack_amd/vod_gt.pad_frame_pairs) -- cloud s consists of its first n_valid[s]
        points, the rest are copies of its point 0; FPS applies the unpadded cloud's tie rule and the kNN tables only
        hold valid candidates, everything else is exact through the duplicate-of-point-0 property.
        level_hook(geo, lvl) / tail_hook(geo): called (on the geometry stream) right after level lvl's ball query, before its
        event is recorded / after the three-NN tables -- the training path enqueues its per-level tables there.
        zeros(n, dtype, device): allocator of the zero-initialised workspaces (default torch.zeros; the training path passes its
        step arena, whose one fill then covers these too)."""
        zeros_given = zeros
        zeros = zeros or (lambda n_, dtype, device: torch.zeros(n_, dtype=dtype, device=device))
        S_, n, _ = xyz.shape
        if n_valid is not None:
            assert n_valid.shape == (S_,) and n_valid.dtype == torch.int32 and n_valid.is_contiguous() and n <= 2048
        nv = n_valid.data_ptr() if n_valid is not None else None
        dev = xyz.device
        self.n, self.samples, self.npoint = n, S_, npoint
        self.xyz = [xyz]
        self.nuniq = []
        self.events = {}
        # ---- all workspaces are allocated on the CURRENT stream, before the fork ------------------------------
        # one int32 workspace for the 3 FPS index rows, the 3 exhausted-cloud counters, the 6 ball-query tables
        # (zero-initialised once: the caller-zero-inits contract of ball_query, lib/pointnet2_utils.py:246)
        # and the 3 three-NN index tables
        ns_all = [ns for row in _PNHeadWeights.NSAMPLES for ns in row]
        nn_rows = [npoint, npoint, n]
        sizes = [S_ * npoint] * 3 + [S_] * 7 + [S_ * npoint * ns for ns in ns_all] + [S_ * r * 3 for r in nn_rows]
        # The two-launch geometry writes every entry a consumer reads (the skipped rows of duplicate centroids are read by nothing:
        # every consumer takes the duplicate-row counters), so the eval path's workspace is not filled (a 9 MB fill per batch until
        # round 6); GEOMETRY_POISON (tests) fills it with an out-of-range index instead, which a stray read would turn into a fault.
        fused_geo = (FUSED_GEOMETRY and n <= 2048 and npoint <= 512 and max(n, npoint) * 12 <= 64 * 1024 and
                     (not knn_frames or (n >= 16 and S_ == 2 * knn_frames)) and (prepare is None or S_ % 2 == 0))
        if fused_geo and zeros_given is None and not finite:
            ws = torch.empty(sum(sizes), dtype=torch.int32, device=dev)
            if GEOMETRY_POISON is not None:
                ws.fill_(GEOMETRY_POISON)
        else:
            ws = zeros(sum(sizes), torch.int32, dev)
        parts = list(torch.split(ws, sizes))
        fps_idx, cnt, tie, tie23, first_tie, ball, nn_idx = parts[0:3], parts[3:6], parts[6], parts[7:9], parts[9], parts[10:16], parts[16:19]
        # level-1 min-distance state at the first tied round, by point index (written for tied clouds only)
        snap = torch.empty(S_ * n, dtype=torch.float32, device=dev) if n <= 2048 else None
        self.fps_idx = [t.view(S_, npoint) for t in fps_idx]
        self.tie = tie
        xyz_all = torch.empty(3, S_, npoint, 3, dtype=torch.float32, device=dev)
        new_xyz = [xyz_all[l] for l in range(3)]
        d2_all = zeros(sum(nn_rows) * S_ * 3, torch.float32, dev) if finite else torch.empty(sum(nn_rows) * S_ * 3, dtype=torch.float32, device=dev)
        d2_parts = torch.split(d2_all, [S_ * r * 3 for r in nn_rows])
        B = knn_frames
        self.knn = [torch.empty(B, n, 16, dtype=torch.int64, device=dev) for _ in range(2)] if B else None
        big = n > 2048
        temp = torch.full((S_, n), 1e10, dtype=torch.float32, device=dev) if big else None
        # every operand a geometry kernel reads or writes through a raw pointer must live as long as this object -- the kernels run
        # asynchronously (on the side stream), and a buffer dropped at the end of __init__ goes back to the allocator's main-stream
        # pool, whose next tenant is then written WHILE the kernel still uses the bytes (round 3: a tie snapshot buffer freed this way
        # put one real frame pair in three off by 2e-3, only with warm allocator pools)
        self._scratch = (snap, temp, n_valid, xyz)

        self.fused_geometry = fused_geo
        self.q1_done = False
        q1w = q1 is not None and prepare is not None
        if prepare is not None and not fused_geo:      # on the caller's stream, before the fork: the feature kernels read raw there
            pc1, pc2, f1, f2, raw = prepare
            _lib.call("rtk_prepare_inputs", S_ // 2, n, pc1.data_ptr(), pc2.data_ptr(), f1.data_ptr(), f2.data_ptr(), xyz.data_ptr(), raw.data_ptr(),
                      _stream())
            self._scratch = self._scratch + (pc1, pc2, f1, f2, raw)
        main = torch.cuda.current_stream()
        if side is not None:
            side.wait_stream(main)
        ctx = torch.cuda.stream(side) if side is not None else _NullCtx()
        self.ball = [[ball[lvl * 2 + s].view(S_, npoint, _PNHeadWeights.NSAMPLES[lvl][s]) for s in range(2)] for lvl in range(3)]
        if fused_geo:
            with ctx:
                for lvl in range(3):
                    self.xyz.append(new_xyz[lvl])
                    self.nuniq.append(cnt[lvl])
                b1 = S_ // 2 if (B or prepare is not None) else S_
                if prepare is not None:
                    pc1, pc2, f1, f2, raw = prepare
                    fr = (pc1.data_ptr(), pc2.data_ptr(), 1, f1.data_ptr(), f2.data_ptr(), xyz.data_ptr(), raw.data_ptr())
                    self._scratch = self._scratch + (pc1, pc2, f1, f2, raw)
                else:
                    fr = (xyz.data_ptr(), xyz[b1:].data_ptr() if b1 < S_ else None, 0, None, None, None, None)
                _lib.call("rtk_geometry_front", b1, S_, n, npoint, *fr, fps_idx[0].data_ptr(), xyz_all.data_ptr(), cnt[0].data_ptr(),
                          tie.data_ptr(), first_tie.data_ptr(), snap.data_ptr(), nv, self.knn[0].data_ptr() if B else None,
                          self.knn[1].data_ptr() if B else None, q1[0].data_ptr() if q1w else None, q1[1].data_ptr() if q1w else None,
                          q1[1].shape[1] if q1w else 0, _stream())
                self.q1_done = q1w
                self._record("front", side)        # xyz / raw (prepare), the kNN tables and the three levels of centroids
                if CHECK_FPS_RELEVEL and not torch.cuda.is_current_stream_capturing():
                    check_fps_relevel(new_xyz[0], torch.stack(self.fps_idx[1:]), xyz_all[1:], torch.stack(list(cnt[1:3])))
                radii = (ctypes.c_float * 6)(*[float(r) for row in _PNHeadWeights.RADII for r in row])
                nsam = (_ci * 6)(*ns_all)
                balls = (_vp * 6)(*[t.data_ptr() for t in ball])
                self.nn = {}
                for i, (name, (u, k)) in enumerate({"fp3": (2, 3), "fp2": (1, 2), "fp1": (0, 1)}.items()):
                    self.nn[name] = (d2_parts[i].view(S_, nn_rows[i], 3), nn_idx[i].view(S_, nn_rows[i], 3), self.xyz[k].shape[1])
                nni = (_vp * 3)(*[self.nn[k][1].data_ptr() for k in ("fp3", "fp2", "fp1")])
                nnd = (_vp * 3)(*[self.nn[k][0].data_ptr() for k in ("fp3", "fp2", "fp1")])
                _lib.call("rtk_geometry_tables", S_, n, npoint, xyz.data_ptr(), xyz_all.data_ptr(), cnt[0].data_ptr(), radii, nsam, balls, nni, nnd,
                          _stream())
                for lvl in range(3):
                    if level_hook is not None:
                        level_hook(self, lvl)
                    self._record(lvl, side)
                self._record("nn", side)
                if B:
                    self.events["knn"] = self.events.get("front")
                if tail_hook is not None:
                    tail_hook(self)
            return
        with ctx:
            # ---- level 1: the only full furthest-point selection on the common path ---------------------------
            if not big:
                _lib.call("rtk_fps_centroids", S_, n, npoint, xyz.data_ptr(), fps_idx[0].data_ptr(), new_xyz[0].data_ptr(),
                          cnt[0].data_ptr(), tie.data_ptr(), nv, snap.data_ptr(), first_tie.data_ptr(), _stream())
            else:   # large clouds: generic FPS + gather, no exhausted-cloud / tie information
                _native.furthest_point_sampling_wrapper(S_, n, npoint, xyz, temp, self.fps_idx[0])
                new_xyz[0].copy_(torch.gather(xyz, 1, self.fps_idx[0].long().unsqueeze(-1).expand(-1, -1, 3)))
                cnt[0].fill_(npoint)
                tie.fill_(npoint)       # unknown: levels 2, 3 run the full selection (no round can settle)

            def relevel():
                # ---- levels 2, 3: FPS of npoint out of the previous level's npoint centroids (model_utils.py:415-417).
                # One launch per level, decided per cloud on the device (no host sync): a cloud whose previous level had no tie is
                # provably the identity on the coordinates and is copied, a tied cloud runs the selection -- resumed at level 1's
                # first tied round, stopped once its picked set is a prefix again (rtk_fps_relevel)
                resume = (fps_idx[0].data_ptr(), snap.data_ptr(), n, first_tie.data_ptr()) if not big else (None, None, 0, None)
                _lib.call("rtk_fps_relevel", S_, npoint, 2, new_xyz[0].data_ptr(), cnt[0].data_ptr(), tie.data_ptr(),
                          fps_idx[1].data_ptr(), new_xyz[1].data_ptr(), cnt[1].data_ptr(), tie23[0].data_ptr(), *resume, _stream())
                if CHECK_FPS_RELEVEL and not torch.cuda.is_current_stream_capturing():      # the check synchronises
                    check_fps_relevel(new_xyz[0], torch.stack(self.fps_idx[1:]), xyz_all[1:], torch.stack(list(cnt[1:3])))
            for lvl in range(3):
                self.xyz.append(new_xyz[lvl])
                self.nuniq.append(cnt[lvl])
            self.ball = []
            for lvl in range(3):
                row = []
                for s in range(2):
                    ns, r = _PNHeadWeights.NSAMPLES[lvl][s], _PNHeadWeights.RADII[lvl][s]
                    bidx = ball[lvl * 2 + s].view(S_, npoint, ns)
                    row.append(bidx)
                self.ball.append(row)
            # order on the side stream = order of first use: level-l tables right after level-l centroids
            for lvl in range(3):
                (r1, r2), (n1, n2) = _PNHeadWeights.RADII[lvl], _PNHeadWeights.NSAMPLES[lvl]
                nsrc = self.xyz[lvl].shape[1]
                if nsrc * 12 <= 64 * 1024:      # both scales in one scan, duplicate centroids skipped
                    _lib.call("rtk_ball_query_pair", S_, nsrc, npoint, float(r1), n1, float(r2), n2, self.xyz[lvl + 1].data_ptr(),
                              self.xyz[lvl].data_ptr(), self.ball[lvl][0].data_ptr(), self.ball[lvl][1].data_ptr(),
                              self.nuniq[lvl].data_ptr(), _stream())
                else:
                    for s in range(2):
                        _native.ball_query_wrapper(S_, nsrc, npoint, float(_PNHeadWeights.RADII[lvl][s]), _PNHeadWeights.NSAMPLES[lvl][s],
                                                   self.xyz[lvl + 1], self.xyz[lvl], self.ball[lvl][s])
                if level_hook is not None:
                    level_hook(self, lvl)
                self._record(lvl, side)
                if lvl == 0:
                    relevel()
            self.nn = {}
            for i, (name, (u, k)) in enumerate({"fp3": (2, 3), "fp2": (1, 2), "fp1": (0, 1)}.items()):
                nu, m = self.xyz[u].shape[1], self.xyz[k].shape[1]
                d2 = d2_parts[i].view(S_, nu, 3)
                idx = nn_idx[i].view(S_, nu, 3)
                # unknown rows of fp3 / fp2 are level centroids: duplicates are never read downstream
                mask = self.nuniq[u - 1].data_ptr() if u > 0 else None
                _lib.call("rtk_three_nn_masked", S_, nu, m, self.xyz[u].data_ptr(), self.xyz[k].data_ptr(), d2.data_ptr(), idx.data_ptr(),
                          mask, self.nuniq[k - 1].data_ptr(), _stream())      # known level k: duplicate centroids beyond nuniq
                self.nn[name] = (d2, idx, m)
            self._record("nn", side)
            if B:
                x1, x2 = xyz[:B], xyz[B:]
                if n_valid is None:
                    _native.knn_point_wrapper(B, n, n, 16, x1, x2, self.knn[0])
                    _native.knn_point_wrapper(B, n, n, 16, x1, x1, self.knn[1])
                else:       # candidates = the valid points of frame 2 / frame 1
                    _lib.call("rtk_knn_point_masked", B, n, n, 16, x1.data_ptr(), x2.data_ptr(), self.knn[0].data_ptr(),
                              n_valid[B:].data_ptr(), _stream())
                    _lib.call("rtk_knn_point_masked", B, n, n, 16, x1.data_ptr(), x1.data_ptr(), self.knn[1].data_ptr(),
                              n_valid[:B].data_ptr(), _stream())
                self._record("knn", side)
            if tail_hook is not None:
                tail_hook(self)

    def _record(self, key, side):
        if side is not None:
            ev = torch.cuda.Event()
            ev.record(side)
            self.events[key] = ev

    def wait(self, key):
        """Make the current stream wait for geometry stage `key` (0,1,2 = levels, 'nn', 'knn')."""
        ev = self.events.get(key)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def head(self, count):
        """View on the first `count` samples (the pc1 half)."""
        g = object.__new__(Geometry)
        g.n, g.samples, g.npoint = self.n, count, self.npoint
        g.xyz = [x[:count] for x in self.xyz]
        g.ball = [[b[:count] for b in row] for row in self.ball]
        g.nuniq = [c[:count] for c in self.nuniq]
        g.fps_idx, g.tie = [i[:count] for i in self.fps_idx], self.tie[:count]
        g.events, g.knn = self.events, self.knn
        g.nn = {k: (d2[:count], idx[:count], m) for k, (d2, idx, m) in self.nn.items()}
        return g


CV_SPLIT_MAX_ROWS = 1 << 22      # rtk_cost_volume_split*: rows of p2 per launch (csrc/fused_split.hip)


def cv_shared_workgroups(samples, n1, dev):
    """Workgroups for the forward cost volume when other batches are in flight (rtk_cost_volume_split_shared): at most 3/4 of the CUs,
    at least half, a multiple of 8 (a share per XCD), the largest count in that range that leaves the slowest workgroup no more than 5 %
    above the average tile count -- else the best balanced one.  0 (all CUs) when the batch does not take the XCD-aware grid."""
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    if samples % 8 or cus % 8:
        return 0
    tiles = (samples // 8) * ((n1 + 7) // 8)                    # per XCD
    lo, hi = max(1, cus // 16), max(1, cus // 8 * 3 // 4)
    balance = lambda w: tiles / (w * -(-tiles // w))
    ok = [w for w in range(hi, lo - 1, -1) if balance(w) >= 0.95]
    w = ok[0] if ok else max(range(lo, hi + 1), key=balance)
    return 8 * min(w, tiles)


def sa_scale(geo, W, lvl, s, q, qcol, out, out_offset):
    """One MSG scale (level lvl, scale s) of PNHead weights W on geometry geo; q[:, qcol:] holds its layer-1 projection.
    Duplicate centroids (geo.nuniq) are skipped; gathers from duplicate source rows alias row 0."""
    sc = W.scales[lvl][s]
    qptr, qpitch = _colptr(q, qcol)
    optr, opitch = _colptr(out)
    src, dst = geo.xyz[lvl], geo.xyz[lvl + 1]
    src_nu = geo.nuniq[lvl - 1].data_ptr() if lvl > 0 else None     # level-0 source rows are the original points
    if _TRACE is not None:      # per (centroid, neighbour) pair: offset layer (3 + bias) x C1, then the resident chain
        macs = sc.nsample * (4 * sc.c1 + sum(co * ci for co, ci in sc.chain.dims))
        _TRACE.append(("sa_scale", _live_rows(geo.nuniq[lvl], geo.npoint, geo.samples), macs))
    if sc.split_image is not None:
        _lib.call("rtk_sa_scale_split", geo.samples, src.shape[1], geo.npoint, sc.nsample, src.data_ptr(), dst.data_ptr(),
                  geo.ball[lvl][s].data_ptr(), qptr, qpitch, sc.c1, sc.w1img.data_ptr(), sc.split_image.data_ptr(), sc.split_scale.data_ptr(),
                  sc.split_bias.data_ptr(),
                  optr, opitch, out_offset, src_nu, geo.nuniq[lvl].data_ptr(), _stream())
        return
    _lib.call("rtk_sa_scale", geo.samples, src.shape[1], geo.npoint, sc.nsample, src.data_ptr(), dst.data_ptr(),
              geo.ball[lvl][s].data_ptr(), qptr, qpitch, ceil16(sc.c1) // 16, sc.w1img.data_ptr(),
              sc.chain.n, sc.chain.arr, optr, opitch, out_offset, src_nu, geo.nuniq[lvl].data_ptr(), _stream())


def run_pnhead(W, geo, q1, out=None, gmax=None):
    """q1 (samples*n, 32): per-point sa1 layer-1 projections (scale 0 | scale 1).  Returns l0_points (samples*n, 128) (written
    into `out`, a possibly column-sliced (samples*n, 128) view, when given).
    All centroid-level tensors hold valid data only in rows < geo.nuniq[level][sample]; the rest are duplicates of the
    sample's row 0 and are never read (consumers alias them)."""
    S_, n, S = geo.samples, geo.n, geo.npoint
    dev = q1.device
    new = lambda rows, c: torch.empty(rows, c, dtype=torch.float32, device=dev)
    nu = geo.nuniq
    sa1 = new(S_ * S, 64)
    geo.wait(0)
    sa_scale(geo, W, 0, 0, q1, 0, sa1, 0)
    sa_scale(geo, W, 0, 1, q1, 16, sa1, 32)
    t1 = pointwise(S_ * S, S, [(sa1, 64, False)], W.trans[0], new(S_ * S, 96), row_nuniq=nu[0])      # l1_points | q2_s0 | q2_s1
    sa2 = new(S_ * S, 96)
    geo.wait(1)
    sa_scale(geo, W, 1, 0, t1, 32, sa2, 0)
    sa_scale(geo, W, 1, 1, t1, 64, sa2, 32)
    t2 = pointwise(S_ * S, S, [(sa2, 96, False)], W.trans[1], new(S_ * S, 192), row_nuniq=nu[1])     # l2_points | q3_s0 | q3_s1
    sa3 = new(S_ * S, 128)
    geo.wait(2)
    sa_scale(geo, W, 2, 0, t2, 64, sa3, 0)
    sa_scale(geo, W, 2, 1, t2, 128, sa3, 64)
    l3 = pointwise(S_ * S, S, [(sa3, 128, False)], W.lin3, new(S_ * S, 64), row_nuniq=nu[2])
    geo.wait("nn")
    d2, idx, m = geo.nn["fp3"]
    f3 = pointwise(S_ * S, S, [(t2[:, 0:64], 64, False)], W.fp["fp3"], new(S_ * S, 128), row_nuniq=nu[1],
                   interp=(l3, 64, m, idx.reshape(-1, 3), d2.reshape(-1, 3), nu[2]))
    d2, idx, m = geo.nn["fp2"]
    f2 = pointwise(S_ * S, S, [(t1[:, 0:32], 32, False)], W.fp["fp2"], new(S_ * S, 128), row_nuniq=nu[0],
                   interp=(f3, 128, m, idx.reshape(-1, 3), d2.reshape(-1, 3), nu[1]))
    d2, idx, m = geo.nn["fp1"]
    if gmax is None:                                                      # global max-pool, fused into fp1's epilogue (ZERO-initialised)
        gmax = torch.zeros(S_, 128, dtype=torch.float32, device=dev)
    out = pointwise(S_ * n, n, [], W.fp["fp1"], out if out is not None else new(S_ * n, 128),
                    interp=(f2, 128, m, idx.reshape(-1, 3), d2.reshape(-1, 3), nu[0]), colmax=gmax)
    return out, gmax


class FusedBackbone:
    """Eval-mode Track4D.backbone (models/track4d.py:67-106) on the fused kernels."""

    def __init__(self, model, split=True):
        """split=False: the cost volume on the fp32-input MFMA kernel (the comparison implementation of the tests)."""
        sd = {k: v.detach() for k, v in model.state_dict().items()}
        dev = next(model.parameters()).device
        self.dev = dev
        self.npoint = model.pn_head.sa1.npoint
        # 5-layer GRU on a length-1 sequence (model_utils.py:279,296): transposed weights for rtk_gru_step
        g = lambda k: torch.stack([sd["fd_layer.torchGRU.%s_l%d" % (k, l)].float() for l in range(5)])
        self.gru_wih = g("weight_ih").transpose(1, 2).contiguous()      # (L, H, 3H)
        self.gru_whh = g("weight_hh").transpose(1, 2).contiguous()
        self.gru_bih, self.gru_bhh = g("bias_ih").contiguous(), g("bias_hh").contiguous()
        self.kernel_events = None      # set to a list to record (start, stop) events around the dominant kernel
        self.kernel_token = None       # [last stop event] shared by the engines of a GraphPipeline while kernel_events is set
        self._split_hook = None
        self.side, self.use_side_stream = None, True    # geometry kernels on a forked stream (GraphPipeline drops it beyond depth 2)
        self.cv_shared = False                          # the cost volume on a share of the CUs (GraphPipeline sets it from depth 3)
        self._last_cv = None
        self.enc = _PNHeadWeights(sd, "pn_head.", dev)
        self.dec = _PNHeadWeights(sd, "fd_layer.mse.", dev)
        z = lambda n: torch.zeros(n, dtype=torch.float64, device=sd["bin_score"].device)
        # encoder sa1 projection of the raw (RCS, v_r) features
        self.enc_q1 = Chain([(self.enc.wq1, z(32), ACT_NONE)], dev)
        self.enc_q1_w = self.enc.wq1.float().to(dev).contiguous()            # (32, 2): evaluated by the geometry's first launch
        # cost volume (fc_layer): conv0 split by input segment [f1 loc|glob (256) || f2 loc|glob (256) || dir (3)]
        w0 = sd["fc_layer.mlp_convs.0.weight"].double().reshape(256, 515)
        b0 = sd["fc_layer.mlp_convs.0.bias"].double()
        self.p1_loc = Chain([(w0[:, 0:128], z(256), ACT_NONE)], dev)
        self.p2_loc = Chain([(w0[:, 256:384], z(256), ACT_NONE)], dev)
        # the halves of conv0 that act on the broadcast global features are per-sample terms (rtk_global_terms: transposed fp32 weights)
        tr = lambda w: w.float().t().contiguous().to(dev)
        self.p1_glob_wt, self.p1_glob_b = tr(w0[:, 128:256]), b0.float().to(dev).contiguous()      # carries the layer's bias
        self.p2_glob_wt = tr(w0[:, 384:512])
        self.cv_wd = offset_image(torch.cat([w0[:, 512:515], z(256)[:, None]], 1), dev)
        self.cv_layers = Chain([(sd["fc_layer.mlp_convs.%d.weight" % i].double().reshape(256, 256),
                                 sd["fc_layer.mlp_convs.%d.bias" % i].double(), ACT_LEAKY) for i in (1, 2)], dev)
        # the same two layers as split images (csrc/split_mfma.h): fp32 results from the fp16 matrix pipe, 3/16 of the matrix time
        self.cv_split = bool(split)
        w23 = [sd["fc_layer.mlp_convs.%d.weight" % i].double().reshape(256, 256).float().to(dev).contiguous() for i in (1, 2)]
        self.cv_bias23 = torch.stack([sd["fc_layer.mlp_convs.%d.bias" % i].double().float() for i in (1, 2)]).to(dev).contiguous()
        self.cv_images = torch.empty(2 * SPLIT_IMAGE_256, dtype=torch.int16, device=dev)
        self.cv_scales = torch.empty(2, dtype=torch.float32, device=dev)
        for l, w in enumerate(w23):
            _lib.call("rtk_pack_split_layer", 256, 256, w.data_ptr(), 0, self.cv_images[l * SPLIT_IMAGE_256:].data_ptr(),
                      self.cv_scales[l:].data_ptr(), _stream())
        torch.cuda.current_stream().synchronize()      # w23 may go
        self.wn1 = _WeightNet(sd, "fc_layer.weightnet1", dev)
        self.wn2 = _WeightNet(sd, "fc_layer.weightnet2", dev)
        # heads
        def predictor(prefix, first_cols):
            layers = []
            for i in range(3):
                w, b = fold_bn(sd["%s.sf_mlp.%d.0.weight" % (prefix, i)], "%s.sf_mlp.%d.1" % (prefix, i), sd)
                layers.append([w, b, ACT_RELU])
            return layers
        cp = predictor("fd_layer.cp", None)
        wl, bl = sd["fd_layer.cp.linear.weight"].double(), sd["fd_layer.cp.linear.bias"].double()
        wc2 = sd["fd_layer.cp.conv2.weight"].double().reshape(3, 32)
        cp.append([wl @ wc2, bl, ACT_SIGMOID])                              # conv2 (no bias) then Linear(3,1): one linear map
        self.cls_head = Chain([tuple(x) for x in cp], dev)
        fp = predictor("fd_layer.fp", None)
        w_first = fp[0][0]
        self.flow_glob_wt, self.flow_glob_b = tr(w_first[:, 128:256]), fp[0][1].float().to(dev).contiguous()   # per-sample GRU term + folded BN shift
        fp[0] = [w_first[:, 0:128], z(128), ACT_RELU]
        fp.append([sd["fd_layer.fp.conv2.weight"].double().reshape(3, 32), z(3), ACT_NONE])
        self.flow_head = Chain([tuple(x) for x in fp], dev)
        # decoder embeddings [feature1 (2) || pc1 local (128) || pc1 global (128) || cor (256)] -> mse.sa1 projections
        wq = self.dec.wq1                                                    # (32, 514)
        wq_pad = torch.zeros(32, 16, dtype=torch.float64, device=wq.device)
        wq_pad[:, :2] = wq[:, 0:2]
        self.dec_q1 = Chain([(torch.cat([wq_pad, wq[:, 2:130], wq[:, 258:514]], 1), z(32), ACT_NONE)], dev)
        self.dec_q1_glob_wt = tr(wq[:, 130:258])

    # --------------------------------------------------------------------------------------------------
    def backbone(self, pc1, pc2, feature1, feature2, h, n_valid=None):
        """n_valid (2,B) int32: point counts of a padded variable-N batch (row 0: frame 1, row 1: frame 2), see
        vod_gt.pad_frame_pairs; outputs at padded positions are those of the sample's point 0 / unspecified."""
        B, _, N = pc1.shape
        dev = pc1.device
        new = lambda rows, c: torch.empty(rows, c, dtype=torch.float32, device=dev)
        xyz = torch.empty(2 * B, N, 3, dtype=torch.float32, device=dev)
        raw = new(2 * B * N, 4)
        q1 = new(2 * B * N, 32)
        # keep the (possibly copied) contiguous inputs referenced until the launch is enqueued: a temporary freed
        # between two .data_ptr() calls could be recycled by the allocator for the next temporary
        ins = [t.contiguous() for t in (pc1, pc2, feature1, feature2)]
        if self.side is None and self.use_side_stream:
            self.side = torch.cuda.Stream(device=dev)
        if n_valid is not None:
            n_valid = n_valid.to(device=dev, dtype=torch.int32).reshape(2 * B).contiguous()
        # layout conversion (rtk_prepare_inputs) + every geometry table: two launches (rtk_geometry_front, rtk_geometry_tables)
        geo = Geometry(xyz, self.npoint, side=self.side if self.use_side_stream else None, knn_frames=B, n_valid=n_valid,
                       prepare=(ins[0], ins[1], ins[2], ins[3], raw), q1=(self.enc_q1_w, q1))
        geo.wait("front")       # raw, q1 (and xyz) come out of the geometry's first launch when it runs on the side stream
        # ---- encoder over both frames at once (same weights; eval-mode BN is per-element) --------------
        if not geo.q1_done:
            pointwise(2 * B * N, N, [(raw, 2, False)], self.enc_q1, q1)
        elif _TRACE is not None:
            _TRACE.append(("pointwise", 2 * B * N, 64))
        # pc{1,2}_features = [local (128) | global max broadcast (128)] (models/track4d.py:89-95) live in ONE point-major buffer:
        # the encoder's last layer writes the local half in place, one broadcast copy fills the global half, and the API's
        # (B,256,N) tensors are permuted VIEWS of it -- no layout pass over the outputs
        feat12 = new(2 * B * N, 256)
        gmax = torch.zeros(3 * B, 128, dtype=torch.float32, device=dev)                  # both PNHeads' global max-pools: one fill
        loc, glob = run_pnhead(self.enc, geo, q1, out=feat12[:, 0:128], gmax=gmax[:2 * B])   # (2B*N, 128) view, (2B, 128)
        f1, f2 = loc[:B * N], loc[B * N:]
        # everything that is a function of the global features, one launch: the per-sample terms of the cost volume's first layer
        # (frame 1: with the layer's bias; frame 2) and of the decoder's sa1 projection, and the broadcast that fills the global half
        # of pc{1,2}_features
        sb1, sb2, sbq = new(B, 256), new(B, 256), new(B, 32)
        global_terms(glob, [(self.p1_glob_wt, self.p1_glob_b, sb1, 0), (self.p2_glob_wt, None, sb2, B), (self.dec_q1_glob_wt, None, sbq, 0)],
                     bcast=feat12[:, 128:], n=N)
        # ---- cost volume ---------------------------------------------------------------------------------
        p1 = pointwise(B * N, N, [(f1, 128, False)], self.p1_loc, new(B * N, 256), sample_bias=sb1)
        p2 = pointwise(B * N, N, [(f2, 128, False)], self.p2_loc, new(B * N, 256), sample_bias=sb2)
        x1, x2 = xyz[:B], xyz[B:]
        geo.wait("knn")
        knn1, knn2 = geo.knn
        cor1 = new(B * N, 256)
        self._last_cv = (B, N, x1, x2, knn1, p1, p2, cor1)
        if self._split_hook is not None:
            # segmented capture (capture(split_cost_volume=True)): the graph ends here, the dominant kernel is launched
            # eagerly between two HIP events at replay time, a second graph takes over.  Every geometry stage has been
            # joined (the kNN tables are the last work on the side stream); later wait() calls must not reference events
            # of the finished capture.
            geo.events.clear()
            self._split_hook()
        else:
            ev = self.kernel_events
            if ev is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            self._cost_volume(*self._last_cv)
            if ev is not None:
                e1.record()
                ev.append((e0, e1))
        cor = new(B * N, 256)
        if _TRACE is not None:
            _TRACE.append(("patch_cost", B * N * 16, (3 * 8 + 8 * 8 + 8 * 256) + 256))
        _lib.call("rtk_patch_cost", B, N, x1.data_ptr(), knn2.data_ptr(), cor1.data_ptr(), 256, self.wn2.arr, cor.data_ptr(), 256, 0,
                  _stream())
        # ---- decoder -------------------------------------------------------------------------------------
        cls = torch.empty(B, N, dtype=torch.float32, device=dev)
        pointwise(B * N, N, [(cor, 256, False)], self.cls_head, cls, out_channels=1, channel_major=True)
        q1d = pointwise(B * N, N, [(raw[:B * N], 2, False), (f1, 128, False), (cor, 256, False)], self.dec_q1, new(B * N, 32),
                        sample_bias=sbq)
        prop, gfeat = run_pnhead(self.dec, geo.head(B), q1d, gmax=gmax[2 * B:])           # (B*N,128), (B,128)
        if h is None:
            h = torch.zeros(5, B, 128, device=dev, dtype=torch.float32)
        gout, h_out, sbf = self._gru_step(gfeat, h)          # sbf: the flow head's per-sample term W_glob gout + b (the kernel's epilogue)
        flow = torch.empty(B, 3, N, dtype=torch.float32, device=dev)
        pointwise(B * N, N, [(prop, 128, False)], self.flow_head, flow, out_channels=3, sample_bias=sbf, channel_major=True)
        # ---- API layouts (B,C,N): permuted views of the point-major tensors (same values as the reference's, no copy) ------
        f12 = feat12.view(2 * B, N, 256).permute(0, 2, 1)
        pc1_features, pc2_features = f12[:B], f12[B:]
        cor_cm = cor.view(B, N, 256).permute(0, 2, 1)
        prop_cm = prop.view(B, N, 128).permute(0, 2, 1)
        return flow, h_out, cls, cor_cm, pc1_features, pc2_features, prop_cm

    # --------------------------------------------------------------------------------------------------
    def _cost_volume(self, B, N, x1, x2, knn1, p1, p2, cor1):
        if _TRACE is not None:      # per (point, neighbour) pair: direction term, layers 2+3, WeightNet 3-8-8-256, weighted sum
            _TRACE.append(("cost_volume", B * N * 16, 3 * 256 + 2 * 256 * 256 + (3 * 8 + 8 * 8 + 8 * 256) + 256))
        if self.cv_split:
            # the kernel requests the gathered p2 rows with 32-bit byte offsets: at most CV_SPLIT_MAX_ROWS rows of p2 per launch --
            # larger batches go in slices of whole samples (the samples are independent)
            step = B if B * N <= CV_SPLIT_MAX_ROWS else max(1, CV_SPLIT_MAX_ROWS // N)
            for b0 in range(0, B, step):
                nb = min(step, B - b0)
                wgs = cv_shared_workgroups(nb, N, x1.device) if self.cv_shared else 0
                r0 = b0 * N
                _lib.call("rtk_cost_volume_split_shared", nb, N, N, x1[b0:].data_ptr(), x2[b0:].data_ptr(), knn1[b0:].data_ptr(), p1[r0:].data_ptr(),
                          p2[r0:].data_ptr(), self.cv_wd.data_ptr(), self.cv_images.data_ptr(), self.cv_scales.data_ptr(),
                          self.cv_bias23[0].data_ptr(), self.cv_bias23[1].data_ptr(), self.wn1.arr, cor1[r0:].data_ptr(), 256, wgs, _stream())
            return
        _lib.call("rtk_cost_volume", B, N, N, x1.data_ptr(), x2.data_ptr(), knn1.data_ptr(), p1.data_ptr(), p2.data_ptr(),
                  self.cv_wd.data_ptr(), self.cv_layers.arr, self.wn1.arr, cor1.data_ptr(), 256, _stream())

    def _gru_step(self, x, h):
        L, B, H = h.shape
        if _TRACE is not None:
            _TRACE.append(("gru", B * L, 2 * 3 * H * H))
            _TRACE.append(("global_terms", B, H * self.flow_glob_wt.shape[1]))
        h_out = torch.empty_like(h)
        y = torch.empty(B, H, dtype=torch.float32, device=h.device)
        sbf = torch.empty(B, self.flow_glob_wt.shape[1], dtype=torch.float32, device=h.device)
        h = h.contiguous()
        _lib.call("rtk_gru_step_head", B, L, H, x.data_ptr(), h.data_ptr(), self.gru_wih.data_ptr(), self.gru_whh.data_ptr(),
                  self.gru_bih.data_ptr(), self.gru_bhh.data_ptr(), h_out.data_ptr(), y.data_ptr(), self.flow_glob_wt.data_ptr(),
                  self.flow_glob_b.data_ptr(), sbf.data_ptr(), sbf.shape[1], _stream())
        return y, h_out, sbf

    def time_dominant_kernel(self, iters=20):
        """(start, stop) HIP-event pairs around `iters` launches of the cost-volume kernel on the current
        stream, on the operands of the last backbone() call."""
        assert self._last_cv is not None, "run backbone() first"
        ev = []
        for _ in range(iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self._cost_volume(*self._last_cv)
            e1.record()
            ev.append((e0, e1))
        torch.cuda.synchronize()
        return ev

    def capture(self, pc1, pc2, feature1, feature2, h, split_cost_volume=False):
        """Capture one backbone() into a hipGraph (all launches, ours and the glue ops, are on the capture
        stream).  Returns step(pc1=None, ...) -> outputs: new inputs are copied into the static buffers.
        split_cost_volume: capture TWO graphs around the dominant kernel and launch it eagerly between them, bracketed by
        HIP events appended to self.kernel_events -- the kernel's duration measured in situ (same stream, same
        neighbours in flight) at the cost of two extra launches per step."""
        static = [t.clone() for t in (pc1, pc2, feature1, feature2, h)]
        saved, self.kernel_events = self.kernel_events, None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self.backbone(*static)
        torch.cuda.current_stream().wait_stream(side)
        if not split_cost_volume:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                outs = self.backbone(*static)
            self.kernel_events = saved

            def step(*new_inputs):
                pairs = [(d, s_) for d, s_ in zip(static, new_inputs) if s_ is not None]
                if pairs:       # ONE launch (rtk_copy_multi) into the static buffers instead of five device-to-device memcpys
                    _copy_inputs(pairs)
                graph.replay()
                return outs
            step.graph = graph
            return step

        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())

        def hook():
            g1.capture_end()
            g2.capture_begin(pool=g1.pool())

        with torch.cuda.stream(cap):
            self._split_hook = hook
            try:
                g1.capture_begin()
                outs = self.backbone(*static)
                g2.capture_end()
            finally:
                self._split_hook = None
        torch.cuda.current_stream().wait_stream(cap)
        cv_args = self._last_cv
        self.kernel_events = saved
        eng = self

        def step(*new_inputs):
            pairs = [(d, s_) for d, s_ in zip(static, new_inputs) if s_ is not None]
            if pairs:
                _copy_inputs(pairs)
            g1.replay()
            ev = eng.kernel_events
            if ev is not None:
                tok = eng.kernel_token
                if tok is not None and tok[0] is not None:
                    # the measured kernels of the batches in flight run one after the other (they would otherwise time-share the
                    # CUs and every event pair would span its neighbours' run time as well); everything else still overlaps
                    torch.cuda.current_stream().wait_event(tok[0])
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            eng._cost_volume(*cv_args)
            if ev is not None:
                e1.record()
                ev.append((e0, e1))
                if tok is not None:
                    tok[0] = e1
            g2.replay()
            return outs
        step.graph = (g1, g2)
        return step


def global_terms(g, jobs, bcast=None, n=0):
    """rtk_global_terms: g (samples, cin); jobs [(wt (cin, cout), bias (cout) or None, out (count, cout), s0)]: out = W g[s0 : s0 + count] + b;
    bcast (samples * n, pitch) view: g[s] copied into its first cin columns for every row of sample s."""
    samples, cin = g.shape
    arr = (_GtJob * max(len(jobs), 1))()
    for j, (wt, bias, out, s0) in enumerate(jobs):
        assert wt.shape == (cin, out.shape[1]) and out.is_contiguous() and wt.is_contiguous()
        arr[j].wt, arr[j].bias, arr[j].out = wt.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr()
        arr[j].cout, arr[j].s0, arr[j].count, arr[j].out_pitch = out.shape[1], s0, out.shape[0], out.shape[1]
    if _TRACE is not None:
        _TRACE.append(("global_terms", 1, sum(cin * o.shape[0] * o.shape[1] for _, _, o, _ in jobs)))
    bptr, bpitch = _colptr(bcast) if bcast is not None else (None, 0)
    _lib.call("rtk_global_terms", samples, cin, g.data_ptr(), len(jobs), arr, bptr, bpitch, n, _stream())


def _copy_inputs(pairs):
    """New inputs into a captured step's static buffers: one launch when every pair is a plain same-dtype contiguous copy, the framework's
    multi-tensor copy otherwise (strided / other-dtype inputs)."""
    if all(d.is_contiguous() and s_.is_contiguous() and d.shape == s_.shape and d.dtype == s_.dtype and d.device == s_.device and
           d.element_size() == 4 for d, s_ in pairs):
        copy_multi(pairs)
    else:
        torch._foreach_copy_([p[0] for p in pairs], [p[1] for p in pairs])


class GraphPipeline:
    """Batch-level software pipeline for throughput: `depth` captured backbone graphs, each with its own static
    buffers and geometry side stream, replayed round-robin on `depth` streams.  While one batch sits in its
    latency-bound phases (FPS chain, small gathers) the MFMA stages of the previous/next batch keep the CUs busy
    (measured on MI355X at B=64, N=256, round 2: 1.71 ms per batch with one graph in flight, 1.41 with two, 1.31 with four).
    With more than two batches in flight the engines run their geometry kernels on their own stream instead of a forked one:
    depth x 2 streams oversubscribe the four hardware queues and serialise falsely (1.43 ms at depth 4 with forked geometry
    streams), and four independent chains fill the idle phases better than two forked ones.
    Weights are shared between the engines.  Usage:  p = GraphPipeline(net, example_inputs); outs = p.submit(*inputs)
    ... p.drain().  Outputs of a submit stay valid until the same slot is reused, `depth` submits later."""

    def __init__(self, model_or_engine, example_inputs, depth=2, split_cost_volume=False):
        import copy
        eng = model_or_engine if isinstance(model_or_engine, FusedBackbone) else FusedBackbone(model_or_engine)
        if depth > 2:
            eng = copy.copy(eng)
            eng.use_side_stream, eng.side = False, None
            # ... and the cost volume leaves a quarter of every XCD to the other batches: it keeps the CUs it runs on whole, and with
            # one workgroup per CU nothing else runs for a third of the step (B = 64: 74.1 -> 76.0 k pairs/s at 192 of 256 workgroups,
            # the kernel itself 0.33 -> 0.39 ms; 232: 73.2 k, 208: 74.4 k, 176: 74.9 k, 160: 75.7 k, 128: 75.5 k)
            eng.cv_shared = True
        self.engines = [eng]
        for _ in range(depth - 1):
            e = copy.copy(eng)          # shallow: packed weights are shared, per-engine state is reset below
            e.side, e._last_cv, e.kernel_events = None, None, None
            self.engines.append(e)
        self.depth = depth
        self._token = [None]
        for e in self.engines:
            e.kernel_token = self._token
        with torch.no_grad():
            self.steps = [e.capture(*example_inputs, split_cost_volume=split_cost_volume) for e in self.engines]
        self.streams = [torch.cuda.Stream() for _ in range(depth)]
        self.i = 0
        self._forked = False

    def submit(self, *inputs):
        cur = torch.cuda.current_stream()
        if not self._forked:
            for s in self.streams:
                s.wait_stream(cur)
            self._forked = True
        k = self.i % self.depth
        self.i += 1
        with torch.cuda.stream(self.streams[k]):
            return self.steps[k](*inputs)

    def set_kernel_events(self, events):
        """Every engine appends its (start, stop) event pairs around the dominant kernel to `events` (None: stop)."""
        for e in self.engines:
            e.kernel_events = events

    def drain(self):
        """Join all in-flight batches into the current stream."""
        cur = torch.cuda.current_stream()
        for s in self.streams:
            cur.wait_stream(s)
        self._forked = False
