"""Training-mode PNHead on de-duplicated levels.

The reference samples npoint = 512 centroids from clouds of N = 256 points (configs.yaml: num_points 256, npoints 512),
so at every set-abstraction level at least half of the centroid rows are exact copies of centroid 0
(SURVEY.md A.5).  Its training step convolves, batch-normalises and back-propagates through all of them
(lib/pointnet2_modules.py:10-56).  Here every level tensor keeps ONE row per unique centroid:

  * geometry (FPS / ball query / three-NN, no gradient) comes from the same `Geometry` tables the inference engine
    uses; indices that point at a duplicate row are redirected to row 0 (identical values);
  * BatchNorm statistics count the missing copies through a per-row weight (w[b][0] = 1 + npoint - nuniq[b]), and the
    backward of the weighted operator is the exact gradient of the reference computation (include/rtk_train.h);
  * BatchNorm + ReLU (+ the max over the neighbourhood) is one HIP operator per layer (train_ops.bn_relu) instead of
    MIOpen batch-norm + ReLU + max_pool2d kernels, forward and backward;
  * the two per-frame encoder calls (models/track4d.py:88-92) run as one stacked batch with per-frame statistics
    (groups = 2), running statistics updated frame 1 first, then frame 2.

Parameters, state-dict keys, running statistics and results are those of the module path (pointnet2_modules.py /
model_utils.PNHead) up to fp32 summation order; tests/test_train_gpu.py checks both against the golden train step.
"""
import torch
import torch.nn.functional as F

from . import _lib
from . import pointnet2_utils as PU
from .train_ops import bn_relu, conv1x1, cost_volume, patch_cost, pw_bn_relu, pw_linear, sa_chain, sa_chain_supported


FUSED_SA_CHAIN = True      # False: one bn_relu + framework convolution per layer (reference structure, kept for tests)


class TrainGeometry:
    """De-duplicated view of fused.Geometry for one stacked batch of clouds xyz (S_, n, 3).

    With `side` (a torch.cuda.Stream) every geometry kernel -- FPS, ball queries, three-NN, and this class's own tables -- is
    enqueued on that stream, forked from the current one; consumers call wait(key) (key = level 0..2, "interp", "inv") before
    the first use, so the ~0.5 ms chain of small latency-bound geometry kernels overlaps the feature kernels instead of
    preceding them.  All buffers are allocated on the current stream before the fork; join() makes the current stream wait
    for everything (the forward ends with it: inside a stream capture every forked stream must be joined)."""

    def __init__(self, xyz, npoint, side=None, n_valid=None, groups=1):
        """n_valid (S_,) int32 on the device: padded batch (vod_gt.pad_frame_pairs) -- cloud s consists of its first n_valid[s] points,
        the rest are copies of its point 0.  FPS applies the unpadded cloud's tie rule, ball indices that hit a padding row are
        redirected to row 0 (same values), and the per-point BatchNorm layers weigh padding rows with 0 and take their per-group
        element counts from the device (`point_w`, `point_counts`; groups = number of consecutive batch slices with their own
        statistics): results, gradients and running statistics of every sample equal those of its own unpadded B = 1 run, and
        the captured step does not depend on the clouds' sizes."""
        from . import fused
        S_, n, _ = xyz.shape
        self.samples, self.n, self.npoint = S_, n, npoint
        self.xyz = xyz                            # (S_, n, 3) contiguous: the correlator takes its point-major coordinates from here
        self.n_valid, self.point_w, self.point_counts = n_valid, None, None
        if n_valid is not None:
            assert n_valid.shape == (S_,) and n_valid.dtype == torch.int32 and n_valid.is_cuda and n_valid.is_contiguous()
            self.point_w = torch.empty(S_, n, dtype=torch.float32, device=xyz.device)
            self.point_counts = torch.empty(groups, dtype=torch.float64, device=xyz.device)
            _lib.call("rtk_train_point_weights", S_, n, groups, n_valid.data_ptr(), self.point_w.data_ptr(), self.point_counts.data_ptr(),
                      torch.cuda.current_stream().cuda_stream)
        U = self.U = min(n, npoint)
        dev = xyz.device
        f32 = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)
        i32 = lambda *sh: torch.empty(*sh, dtype=torch.int32, device=dev)
        NS = fused._PNHeadWeights.NSAMPLES
        self.row_w = [f32(S_, U) for _ in range(3)]
        self.ball = [[i32(S_, U, ns) for ns in NS[lvl]] for lvl in range(3)]
        self.dxyz = [[f32(S_, 3, U, ns) for ns in NS[lvl]] for lvl in range(3)]
        self.inv = []
        for lvl in range(3):
            row = []
            for ns in NS[lvl]:
                n_src = n if lvl == 0 else U
                # positions sorted by the source row they gather: the first layer's backward is a gather instead of a scatter
                row.append((i32(S_, n_src + 1), torch.empty(S_, U * ns, dtype=torch.int16, device=dev))
                           if U * ns <= 65536 and n_src <= 8192 and ns % 4 == 0 else None)
            self.inv.append(row)
        self.interp = {name: (i32(S_, n if u == 0 else U, 3), f32(S_, n if u == 0 else U, 3))
                       for name, (u, k) in {"fp3": (2, 3), "fp2": (1, 2), "fp1": (0, 1)}.items()}
        # inverse tables of the interpolation indices (known point -> the (unknown point, slot) positions that use it): the
        # interpolation's backward is a gather as well
        self.interp_inv = {name: (i32(S_, U + 1), torch.empty(S_, 3 * io.shape[1], dtype=torch.int16, device=dev))
                           if 3 * io.shape[1] <= 65536 and io.shape[1] <= 2048 else None for name, (io, _) in self.interp.items()}
        st = lambda: torch.cuda.current_stream().cuda_stream

        def level_tables(geo, lvl):
            nu = geo.nuniq
            _lib.call("rtk_train_row_weights", S_, U, npoint, nu[lvl].data_ptr(), self.row_w[lvl].data_ptr(), st())
            src, dst = geo.xyz[lvl], geo.xyz[lvl + 1]             # (S_, n or npoint, 3), (S_, npoint, 3)
            for s in range(2):
                # source rows the level tensor does not hold (>= U; level 0 holds all n) are copies of row 0: redirect; neighbour -
                # centroid offsets (no gradient).  Rows in [nuniq, U) and a padded batch's padding rows exist as copies and are used.
                _lib.call("rtk_train_group_geometry", S_, src.shape[1], npoint, U, NS[lvl][s], src.data_ptr(), dst.data_ptr(),
                          geo.ball[lvl][s].data_ptr(), U if lvl > 0 else n, nu[lvl].data_ptr(), self.ball[lvl][s].data_ptr(),
                          self.dxyz[lvl][s].data_ptr(), st())

        def tail_tables(geo):
            from .train_ops import group_inverse_index_multi
            nu = geo.nuniq
            jobs = []
            for name, (u, k) in {"fp3": (2, 3), "fp2": (1, 2), "fp1": (0, 1)}.items():
                d2, idx, _ = geo.nn[name]
                io, wo = self.interp[name]
                _lib.call("rtk_train_interp_weights", S_, d2.shape[1], io.shape[1], d2.data_ptr(), idx.data_ptr(), nu[k - 1].data_ptr(),
                          io.data_ptr(), wo.data_ptr(), st())
                if self.interp_inv[name] is not None:              # the redirected indices point at rows < nuniq <= U
                    off, inv = self.interp_inv[name]
                    # fp3 / fp2: the unknown rows are level centroids, the duplicate ones (>= nuniq) are read by nothing downstream
                    # (every consumer redirects them): zero gradient, kept out of the table
                    # fp1 on padded clouds: the padding points are copies of point 0; their positions stay out as well and the
                    # backward folds their gradient into point 0's (three_interpolate(..., n_valid))
                    jobs.append((U, 3 * io.shape[1], io, off, inv, nu[u - 1] if u > 0 else self.n_valid, 3))
            geo._record("interp", side)
            for lvl in range(3):                                   # needed by the backward only
                for s in range(2):
                    if self.inv[lvl][s] is not None:
                        off, inv = self.inv[lvl][s]
                        jobs.append((n if lvl == 0 else U, U * NS[lvl][s], self.ball[lvl][s], off, inv))
            if jobs:                                               # all nine inverse tables: one launch
                group_inverse_index_multi(S_, jobs)
            geo._record("inv", side)

        from .train_ops import arena_zeros
        geo = fused.Geometry(xyz, npoint, side=side, knn_frames=0, finite=True, n_valid=n_valid, level_hook=level_tables, tail_hook=tail_tables,
                             zeros=arena_zeros)
        self.events, self.side = geo.events, side
        self.l3_xyz = geo.xyz[3]
        self._geo = geo                                            # keeps the tables' inputs alive until the side stream is joined

    def wait(self, key):
        """Make the current stream wait for stage `key`: 0..2 (level tables), "interp", "inv"."""
        ev = self.events.get(key)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def join(self):
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)

    def head(self, count):
        """View on the first `count` samples (the pc1 half of a stacked batch)."""
        g = object.__new__(TrainGeometry)
        g.samples, g.n, g.npoint, g.U = count, self.n, self.npoint, self.U
        g.row_w = [w[:count] for w in self.row_w]
        g.ball = [[b[:count] for b in row] for row in self.ball]
        g.dxyz = [[d[:count] for d in row] for row in self.dxyz]
        g.inv = [[None if t is None else (t[0][:count], t[1][:count]) for t in row] for row in self.inv]
        g.interp = {k: (i[:count], w[:count]) for k, (i, w) in self.interp.items()}
        g.interp_inv = {k: None if t is None else (t[0][:count], t[1][:count]) for k, t in self.interp_inv.items()}
        g.l3_xyz = self.l3_xyz[:count]
        g.events, g.side, g._geo = self.events, self.side, self._geo
        # padded batch: the head is the first statistics group (frame 1 of a stacked pair batch)
        g.n_valid = None if self.n_valid is None else self.n_valid[:count]
        g.point_w = None if self.point_w is None else self.point_w[:count]
        g.point_counts = None if self.point_counts is None else self.point_counts[:1]
        return g


def supported(head):
    """The de-duplicated path covers the configuration RaTrack instantiates: max-pooled MSG levels whose SharedMLP
    layers are Conv2d(no bias) + BatchNorm2d + ReLU."""
    for sa in (head.sa1, head.sa2, head.sa3):
        if sa.pool_method != "max_pool" or sa.npoint is None:
            return False
        for mlp in sa.mlps:
            for layer in mlp.children():
                if not hasattr(layer, "bn") or not hasattr(layer, "activation") or layer.conv.bias is not None:
                    return False
    return True


def _sa_scale(mlp, tg, lvl, s, feats, groups):
    """One MSG scale: (project -> gather) + offset conv -> [BN+ReLU -> conv]* -> BN+ReLU+max.  feats: list of (S_,C_i,n_src)
    tensors whose channel concatenation is the level's feature tensor (never materialised)."""
    feats = list(feats) if isinstance(feats, (list, tuple)) else [feats]
    layers = list(mlp.children())
    w = layers[0].conv.weight                                     # (C1, 3+C, 1, 1): [d_xyz | features]
    if hasattr(tg, "wait"):
        tg.wait(lvl)
    idx = tg.ball[lvl][s]
    ns = idx.shape[2]
    count = (tg.samples // groups) * tg.npoint * ns
    if FUSED_SA_CHAIN and sa_chain_supported(layers) and ns >= 4:
        inv = tg.inv[lvl][s] if getattr(tg, "inv", None) is not None else None
        if inv is not None and getattr(tg, "events", None):
            inv = inv + (tg.events.get("inv"),)                    # the backward waits for the table (built last on the side stream)
        return sa_chain(feats, w, idx, tg.dxyz[lvl][s], layers, tg.row_w[lvl], count, groups, inv=inv)
    # reference structure (kept for tests): per-POINT projection by the layer's feature columns (a 1x1 conv and a gather commute)
    cols, c = [], 3
    for f in feats:
        cols.append(c)
        c += f.shape[1]
    proj = pw_linear(feats, w, cols=cols)
    wx = w[:, :3]
    z = conv1x1(tg.dxyz[lvl][s], wx) + PU.grouping_operation(proj, idx)
    x = None
    for i, layer in enumerate(layers):
        if i > 0:
            z = conv1x1(x, layer.conv.weight)
        x = bn_relu(z, layer.bn.bn, tg.row_w[lvl], count, groups, pool=(i == len(layers) - 1))
    return x                                                      # (S_, C_out, U)


def _fp(fp, tg, name, skip, known_feats, row_w, count_rows, groups, group_counts=None):
    if hasattr(tg, "wait"):
        tg.wait("interp")
    idx, weight = tg.interp[name]
    table = getattr(tg, "interp_inv", {}).get(name)
    if table is not None and known_feats.shape[2] + 1 == table[0].shape[1]:
        from .train_ops import three_interpolate
        x = three_interpolate(known_feats, idx, weight, table, n_valid=tg.n_valid if name == "fp1" else None)
    else:
        x = PU.three_interpolate(known_feats.contiguous(), idx, weight)
    srcs = [x] if skip is None else [x, skip]                      # lib/pointnet2_modules.py:150-153: cat([interpolated, skip])
    for layer in fp.mlp.children():
        x = pw_bn_relu(srcs, layer.conv.weight, layer.bn.bn, row_w, (tg.samples // groups) * count_rows, groups, group_counts=group_counts)
        srcs = [x]
    return x


_MSG_SIDE = {}


def side_stream(device, key):
    """A persistent side stream per (device, key)."""
    return _MSG_SIDE.setdefault((device, key), torch.cuda.Stream(device=device))


MSG_STREAMS = True      # the scales of an MSG level on streams of their own (see _msg_scales; train step 7.44 -> 7.16 ms at B = 64)
def _msg_scales(mlps, tg, lvl, feats, groups):
    """The scales of one MSG level (independent SharedMLP chains over different ball-query tables of the same centroids).  With
    MSG_STREAMS the scales after the first run on side streams forked from the current one and joined before the level's linear
    layer: their kernels (many of them latency-bound launches of a few microseconds) overlap the first scale's, forward and -- the
    autograd engine runs a node's backward on the stream of its forward -- backward."""
    if not (MSG_STREAMS and feats[0].is_cuda and len(mlps) > 1):
        return [_sa_scale(mlp, tg, lvl, s, feats, groups) for s, mlp in enumerate(mlps)]
    cur = torch.cuda.current_stream()
    outs = [None] * len(mlps)
    sides = []
    for s in range(1, len(mlps)):
        side = side_stream(feats[0].device, s)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            outs[s] = _sa_scale(mlps[s], tg, lvl, s, feats, groups)
        sides.append(side)
    outs[0] = _sa_scale(mlps[0], tg, lvl, 0, feats, groups)
    for s, side in enumerate(sides, 1):
        cur.wait_stream(side)
        outs[s].record_stream(cur)
    return outs


def pnhead_train(head, tg, features, groups=1):
    """PNHead.forward (model_utils.py:393-424) in training mode on geometry tg.  features (S_,Cf,n), or a list of tensors whose
    channel concatenation it is -> l0_points (S_,128,n).  groups: number of consecutive batch slices with their own BatchNorm
    statistics."""
    feats = list(features) if isinstance(features, (list, tuple)) else [features]
    levels = []
    for lvl, (sa, linear) in enumerate(((head.sa1, head.linear1), (head.sa2, head.linear2), (head.sa3, head.linear3))):
        outs = _msg_scales(sa.mlps, tg, lvl, feats, groups)
        # nn.Linear over the channel axis of the two scales' (virtually concatenated) outputs
        feats = [pw_linear(outs, linear.weight, linear.bias)]     # (S_, C, U)
        levels.append(feats[0])
    l1, l2, l3 = levels
    S, n = tg.npoint, tg.n
    l2 = _fp(head.fp3, tg, "fp3", l2, l3, tg.row_w[1], S, groups)
    l1 = _fp(head.fp2, tg, "fp2", l1, l2, tg.row_w[0], S, groups)
    # level 0 = the clouds' own points: in a padded batch the padding rows carry statistics weight 0, counts come from the device
    return _fp(head.fp1, tg, "fp1", None, l1, getattr(tg, "point_w", None), n, groups, group_counts=getattr(tg, "point_counts", None))


def correlator_supported(fc):
    return (not fc.bn and fc.nsample == 16 and abs(fc.slope - 0.1) < 1e-12 and len(fc.mlp_convs) == 3
            and all(c.out_channels == 256 for c in fc.mlp_convs) and not fc.weightnet1.bn and not fc.weightnet2.bn)


def _knn16(points, query, n_valid):
    """knn_point(16, points, query) (model_utils.py:85-99); n_valid (B,) int32: only the first n_valid[b] points are candidates."""
    if n_valid is None:
        from .model_utils import knn_point
        return knn_point(16, points, query).contiguous()
    B, S, _ = query.shape
    idx = torch.empty(B, S, 16, dtype=torch.int64, device=query.device)
    _lib.call("rtk_knn_point_masked", B, S, points.shape[1], 16, query.data_ptr(), points.data_ptr(), idx.data_ptr(), n_valid.data_ptr(),
              torch.cuda.current_stream().cuda_stream)
    return idx


def correlator_train(fc, pc1, pc2, feature1, feature2, n_valid1=None, n_valid2=None, xyz=None):
    """FeatureCorrelator.forward (model_utils.py:166-250) in training mode: the point-to-patch cost volume is one
    fused operator (forward kernel of the inference engine + its backward kernel), and so is the patch-to-patch
    aggregation.  pc (B,3,N), features (B,D,N) -> (B,256,N1); xyz: optional ((B,N1,3), (B,N2,3)) contiguous point-major copies of
    pc1, pc2.  n_valid1 / n_valid2 (B,) int32: padded batch -- kNN candidates
    are the valid points of the frame searched (padding queries are copies of their cloud's point 0 and get its result)."""
    B, C, N1 = pc1.shape
    # point-major coordinates: the caller's (TrainGeometry already holds both frames' as one contiguous tensor) or two transposed copies
    x1, x2 = xyz if xyz is not None else (pc1.permute(0, 2, 1).contiguous(), pc2.permute(0, 2, 1).contiguous())
    D1, D2 = feature1.shape[1], feature2.shape[1]
    knn = _knn16(x2, x1, n_valid2)
    conv0, conv1, conv2 = fc.mlp_convs
    w0 = conv0.weight.flatten(1)              # (a view whose backward is a view: `weight[:, :, 0, 0]` would cost two zeros + copy)
    # layer 1 of the cost-volume MLP split by input segment: per-point projections of both frames' features, written point-major
    p1 = pw_linear([feature1], conv0.weight, conv0.bias, cols=[0], out_point_major=True).permute(0, 2, 1).reshape(B * N1, 256)
    p2 = pw_linear([feature2], conv0.weight, None, cols=[D1], out_point_major=True).permute(0, 2, 1).reshape(-1, 256)
    wn = fc.weightnet1.mlp_convs
    x = cost_volume(p1, p2, w0[:, D1 + D2:], conv1.weight.flatten(1), conv1.bias, conv2.weight.flatten(1), conv2.bias,
                    wn[0].weight.flatten(1), wn[0].bias, wn[1].weight.flatten(1), wn[1].bias, wn[2].weight.flatten(1),
                    wn[2].bias, x1, x2, knn)                                   # (B*N1, 256) point-major
    knn = _knn16(x1, x1, n_valid1)
    wn = fc.weightnet2.mlp_convs
    x = patch_cost(x, wn[0].weight.flatten(1), wn[0].bias, wn[1].weight.flatten(1), wn[1].bias, wn[2].weight.flatten(1),
                   wn[2].bias, x1, knn, live=n_valid1)
    return x.view(B, N1, 256).permute(0, 2, 1)
