"""CPU restatement of RaTrack's backbone graph (Track4D.backbone and everything under it).

TEST INFRASTRUCTURE ONLY -- the checker for tests/, __graft_entry__.smoke() and the timed
`cpu_baseline` of bench.py.  Nothing under ratrack_amd/ imports this file.

Functional style on purpose (plain functions over a state-dict of CPU tensors, no nn.Module): it is
an independent second statement of the graph, pinned against the golden vectors that
tools/make_golden.py captured from the reference's own Python (tests/test_oracle_golden.py).
Native ops come from oracle/pointnet2_ref.c; dense layers are PyTorch-CPU fp32 ops, exactly what
the reference graph itself dispatches to.  All file:line citations are into /root/reference/src.
"""
import torch
import torch.nn.functional as F

from . import pointnet2_ref as P

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# ---- autograd wrappers (lib/pointnet2_utils.py:184-225, 136-181): grads w.r.t. features only ----

class _Group(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, idx):
        ctx.save_for_backward(idx)
        ctx.n = features.shape[2]
        return P.group(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        B, C, M, ns = grad_out.shape
        g = torch.zeros(B, C, ctx.n, dtype=torch.float32)
        P.group_points_grad_wrapper(B, C, ctx.n, M, ns, grad_out.contiguous(), idx, g)
        return g, None


class _Interp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, idx, weight):
        ctx.save_for_backward(idx, weight)
        ctx.m = features.shape[2]
        return P.three_interpolate(features, idx, weight)

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        B, c, n = grad_out.shape
        g = torch.zeros(B, c, ctx.m, dtype=torch.float32)
        P.three_interpolate_grad_wrapper(B, c, n, ctx.m, grad_out.contiguous(), idx, weight.contiguous(), g)
        return g, None, None


# ---- float64 arbiter mode -------------------------------------------------------------------------
# Feed float64 weights / features and the graph runs in float64: the INDEX-producing ops (FPS, ball query, three-NN, kNN) still
# see the fp32 coordinates (coordinates are fp32 values, the casts are exact), so the geometry is that of the fp32 run and the
# result arbitrates between two fp32 implementations (tests: "which side does the rounding sit on").  Value ops (group,
# interpolate) become differentiable torch gathers.

def _f32(t):
    return t if t.dtype == torch.float32 else t.float()


def _group_any(features, idx):
    if features.dtype == torch.float32:
        return _Group.apply(features.contiguous(), idx)
    B, C, _ = features.shape
    _, M, ns = idx.shape
    return torch.gather(features, 2, idx.long().view(B, 1, M * ns).expand(-1, C, -1)).view(B, C, M, ns)


def _interp_any(feats, idx, weight):
    if feats.dtype == torch.float32:
        return _Interp.apply(feats.contiguous(), idx, weight)
    B, C, _ = feats.shape
    n = idx.shape[1]
    g = torch.gather(feats, 2, idx.long().view(B, 1, n * 3).expand(-1, C, -1)).view(B, C, n, 3)
    return (g * weight.unsqueeze(1)).sum(-1)


# ---- building blocks ------------------------------------------------------------------------------

def _bn(x, sd, p, training):
    """nn.BatchNorm2d semantics incl. running-stat update (momentum 0.1, unbiased var) in train mode."""
    if training and (p + ".num_batches_tracked") in sd:
        sd[p + ".num_batches_tracked"] += 1
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        training, BN_MOMENTUM, BN_EPS)


def shared_mlp(x, sd, prefix, training):
    """lib/pytorch_utils.py:5-32: [Conv2d 1x1 (no bias) -> BatchNorm2d -> ReLU] x L on (B,C,S,ns)."""
    i = 0
    while (prefix + ".layer%d.conv.weight" % i) in sd:
        x = F.conv2d(x, sd[prefix + ".layer%d.conv.weight" % i])
        x = _bn(x, sd, prefix + ".layer%d.bn.bn" % i, training)
        x = F.relu(x)
        i += 1
    return x


def query_and_group(radius, nsample, xyz, new_xyz, features, trace=None):
    """lib/pointnet2_utils.py:269-292: [grouped_xyz - centroid (3) || grouped features (C)]."""
    idx = P.ball_query(radius, nsample, _f32(xyz), _f32(new_xyz))
    if trace is not None:
        trace.setdefault("ball_idx", []).append(idx)
    xyz_trans = xyz.transpose(1, 2).contiguous()
    grouped_xyz = P.group(xyz_trans, idx) if xyz.dtype == torch.float32 else _group_any(xyz_trans, idx)
    grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
    grouped_features = _group_any(features.contiguous(), idx)
    return torch.cat([grouped_xyz, grouped_features], dim=1)


def sa_module_msg(sd, prefix, xyz, features, npoint, radii, nsamples, training, trace=None):
    """lib/pointnet2_modules.py:19-55 (PointnetSAModuleMSG, max_pool)."""
    xyz_flipped = xyz.transpose(1, 2).contiguous()
    fps_idx = P.fps(_f32(xyz), npoint)
    if trace is not None:
        trace.setdefault("fps_idx", []).append(fps_idx)
    new_xyz = P.gather(_f32(xyz_flipped), fps_idx).transpose(1, 2).contiguous().to(xyz.dtype)
    outs = []
    for i, (r, ns) in enumerate(zip(radii, nsamples)):
        g = query_and_group(r, ns, xyz, new_xyz, features, trace)
        g = shared_mlp(g, sd, "%s.mlps.%d" % (prefix, i), training)
        g = F.max_pool2d(g, kernel_size=[1, g.size(3)]).squeeze(-1)
        outs.append(g)
    return new_xyz, torch.cat(outs, dim=1)


def fp_module(sd, prefix, unknown, known, unknown_feats, known_feats, training, trace=None):
    """lib/pointnet2_modules.py:129-158 (PointnetFPModule)."""
    d2, idx = P.three_nn(_f32(unknown), _f32(known))
    if trace is not None:
        trace.setdefault("three_nn", []).append((d2, idx))
    dist = torch.sqrt(d2.to(unknown.dtype))
    dist_recip = 1.0 / (dist + 1e-8)
    norm = torch.sum(dist_recip, dim=2, keepdim=True)
    weight = dist_recip / norm
    interp = _interp_any(known_feats.contiguous(), idx, weight)
    x = torch.cat([interp, unknown_feats], dim=1) if unknown_feats is not None else interp
    x = shared_mlp(x.unsqueeze(-1), sd, prefix + ".mlp", training)
    return x.squeeze(-1)


def pn_head(sd, prefix, pc, features, npoint, training, trace=None):
    """utils/model_utils/model_utils.py:393-424 (PNHead).  pc (B,N,3), features (B,Cf,N)."""
    l0_points = features.contiguous()
    l0_xyz = pc.contiguous()
    lin = lambda x, name: F.linear(x.permute(0, 2, 1), sd[prefix + name + ".weight"], sd[prefix + name + ".bias"]) \
        .permute(0, 2, 1).contiguous()
    l1_xyz, l1_points = sa_module_msg(sd, prefix + "sa1", l0_xyz, l0_points, npoint, [2, 4], [4, 8], training, trace)
    if trace is not None:
        trace.setdefault("acts", {})[prefix + "sa1"] = l1_points
    l1_points = lin(l1_points, "linear1")
    l2_xyz, l2_points = sa_module_msg(sd, prefix + "sa2", l1_xyz, l1_points, npoint, [4, 8], [8, 16], training, trace)
    if trace is not None:
        trace["acts"][prefix + "sa2"] = l2_points
    l2_points = lin(l2_points, "linear2")
    l3_xyz, l3_points = sa_module_msg(sd, prefix + "sa3", l2_xyz, l2_points, npoint, [8, 16], [16, 32], training, trace)
    if trace is not None:
        trace["acts"][prefix + "sa3"] = l3_points
    l3_points = lin(l3_points, "linear3")
    l2_points = fp_module(sd, prefix + "fp3", l2_xyz, l3_xyz, l2_points, l3_points, training, trace)
    l1_points = fp_module(sd, prefix + "fp2", l1_xyz, l2_xyz, l1_points, l2_points, training, trace)
    l0_points = fp_module(sd, prefix + "fp1", l0_xyz, l1_xyz, None, l1_points, training, trace)
    if trace is not None:
        trace["acts"][prefix + "fp1"] = l0_points
    return l3_xyz, l0_points


def index_points(points, idx):
    """model_utils.py:42-59: points (B,N,C), idx (B,S,k) -> (B,S,k,C)."""
    B = points.shape[0]
    bidx = torch.arange(B).view(B, 1, 1).expand_as(idx)
    return points[bidx, idx, :]


def weight_net(sd, prefix, xyz):
    """model_utils.py:359-390 (WeightNet, bn=False): ReLU after every conv including the last."""
    w = xyz
    for i in range(3):
        w = F.relu(F.conv2d(w, sd["%s.mlp_convs.%d.weight" % (prefix, i)], sd["%s.mlp_convs.%d.bias" % (prefix, i)]))
    return w


def feature_correlator(sd, prefix, pc1, pc2, feature1, feature2, nsample=16, trace=None):
    """model_utils.py:193-250 (FeatureCorrelator.forward), bn=False, LeakyReLU(0.1)."""
    B, C, N1 = pc1.shape
    pc1 = pc1.permute(0, 2, 1).contiguous()
    pc2 = pc2.permute(0, 2, 1).contiguous()
    feature1 = feature1.permute(0, 2, 1)
    feature2 = feature2.permute(0, 2, 1)
    D1 = feature1.shape[2]
    # point-to-patch volume
    knn_idx = P.knn_point(nsample, _f32(pc2), _f32(pc1))
    if trace is not None:
        trace.setdefault("knn_idx", []).append(knn_idx)
    neighbor_xyz = index_points(pc2, knn_idx)
    direction_xyz = neighbor_xyz - pc1.reshape(B, N1, 1, C)
    grouped_feature2 = index_points(feature2, knn_idx)
    grouped_feature1 = feature1.reshape(B, N1, 1, D1).repeat(1, 1, nsample, 1)
    x = torch.cat([grouped_feature1, grouped_feature2, direction_xyz], dim=-1).permute(0, 3, 2, 1)
    for i in range(3):
        x = F.leaky_relu(F.conv2d(x, sd["%s.mlp_convs.%d.weight" % (prefix, i)], sd["%s.mlp_convs.%d.bias" % (prefix, i)]), 0.1)
    weights = weight_net(sd, prefix + ".weightnet1", direction_xyz.permute(0, 3, 2, 1))
    x = torch.sum(weights * x, dim=2)  # (B, C, N)
    # patch-to-patch cost
    knn_idx = P.knn_point(nsample, _f32(pc1), _f32(pc1))
    if trace is not None:
        trace["knn_idx"].append(knn_idx)
    neighbor_xyz = index_points(pc1, knn_idx)
    direction_xyz = neighbor_xyz - pc1.view(B, N1, 1, C)
    weights = weight_net(sd, prefix + ".weightnet2", direction_xyz.permute(0, 3, 2, 1))
    x = index_points(x.permute(0, 2, 1), knn_idx)
    x = weights * x.permute(0, 3, 2, 1)
    return torch.sum(x, dim=2)


def _predictor(sd, prefix, feat, training):
    """model_utils.py:308-357: [Conv2d(no bias) -> BN -> ReLU] x 3 -> conv2 (no bias)."""
    x = feat.unsqueeze(3)
    for i in range(3):
        x = F.conv2d(x, sd["%s.sf_mlp.%d.0.weight" % (prefix, i)])
        x = _bn(x, sd, "%s.sf_mlp.%d.1" % (prefix, i), training)
        x = F.relu(x)
    return F.conv2d(x, sd[prefix + ".conv2.weight"]).squeeze(3)


def gru_step(sd, prefix, x, h, num_layers=5):
    """nn.GRU(128,128,5) on a length-1 sequence (model_utils.py:279,296): gate order (r,z,n)."""
    h_out = []
    inp = x
    for l in range(num_layers):
        gi = F.linear(inp, sd["%s.weight_ih_l%d" % (prefix, l)], sd["%s.bias_ih_l%d" % (prefix, l)])
        gh = F.linear(h[l], sd["%s.weight_hh_l%d" % (prefix, l)], sd["%s.bias_hh_l%d" % (prefix, l)])
        i_r, i_z, i_n = gi.chunk(3, 1)
        h_r, h_z, h_n = gh.chunk(3, 1)
        r = torch.sigmoid(i_r + h_r)
        z = torch.sigmoid(i_z + h_z)
        n = torch.tanh(i_n + r * h_n)
        hn = (1 - z) * n + z * h[l]
        h_out.append(hn)
        inp = hn
    return inp, torch.stack(h_out, 0)


def flow_decoder(sd, prefix, pc1, feature1, pc1_features, cor_features, h, npoint, training, trace=None):
    """model_utils.py:281-305 (FlowDecoder.forward)."""
    cls = _predictor(sd, prefix + ".cp", cor_features, training)
    cls = F.linear(cls.permute(0, 2, 1), sd[prefix + ".cp.linear.weight"], sd[prefix + ".cp.linear.bias"])
    cls = torch.sigmoid(cls).squeeze(2)
    embeddings = torch.cat((feature1, pc1_features, cor_features), dim=1)
    _, prop = pn_head(sd, prefix + ".mse.", pc1.permute(0, 2, 1).contiguous(), embeddings, npoint, training, trace)
    gfeat = torch.max(prop, -1)[0]  # (B,128)
    g, h = gru_step(sd, prefix + ".torchGRU", gfeat, h)
    g = g.unsqueeze(2).expand(prop.size(0), prop.size(1), pc1.size(2))
    new_features = torch.cat((prop, g), dim=1)
    output = _predictor(sd, prefix + ".fp", new_features, training)
    return output, h, prop, cls


def backbone(sd, pc1, pc2, feature1, feature2, h, npoint=512, training=False, trace=None):
    """models/track4d.py:67-106 (Track4D.backbone).  Returns the reference's 7-tuple
    (flow, h, cls, cor_features, pc1_features, pc2_features, prop_features)."""
    B = pc1.shape[0]
    if h is None:
        h = torch.zeros(5, B, 128, dtype=feature1.dtype)
    _, f1 = pn_head(sd, "pn_head.", pc1.permute(0, 2, 1).contiguous(), feature1, npoint, training, trace)
    _, f2 = pn_head(sd, "pn_head.", pc2.permute(0, 2, 1).contiguous(), feature2, npoint, training,
                    None if trace is None else trace.setdefault("pc2", {}))
    g1 = torch.max(f1, -1)[0].unsqueeze(2).expand(-1, -1, pc1.size(2))
    g2 = torch.max(f2, -1)[0].unsqueeze(2).expand(-1, -1, pc2.size(2))
    pc1_features = torch.cat((f1, g1), dim=1)
    pc2_features = torch.cat((f2, g2), dim=1)
    cor = feature_correlator(sd, "fc_layer", pc1, pc2, pc1_features, pc2_features, 16, trace)
    output, h, prop, cls = flow_decoder(sd, "fd_layer", pc1, feature1, pc1_features, cor, h, npoint, training,
                                        None if trace is None else trace.setdefault("mse", {}))
    return output, h, cls, cor, pc1_features, pc2_features, prop


# ---- losses (losses/loss.py) and metrics (main_utils.py:342-389) ----------------------------------

def flow_loss(pc1_warp, gt_flow):
    """losses/loss.py:85-89: mean_N ||pc1_warp - gt||_2 of batch element 0 only."""
    sc = ((pc1_warp - gt_flow).pow(2).sum(dim=1)).sqrt()
    return torch.mean(sc, dim=1)[0]


def motion_seg_loss(pred_cls, gt_cls):
    """losses/loss.py:124-146: 0.4*BCE(pos) + 0.6*BCE(neg); gt_cls bool (N,), pred (1,N)."""
    t, f = gt_cls == True, gt_cls == False  # noqa: E712
    g = gt_cls.to(pred_cls.dtype).unsqueeze(0)
    bce = torch.nn.BCELoss(reduction="mean")
    return 0.4 * bce(pred_cls[:, t], g[:, t]) + 0.6 * bce(pred_cls[:, f], g[:, f])


def track_4d_loss(pc1_warp, cls, gt_flow, gt_cls, pretrain=False):
    """losses/loss.py:8-31 with no tracked objects (affinity_loss of empty mappings = 0, :70-71)."""
    sf = flow_loss(pc1_warp, gt_flow)
    trk = torch.tensor(0)
    seg = motion_seg_loss(cls, gt_cls)
    if sf.isnan():
        sf = torch.tensor(0, dtype=torch.float32, requires_grad=True)
    if seg.isnan():
        seg = torch.tensor(0, dtype=torch.float32, requires_grad=True)
    total = 0.5 * sf + 0.5 * trk + 1 * seg
    if pretrain:
        total = 0 + 0 + seg
    return total, {"Loss": total, "SceneFlowLoss": sf, "TrackingLoss": trk, "SegLoss": seg}


def epe(pc1_warp, gt_warp):
    """main_utils.py:348-349: mean sqrt(sum_xyz (pred-gt)^2 + 1e-20)."""
    return torch.sqrt(((pc1_warp - gt_warp) ** 2).sum(1) + 1e-20).mean()
