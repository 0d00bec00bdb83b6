"""Model building blocks of the RaTrack backbone: PNHead (3-level MSG PointNet++ encoder/decoder),
FeatureCorrelator (kNN cost volume), WeightNet, FlowDecoder, FlowPredictor, ClsPredictor.

Counterpart of the reference's utils/model_utils/model_utils.py (classes at :166-250, :253-305,
:308-357, :359-390, :393-424): same constructor arguments, forward signatures, tensor layouts and
state-dict keys, so a reference checkpoint loads unchanged.  Parameters that exist in a reference
checkpoint but are never used on the backbone path are kept as inert members so
that the key set matches.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pointnet2_utils as PU
from .pointnet2_modules import PointnetFPModule, PointnetSAModuleMSG

knn_point = PU.knn_point


def index_points(points, idx):
    """points (B,N,C), idx (B,S[,k]) int64 -> (B,S[,k],C).  model_utils.py:42-59."""
    B = points.shape[0]
    shape = [B] + [1] * (idx.dim() - 1)
    batch = torch.arange(B, dtype=torch.long, device=points.device).view(shape).expand_as(idx)
    return points[batch, idx, :]


class WeightNet(nn.Module):
    """3 -> 8 -> 8 -> C 1x1 convs, ReLU after every conv (the last one included), bn=False.
    model_utils.py:359-390.  `mlp_bns` exist in reference checkpoints but are never applied."""

    def __init__(self, in_channel, out_channel, hidden_unit=(8, 8), bn=False):
        super().__init__()
        self.bn = bn
        dims = [in_channel] + list(hidden_unit) + [out_channel]
        self.mlp_convs = nn.ModuleList(nn.Conv2d(dims[i], dims[i + 1], 1) for i in range(len(dims) - 1))
        self.mlp_bns = nn.ModuleList(nn.BatchNorm2d(dims[i + 1]) for i in range(len(dims) - 1))

    def forward(self, localized_xyz):
        w = localized_xyz
        for i, conv in enumerate(self.mlp_convs):
            w = conv(w)
            if self.bn:
                w = self.mlp_bns[i](w)
            w = F.relu(w)
        return w


class FeatureCorrelator(nn.Module):
    """kNN cost volume: point-to-patch (pc1 -> pc2) then patch-to-patch (pc1 -> pc1).
    model_utils.py:166-250.  pc (B,3,N), features (B,D,N) -> (B, mlp[-1], N1)."""

    project_first = True

    def __init__(self, nsample, in_channel, mlp, bn=False, use_leaky=True):
        super().__init__()
        self.nsample = nsample
        self.bn = bn
        self.mlp_convs = nn.ModuleList()
        if bn:
            self.mlp_bns = nn.ModuleList()
        last = in_channel
        for out in mlp:
            self.mlp_convs.append(nn.Conv2d(last, out, 1))
            if bn:
                self.mlp_bns.append(nn.BatchNorm2d(out))
            last = out
        self.cls_mlp = nn.Linear(16, 1)          # present in checkpoints, unused (model_utils.py:184)
        self.weightnet1 = WeightNet(3, last)
        self.weightnet2 = WeightNet(3, last)
        self.slope = 0.1 if use_leaky else 0.0

    def forward(self, pc1, pc2, feature1, feature2):
        B, C, N1 = pc1.shape
        pc1 = pc1.permute(0, 2, 1)
        pc2 = pc2.permute(0, 2, 1)
        feature1 = feature1.permute(0, 2, 1)
        feature2 = feature2.permute(0, 2, 1)
        D1 = feature1.shape[2]

        knn_idx = knn_point(self.nsample, pc2, pc1)                         # (B,N1,k)
        direction = index_points(pc2, knn_idx) - pc1.reshape(B, N1, 1, C)   # neighbour - query
        D2 = feature2.shape[2]
        if self.project_first:
            # conv0([f1 || f2[idx] || d]) = W1.f1 + (W2.f2)[idx] + Wd.d + b: project per point, then gather -- the reference's
            # (B, 515, k, N) input tensor (540 MB at B=64) and the 515->256 conv over N*k positions never exist
            conv0 = self.mlp_convs[0]
            w = conv0.weight[:, :, 0, 0]                                     # (Cout, D1+D2+3)
            p1 = F.linear(feature1, w[:, :D1], conv0.bias)                   # (B,N1,Cout)
            p2 = F.linear(feature2, w[:, D1:D1 + D2])                        # (B,N2,Cout)
            x = p1.unsqueeze(2) + index_points(p2, knn_idx) + F.linear(direction, w[:, D1 + D2:])     # (B,N1,k,Cout)
            x = x.permute(0, 3, 2, 1)                                        # (B,Cout,k,N1)
            if self.bn:
                x = self.mlp_bns[0](x)
            x = F.leaky_relu(x, self.slope)
            rest = list(enumerate(self.mlp_convs))[1:]
        else:
            grouped2 = index_points(feature2, knn_idx)
            grouped1 = feature1.reshape(B, N1, 1, D1).expand(-1, -1, self.nsample, -1)
            x = torch.cat([grouped1, grouped2, direction], dim=-1).permute(0, 3, 2, 1)   # (B, D1+D2+3, k, N1)
            rest = list(enumerate(self.mlp_convs))
        for i, conv in rest:
            x = conv(x)
            if self.bn:
                x = self.mlp_bns[i](x)
            x = F.leaky_relu(x, self.slope)
        w = self.weightnet1(direction.permute(0, 3, 2, 1))
        x = torch.sum(w * x, dim=2)                                         # (B,C,N1)

        knn_idx = knn_point(self.nsample, pc1, pc1)
        direction = index_points(pc1, knn_idx) - pc1.reshape(B, N1, 1, C)
        w = self.weightnet2(direction.permute(0, 3, 2, 1))
        x = index_points(x.permute(0, 2, 1), knn_idx).permute(0, 3, 2, 1)
        return torch.sum(w * x, dim=2)


class _Predictor(nn.Module):
    def __init__(self, in_channel, mlp):
        super().__init__()
        self.sf_mlp = nn.ModuleList()
        last = in_channel
        for out in mlp:
            self.sf_mlp.append(nn.Sequential(nn.Conv2d(last, out, 1, bias=False), nn.BatchNorm2d(out), nn.ReLU(inplace=False)))
            last = out
        self.conv2 = nn.Conv2d(mlp[-1], 3, 1, bias=False)

    def _trunk(self, feat, point_w=None, point_counts=None):
        """point_w (B,N) / point_counts (1,) device float64: statistics weights and element count of a padded batch
        (train_path.TrainGeometry); training path only."""
        x = feat.unsqueeze(3)
        if self.training and feat.is_cuda and torch.is_grad_enabled():
            # training on the GPU: the per-point layer operators (train_ops.pw_bn_relu / pw_linear: conv + batch statistics in one
            # kernel, hand-written backward); same parameters, statistics and running-stat updates as the nn.Sequential blocks
            from .train_ops import pw_bn_relu, pw_linear
            x = feat
            for block in self.sf_mlp:
                x = pw_bn_relu([x], block[0].weight, block[1], row_weight=point_w, group_counts=point_counts)
            return pw_linear([x], self.conv2.weight)
        assert point_w is None, "padded batches train on the HIP training path only"
        for block in self.sf_mlp:
            x = block(x)
        return self.conv2(x).squeeze(3)


class FlowPredictor(_Predictor):
    """(B,C,N) -> (B,3,N) scene flow.  model_utils.py:308-329."""

    def forward(self, feat, point_w=None, point_counts=None):
        return self._trunk(feat, point_w, point_counts)


class ClsPredictor(_Predictor):
    """(B,C,N) -> (B,N) moving-point probability.  model_utils.py:332-357."""

    def __init__(self, in_channel, mlp):
        super().__init__(in_channel, mlp)
        self.linear = nn.Linear(3, 1)

    def forward(self, feat, point_w=None, point_counts=None):
        t = self._trunk(feat, point_w, point_counts)
        if self.training and t.is_cuda and torch.is_grad_enabled():
            from .train_ops import pw_linear
            return torch.sigmoid(pw_linear([t], self.linear.weight, self.linear.bias)).squeeze(1)     # Linear(3,1) over the channel axis
        x = self.linear(t.permute(0, 2, 1))
        return torch.sigmoid(x).squeeze(2)


class PNHead(nn.Module):
    """sa1 -> linear1 -> sa2 -> linear2 -> sa3 -> linear3 -> fp3 -> fp2 -> fp1.
    model_utils.py:393-424.  pc (B,N,3), features (B,Cf,N), in_channels = Cf + 3
    -> (l3_xyz (B,S,3), l0_points (B,128,N))."""

    def __init__(self, sample_point_num, in_channels):
        super().__init__()
        S, C = sample_point_num, in_channels
        self.sa1 = PointnetSAModuleMSG(npoint=S, radii=[2, 4], nsamples=[4, 8], mlps=[[C, 16, 16, 32], [C, 16, 16, 32]])
        self.sa2 = PointnetSAModuleMSG(npoint=S, radii=[4, 8], nsamples=[8, 16], mlps=[[3 + 32, 32, 32], [3 + 32, 32, 64]])
        self.sa3 = PointnetSAModuleMSG(npoint=S, radii=[8, 16], nsamples=[16, 32], mlps=[[3 + 64, 64, 64], [3 + 64, 64, 64]])
        self.fp3 = PointnetFPModule(mlp=[128, 128])
        self.fp2 = PointnetFPModule(mlp=[160, 128])
        self.fp1 = PointnetFPModule(mlp=[128, 128])
        self.linear1 = nn.Linear(64, 32)
        self.linear2 = nn.Linear(96, 64)
        self.linear3 = nn.Linear(128, 64)

    @staticmethod
    def _lin(layer, x):
        return layer(x.permute(0, 2, 1)).permute(0, 2, 1).contiguous()

    def forward(self, pc, features):
        l0_points, l0_xyz = features.contiguous(), pc.contiguous()
        l1_xyz, l1_points = self.sa1(l0_xyz, l0_points)
        l1_points = self._lin(self.linear1, l1_points)
        l2_xyz, l2_points = self.sa2(l1_xyz, l1_points)
        l2_points = self._lin(self.linear2, l2_points)
        l3_xyz, l3_points = self.sa3(l2_xyz, l2_points)
        l3_points = self._lin(self.linear3, l3_points)
        l2_points = self.fp3(l2_xyz, l3_xyz, l2_points, l3_points)
        l1_points = self.fp2(l1_xyz, l2_xyz, l1_points, l2_points)
        l0_points = self.fp1(l0_xyz, l1_xyz, None, l1_points)
        return l3_xyz, l0_points


class _UnusedPointGRU(nn.Module):
    """Key-compatible shell of `fd_layer.pnnGru` (reference: models/utils/flowstep3d.py:135-152,
    instantiated at model_utils.py:278, forward never called): 3 x {Conv2d(259,128,1,no bias), BN}."""

    class _Gate(nn.Module):
        def __init__(self, cin, cout):
            super().__init__()
            self.mlp_convs = nn.ModuleList([nn.Conv2d(cin, cout, 1, bias=False)])
            self.mlp_bns = nn.ModuleList([nn.BatchNorm2d(cout)])

    def __init__(self, hidden_dim, input_dim):
        super().__init__()
        cin = hidden_dim + input_dim + 3
        self.convz = self._Gate(cin, hidden_dim)
        self.convr = self._Gate(cin, hidden_dim)
        self.convq = self._Gate(cin, hidden_dim)


class FlowDecoder(nn.Module):
    """cls head on the cost volume; multi-scale embedding propagation (PNHead over 514 channels);
    global max -> 5-layer GRU step -> flow head.  model_utils.py:253-305."""

    def __init__(self, fc_inch, args):
        super().__init__()
        ep_inch = fc_inch * 2 + 5
        sf_inch = 4 * int(fc_inch / 8) * 2
        sf_mlps = [int(sf_inch / 2), int(sf_inch / 4), int(sf_inch / 8)]
        self.mse = PNHead(args.npoints, ep_inch)
        self.fp = FlowPredictor(in_channel=sf_inch, mlp=sf_mlps)
        self.cp = ClsPredictor(in_channel=sf_inch, mlp=sf_mlps)
        self.mlp2 = nn.ModuleList([nn.Linear(3, 1)])                          # unused (model_utils.py:269-274)
        self.gru2 = nn.GRU(input_size=fc_inch, hidden_size=fc_inch)           # unused (:276)
        self.pnnGru = _UnusedPointGRU(fc_inch // 2, fc_inch // 2)             # unused (:278)
        self.torchGRU = nn.GRU(fc_inch // 2, fc_inch // 2, 5)

    def forward(self, pc1, feature1, pc1_features, cor_features, h, train_geo=None):
        """train_geo: optional train_path.TrainGeometry of pc1 (training mode): the 514-channel PNHead then runs on
        the de-duplicated levels."""
        pw, pcnt = (getattr(train_geo, "point_w", None), getattr(train_geo, "point_counts", None)) if train_geo is not None else (None, None)
        cls = self.cp(cor_features, pw, pcnt)
        parts = (feature1, pc1_features, cor_features) if feature1 is not None else (pc1_features, cor_features)
        if train_geo is not None:
            from .train_path import pnhead_train
            prop = pnhead_train(self.mse, train_geo, list(parts))          # the 514-channel concatenation stays virtual
        else:
            _, prop = self.mse(pc1.permute(0, 2, 1).contiguous(), torch.cat(parts, dim=1))
        gfeat = torch.max(prop, -1)[0].unsqueeze(2)
        if h is None:   # the reference hard-wires (5,1,128) (model_utils.py:294-295); batch-general here
            h = torch.zeros(5, prop.size(0), 128, device=prop.device, dtype=prop.dtype)
        if self.training and prop.is_cuda and torch.is_grad_enabled():
            from .train_ops import gru_step                      # one forward + one backward kernel instead of MIOpen's RNN
            y, h = gru_step(gfeat.squeeze(2), h, self.torchGRU)
            gfeat = y.unsqueeze(2).expand(prop.size(0), prop.size(1), pc1.size(2))
        else:
            gfeat, h = self.torchGRU(gfeat.permute(2, 0, 1), h)
            gfeat = gfeat.permute(1, 2, 0).expand(prop.size(0), prop.size(1), pc1.size(2))
        output = self.fp(torch.cat((prop, gfeat), dim=1), pw, pcnt)
        return output, h, prop, cls
