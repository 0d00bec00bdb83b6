"""Set-abstraction (MSG) and feature-propagation modules: counterparts of the reference's
lib/pointnet2_modules.py (PointnetSAModuleMSG :58-94, PointnetFPModule :118-158) with identical
constructor keywords, forward signatures and state-dict keys, on top of the gfx950 ops."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pointnet2_utils as PU
from .pytorch_utils import SharedMLP


class PointnetSAModuleMSG(nn.Module):
    """FPS -> centroids; per scale: ball query + group -> SharedMLP -> max over the neighbourhood;
    scales concatenated on the channel axis.  NOTE the reference does NOT add 3 to mlp[0]
    (lib/pointnet2_modules.py:88-91): callers pass the full input width."""

    def __init__(self, *, npoint, radii, nsamples, mlps, bn=True, use_xyz=True, pool_method="max_pool"):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.npoint = npoint
        self.pool_method = pool_method
        self.groupers = nn.ModuleList(
            PU.QueryAndGroup(r, ns, use_xyz=use_xyz) if npoint is not None else PU.GroupAll(use_xyz)
            for r, ns in zip(radii, nsamples))
        self.mlps = nn.ModuleList(SharedMLP(list(spec), bn=bn) for spec in mlps)

    project_first = True

    @staticmethod
    def _project_then_group(grouper, mlp, xyz, new_xyz, features):
        """Same value as mlp(grouper(xyz, new_xyz, features)) with the first 1x1 conv moved in front of the gather:
        conv([d_xyz || feats[idx]]) = Wx.d_xyz + (Wf.feats)[idx]  (a 1x1 conv and a gather commute).  The reference
        materialises the grouped (B, 3+C, npoint, nsample) tensor -- 813 MB for the decoder's 514-channel level at B=64 --
        convolves it and back-propagates through it; here only the C1-channel projection is gathered (and scattered in
        the backward).  Parameters, BatchNorm statistics and results are unchanged up to fp32 summation order."""
        layer0 = mlp.layer0
        w = layer0.conv.weight                                                    # (C1, 3+C, 1, 1), no bias (bn=True)
        c1 = w.shape[0]
        idx = PU.ball_query(grouper.radius, grouper.nsample, xyz, new_xyz)
        grouped_xyz = PU.grouping_operation(xyz.transpose(1, 2).contiguous(), idx) - new_xyz.transpose(1, 2).unsqueeze(-1)
        proj = F.conv1d(features, w[:, 3:, 0, :])                                 # (B, C1, n): per-POINT projection
        x = F.conv2d(grouped_xyz, w[:, :3]) + PU.grouping_operation(proj.contiguous(), idx)
        if layer0.conv.bias is not None:
            x = x + layer0.conv.bias.view(1, c1, 1, 1)
        for name, mod in layer0.named_children():                                 # bn / activation of layer 0
            if name != "conv":
                x = mod(x)
        for name, layer in mlp.named_children():
            if name != "layer0":
                x = layer(x)
        return x

    def forward(self, xyz, features=None, new_xyz=None):
        """xyz (B,N,3), features (B,C,N) -> new_xyz (B,npoint,3), (B, sum_k mlps[k][-1], npoint)."""
        if new_xyz is None and self.npoint is not None:
            flipped = xyz.transpose(1, 2).contiguous()
            idx = PU.furthest_point_sample(xyz, self.npoint)
            new_xyz = PU.gather_operation(flipped, idx).transpose(1, 2).contiguous()
        outs = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            if self.project_first and isinstance(grouper, PU.QueryAndGroup) and grouper.use_xyz and features is not None:
                x = self._project_then_group(grouper, mlp, xyz, new_xyz, features)
            else:
                x = mlp(grouper(xyz, new_xyz, features))      # (B, C', npoint, nsample)
            if self.pool_method == "max_pool":
                x = F.max_pool2d(x, kernel_size=[1, x.size(3)])
            elif self.pool_method == "avg_pool":
                x = F.avg_pool2d(x, kernel_size=[1, x.size(3)])
            else:
                raise NotImplementedError(self.pool_method)
            outs.append(x.squeeze(-1))
        return new_xyz, torch.cat(outs, dim=1)


class PointnetSAModule(PointnetSAModuleMSG):
    def __init__(self, *, mlp, npoint=None, radius=None, nsample=None, bn=True, use_xyz=True, pool_method="max_pool"):
        super().__init__(mlps=[mlp], npoint=npoint, radii=[radius], nsamples=[nsample], bn=bn, use_xyz=use_xyz,
                         pool_method=pool_method)


class PointnetFPModule(nn.Module):
    """three_nn inverse-distance interpolation of `known_feats` onto `unknown`, concatenated with the
    skip features, then a SharedMLP.  lib/pointnet2_modules.py:129-158."""

    def __init__(self, *, mlp, bn=True):
        super().__init__()
        self.mlp = SharedMLP(list(mlp), bn=bn)

    def forward(self, unknown, known, unknow_feats, known_feats):
        if known is not None:
            dist, idx = PU.three_nn(unknown, known)
            recip = 1.0 / (dist + 1e-8)
            weight = recip / torch.sum(recip, dim=2, keepdim=True)
            interpolated = PU.three_interpolate(known_feats, idx, weight)
        else:
            interpolated = known_feats.expand(*known_feats.size()[0:2], unknown.size(1))
        x = torch.cat([interpolated, unknow_feats], dim=1) if unknow_feats is not None else interpolated
        return self.mlp(x.unsqueeze(-1)).squeeze(-1)
