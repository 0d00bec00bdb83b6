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

    def forward(self, xyz, features=None, new_xyz=None):
        """xyz (B,N,3), features (B,C,N) -> new_xyz (B,npoint,3), (B, sum_k mlps[k][-1], npoint)."""
        if new_xyz is None and self.npoint is not None:
            flipped = xyz.transpose(1, 2).contiguous()
            idx = PU.furthest_point_sample(xyz, self.npoint)
            new_xyz = PU.gather_operation(flipped, idx).transpose(1, 2).contiguous()
        outs = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            x = mlp(grouper(xyz, new_xyz, features))          # (B, C', npoint, nsample)
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
