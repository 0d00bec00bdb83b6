"""Track4D: the RaTrack model top (reference: models/track4d.py:13-106) with its Python API kept --
`Track4D(args)`, `backbone(pc1, pc2, feature1, feature2, h)` -> 7-tuple, `forward(...)` -> 10-tuple,
reference state-dict keys -- on top of the gfx950 kernels.

Scope: `backbone()` is the hot path and is batch-general.  The post-backbone half of the
reference's forward (moving-point clustering, Affinity MLP, log-Sinkhorn association, models/track4d.py:56-63,
108-223) lives in ratrack_amd/association.py and is B = 1 logic, as in the reference.
"""
import torch
import torch.nn as nn

from . import association as A
from .model_utils import FeatureCorrelator, FlowDecoder, PNHead

class Affinity(nn.Module):
    """Object-pair affinity MLP (models/track4d.py:226-246).  Not on the backbone path; kept so that
    reference checkpoints load with identical keys."""

    def __init__(self, emb_dims=137):
        super().__init__()
        d = emb_dims
        self.affinity = nn.Sequential(nn.Linear(d, d * 4), nn.ReLU(), nn.Linear(d * 4, d * 2), nn.ReLU(),
                                      nn.Linear(d * 2, d // 2), nn.ReLU(), nn.Linear(d // 2, d // 4), nn.ReLU(),
                                      nn.Linear(d // 4, 1), nn.Sigmoid())

    def forward(self, src_embedding, tgt_embedding):
        return self.affinity((src_embedding - tgt_embedding)[0])


class Track4D(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.rigid_thres = args.rigid_thres
        self.rigid_pcs = 0.25
        self.npoints = args.num_points          # stored, unused -- as in the reference
        fc_inch = 2 * 128
        self.pn_head = PNHead(args.npoints, 5)
        self.fc_layer = FeatureCorrelator(16, in_channel=fc_inch * 2 + 3, mlp=[fc_inch, fc_inch, fc_inch])
        self.fd_layer = FlowDecoder(fc_inch=fc_inch, args=args)
        self.affinity = Affinity(141)
        self.register_parameter("bin_score", nn.Parameter(torch.tensor(1.)))
        self.min_obj_points = args.min_obj_points
        self.associator = A.Associator(self.affinity)
        self._fused = None     # lazily built fused inference engine (ratrack_amd.fused)
        self._fused_version = -1
        self._fused_tensors = None
        self._use_fused = True
        self._dedup_train = True   # training mode: PNHead on de-duplicated levels with the HIP BatchNorm operators (train_path.py)

    # ---- hot path ---------------------------------------------------------------------------------
    def backbone(self, pc1, pc2, feature1, feature2, h, n_valid=None):
        """pc (B,3,N), feature (B,2,N) = (RCS, v_r), h (5,B,128) or None ->
        (flow (B,3,N), h, cls (B,N), cor_features (B,256,N), pc1_features (B,256,N),
         pc2_features (B,256,N), prop_features (B,128,N)).  models/track4d.py:67-106.
        n_valid (2,B) int32 (optional): a padded batch of clouds of different sizes (vod_gt.pad_frame_pairs; the reference
        itself only runs B = 1): every sample's valid columns equal its own unpadded B = 1 result."""
        # eval mode runs the fused inference engine -- also with autograd enabled (the reference's evaluation loop calls
        # net.eval() but never enters torch.no_grad(), main_utils.py:44-127), as long as no INPUT asks for a gradient; the
        # outputs then carry no graph.  (An input that requires a gradient takes the module path below by itself; `_use_fused` is a test hook.)
        wants_graph = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (pc1, pc2, feature1, feature2, h))
        if self._use_fused and not self.training and not wants_graph and pc1.is_cuda:
            eng = self._fused_engine()
            N1, N2 = pc1.shape[2], pc2.shape[2]
            if N1 == N2:
                return eng.backbone(pc1, pc2, feature1, feature2, h, n_valid=n_valid)
            # consecutive real frames differ in size: pad the smaller cloud with copies of its point 0 (exact, see
            # vod_gt.pad_frame_pairs) and cut the outputs back
            assert n_valid is None, "n_valid batches are already padded to a common size"
            B, Nm = pc1.shape[0], max(N1, N2)
            pad = lambda t: t if t.shape[2] == Nm else torch.cat([t, t[:, :, :1].expand(-1, -1, Nm - t.shape[2])], dim=2)
            nv = torch.tensor([[N1] * B, [N2] * B], dtype=torch.int32, device=pc1.device)
            flow, h, cls, cor, f1, f2, prop = eng.backbone(pad(pc1), pad(pc2), pad(feature1), pad(feature2), h, n_valid=nv)
            return (flow[:, :, :N1].contiguous(), h, cls[:, :N1].contiguous(), cor[:, :, :N1].contiguous(),
                    f1[:, :, :N1].contiguous(), f2[:, :, :N2].contiguous(), prop[:, :, :N1].contiguous())
        tg1 = nv = None
        cut = None
        if self.training and self._dedup_train and pc1.is_cuda:
            from . import train_path as TP
            if not (TP.supported(self.pn_head) and TP.supported(self.fd_layer.mse) and TP.correlator_supported(self.fc_layer)):
                # no silent drop to the framework's convolutions: the module path is an explicit choice
                raise NotImplementedError("Track4D: this layer configuration is outside the hand-written training path (max-pooled MSG "
                                          "levels of Conv2d(no bias)+BatchNorm2d+ReLU, 3x256-channel correlator without BatchNorm); set "
                                          "the module path (PyTorch-ROCm dense layers) is reachable through the test hook net._dedup_train only")
            # training step: both frames as one stacked batch (per-frame BatchNorm statistics) on de-duplicated levels.
            # Clouds of different sizes -- every real consecutive pair (dataset_classes/track_vod_3d.py:80-84,119), or a padded
            # batch with n_valid -- are padded with copies of their own point 0 and travel with their true counts on the
            # device: same results, gradients and running statistics as the unpadded B = 1 runs (train_path.TrainGeometry)
            B, N1, N2 = pc1.shape[0], pc1.shape[2], pc2.shape[2]
            if n_valid is not None:
                assert N1 == N2, "n_valid batches are already padded to a common size"
                nv = n_valid.to(device=pc1.device, dtype=torch.int32).reshape(2 * B).contiguous()
            elif N1 != N2:
                Nm = max(N1, N2)
                pad = lambda t: t if t.shape[2] == Nm else torch.cat([t, t[:, :, :1].expand(-1, -1, Nm - t.shape[2])], dim=2)
                nv = torch.cat([torch.full((B,), N1, dtype=torch.int32, device=pc1.device),
                                torch.full((B,), N2, dtype=torch.int32, device=pc1.device)])
                pc1, pc2, feature1, feature2 = pad(pc1), pad(pc2), pad(feature1), pad(feature2)
                cut = (N1, N2)
            with torch.no_grad():      # (geometry on a forked stream was measured slower inside the captured step: DESIGN.md section 5)
                tg = TP.TrainGeometry(torch.cat([pc1, pc2], 0).permute(0, 2, 1).contiguous(), self.pn_head.sa1.npoint, n_valid=nv, groups=2)
            f = TP.pnhead_train(self.pn_head, tg, torch.cat([feature1, feature2], 0), groups=2)
            # both frames' [per-point features ; global feature] in one kernel (and one in the backward) instead of max, expand,
            # cat -- and sum, zero fill, scatter, accumulate -- per frame
            from .train_ops import gmax_cat
            pf = gmax_cat(f)
            (pc1_features, pc2_features), tg1 = pf.view(2, B, pf.shape[1], pf.shape[2]).unbind(0), tg.head(B)
        if tg1 is None and n_valid is not None:
            return self._backbone_per_sample(pc1, pc2, feature1, feature2, h, n_valid)
        if tg1 is None:
            xyz1_new, f1 = self.pn_head(pc1.permute(0, 2, 1).contiguous(), feature1)
            xyz2_new, f2 = self.pn_head(pc2.permute(0, 2, 1).contiguous(), feature2)
        if tg1 is None:
            g1 = torch.max(f1, -1)[0].unsqueeze(2).expand(-1, -1, pc1.size(2))
            g2 = torch.max(f2, -1)[0].unsqueeze(2).expand(-1, -1, pc2.size(2))
            pc1_features = torch.cat((f1, g1), dim=1)
            pc2_features = torch.cat((f2, g2), dim=1)
        if tg1 is not None:
            cor_features = TP.correlator_train(self.fc_layer, pc1, pc2, pc1_features, pc2_features,
                                               n_valid1=None if nv is None else nv[:pc1.shape[0]], n_valid2=None if nv is None else nv[pc1.shape[0]:],
                                               xyz=(tg.xyz[:pc1.shape[0]], tg.xyz[pc1.shape[0]:]))
        else:
            assert nv is None, "padded batches need the fused correlator of the training path"
            cor_features = self.fc_layer(pc1, pc2, pc1_features, pc2_features)
        output, h, prop_features, cls = self.fd_layer(pc1, feature1, pc1_features, cor_features, h, train_geo=tg1)
        if tg1 is not None:
            tg1.join()          # every forked stream is joined before the forward returns (a requirement inside a stream capture)
        if cut is not None:     # frames of different sizes were padded above: hand back the callers' sizes
            N1, N2 = cut
            return (output[:, :, :N1], h, cls[:, :N1], cor_features[:, :, :N1], pc1_features[:, :, :N1], pc2_features[:, :, :N2],
                    prop_features[:, :, :N1])
        return output, h, cls, cor_features, pc1_features, pc2_features, prop_features

    def _backbone_per_sample(self, pc1, pc2, feature1, feature2, h, n_valid):
        """The literal meaning of a padded batch: every sample run on its own unpadded clouds (B = 1, as the reference does),
        results re-padded with zeros.  Used by the module / training paths and as the cross-check of the fused path."""
        B, _, N = pc1.shape
        nv = n_valid.cpu().tolist()
        outs = []
        for b in range(B):
            n1, n2 = nv[0][b], nv[1][b]
            hb = None if h is None else h[:, b:b + 1].contiguous()
            o = self.backbone(pc1[b:b + 1, :, :n1].contiguous(), pc2[b:b + 1, :, :n2].contiguous(), feature1[b:b + 1, :, :n1].contiguous(),
                              feature2[b:b + 1, :, :n2].contiguous(), hb)
            outs.append([o[1]] + [torch.nn.functional.pad(t, (0, N - t.shape[-1])) for i, t in enumerate(o) if i != 1])
        hs = torch.cat([o[0] for o in outs], dim=1)
        rest = [torch.cat([o[i] for o in outs], dim=0) for i in range(1, 7)]
        return (rest[0], hs) + tuple(rest[1:])

    def _weights_version(self):
        """Sum of the in-place modification counters of every parameter and buffer the folded engine was built from: `p.data.mul_(2)`,
        `running_mean.copy_(...)`, a torch optimizer step taken in eval mode ... all bump it, so a stale folded engine is detected
        without a hook.  The tensor list is cached with the engine (walking the module tree costs 1 ms per call -- more than a B = 64
        forward; the cached sum costs 35 us) and dropped with it (load_state_dict / train() / .to()).  Writers that go through raw
        pointers bump the counters themselves: `optim.FusedAdam.step` (rtk_adam_multi) calls
        `torch.autograd.graph.increment_version` on what it updated; the training BatchNorm kernels only run in train mode, and
        `train()` drops the engine.  Replacing a Parameter OBJECT (`net.x.weight = nn.Parameter(...)`) is not seen: call
        `invalidate_fused()`."""
        ts = self._fused_tensors
        if ts is None:
            ts = self._fused_tensors = list(self.parameters()) + list(self.buffers())
        return sum(t._version for t in ts)

    def _fused_engine(self):
        """The folded / packed inference engine (ratrack_amd.fused.FusedBackbone), rebuilt when the weights moved.  There is no
        fallback: a missing HIP library or fused module raises (the module path is the tests' comparison implementation, behind the `_use_fused` hook)."""
        if self._fused is not None and self._fused_version != self._weights_version():
            self._fused = None          # weights were edited in place since the engine folded / packed them
        if self._fused is None:
            from . import fused
            self._fused_tensors = None
            self._fused = fused.FusedBackbone(self)
            self._fused_version = self._weights_version()
        return self._fused

    def invalidate_fused(self):
        """Call after changing weights (load_state_dict does it automatically)."""
        self._fused = self._fused_tensors = None

    def load_state_dict(self, *a, **k):
        self._fused = self._fused_tensors = None
        return super().load_state_dict(*a, **k)

    def train(self, mode=True):
        if mode:
            self._fused = self._fused_tensors = None      # folded BN constants go stale once training resumes
        return super().train(mode)

    def _apply(self, fn, *a, **k):
        self._fused = self._fused_tensors = None          # .to(device) / .float() / .cuda(): the packed weight images belong to the old tensors
        return super()._apply(fn, *a, **k)

    @property
    def max_id(self):
        return self.associator.max_id

    @max_id.setter
    def max_id(self, v):
        self.associator.max_id = v

    def detect_and_associate(self, pc1, feature1, flow, cls, prop_features, objects_prev):
        """models/track4d.py:53-63: select the moving points, cluster them into objects, associate with the previous
        frame's objects.  B = 1.  Returns (pc1_warp, aff_list, aff_mat, indices1, confs, objects, objects_curr)."""
        assert pc1.shape[0] == 1, "detection / association is single-frame logic (B = 1), as in the reference"
        pc1_warp = pc1 + flow
        point_features = torch.cat((pc1_warp, pc1, flow, feature1, prop_features), dim=1)      # (1,139,N)
        if point_features.is_cuda:      # every cloud size on the device (rtk_dbscan: LDS tables up to ~2900 points, a global workspace beyond)
            objects_curr = A.cluster_objects_device(point_features, cls, eps=1.5, min_samples=self.min_obj_points)
        else:       # host tensors only: the CPU tests of the association bookkeeping (the backbone itself has no CPU path)
            mov_mask = (cls > 0.5).squeeze(0)
            objects_curr = A.cluster_objects(point_features[:, :, mov_mask], eps=1.5, min_samples=self.min_obj_points)
        aff_list, aff_mat, indices1, confs, objects = self.associator(objects_curr, objects_prev or dict())
        return pc1_warp, aff_list, aff_mat, indices1, confs, objects, objects_curr

    def forward(self, pc1, pc2, feature1, feature2, h, objects_prev=None):
        """The reference's 10-tuple (models/track4d.py:49-65):
        (h, pc1_warp, cls, aff_list, aff_mat, indices1, confs, objects, timeout_obj_curr, objects_curr)."""
        output, h, cls, cor, pc1_features, pc2_features, prop = self.backbone(pc1, pc2, feature1, feature2, h)
        pc1_warp, aff_list, aff_mat, indices1, confs, objects, objects_curr = self.detect_and_associate(
            pc1, feature1, output, cls, prop, objects_prev)
        return h, pc1_warp, cls, aff_list, aff_mat, indices1, confs, objects, dict(), objects_curr


class Args(dict):
    """Attribute-style config (the reference's EasyDict, utils/parser_util.py:12-35); defaults are the
    hot-path keys of configs.yaml (:5,:25,:31,:33)."""

    def __init__(self, **kw):
        base = dict(num_points=256, npoints=512, rigid_thres=0.15, min_obj_points=2)
        base.update(kw)
        super().__init__(base)
        self.__dict__ = self


def init_model(args=None, device="cuda"):
    """Counterpart of models/model.py:17-43 (factory; no DataParallel -- see ratrack_amd/ddp.py)."""
    return Track4D(args or Args()).to(device)
