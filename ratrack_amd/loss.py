"""Multi-task loss of the RaTrack backbone (reference: losses/loss.py).

`track_4d_loss` keeps the reference's 19-argument signature and return value (:8-31) so the epoch
loop can call it unchanged; `backbone_loss` is the batch-general form used by the trainer here.
Reference semantics that are easy to lose and are reproduced on purpose:
  * the scene-flow term is the mean over points of ||pc1_warp - gt||_2 (`gt_flow` holds GT *warped
    positions*, not displacements) of batch element 0 only (:85-89) -- generalised to the batch mean,
    identical at the reference's B = 1;
  * the segmentation term is 0.4*BCE(positives) + 0.6*BCE(negatives) (:124-146); a class with no
    sample gives BCE over an empty set = NaN, and NaN terms are replaced by 0 (:15-20);
  * total = 0.5*flow + 0.5*tracking + seg, or seg alone while pre-training (:22-24).
Masked means are computed with arithmetic masks (no boolean indexing), so the loss adds no
device->host synchronisation to the training step.
"""
import torch
import torch.nn.functional as F


def flow_loss(pc1_warp, gt_flow):
    """(B,3,N),(B,3,N) -> (B,) per-sample mean end-point distance."""
    return (pc1_warp - gt_flow).pow(2).sum(dim=1).sqrt().mean(dim=1)


def motion_seg_loss(pred_cls, gt_cls):
    """pred (B,N) probabilities, gt (B,N) or (N,) bool -> ((B,) 0.4*BCE_pos + 0.6*BCE_neg, (B,) bool defined).
    A sample without positives (or without negatives) has an undefined term -- the reference gets NaN from a mean over
    an empty selection and then zeroes the WHOLE segmentation loss (losses/loss.py:19-20).  The division is masked
    (x / max(count,1) selected by count > 0) so that the zeroed samples also have exactly zero -- not NaN -- gradient."""
    if gt_cls.dim() == 1:
        gt_cls = gt_cls.unsqueeze(0).expand_as(pred_cls)
    g = gt_cls.to(pred_cls.dtype)
    bce = F.binary_cross_entropy(pred_cls, g, reduction="none")
    npos, nneg = g.sum(1), (1 - g).sum(1)
    pos = (bce * g).sum(1) / npos.clamp_min(1)
    neg = (bce * (1 - g)).sum(1) / nneg.clamp_min(1)
    return 0.4 * pos + 0.6 * neg, (npos > 0) & (nneg > 0)


def affinity_loss(mappings_prev, mappings_curr, aff_mat):
    """losses/loss.py:48-72: BCE between the predicted affinity list and the identity-match matrix."""
    if len(mappings_prev) == 0 or len(mappings_curr) == 0:
        return torch.tensor(0)
    prev, curr = list(mappings_prev.keys()), list(mappings_curr.keys())
    gt = torch.tensor([1.0 if m == n else 0.0 for m in prev for n in curr], device=aff_mat.device)
    return F.binary_cross_entropy(aff_mat.float(), gt)


def _nan_to_zero(x):
    return torch.where(torch.isnan(x), torch.zeros_like(x), x)


def backbone_loss(pc1_warp, cls, gt_flow, gt_cls, pretrain=False, trk_loss=None):
    """Batch mean of the per-sample reference loss.  Returns (total, items) with the reference's keys."""
    sf = _nan_to_zero(flow_loss(pc1_warp, gt_flow)).mean()
    seg_i, defined = motion_seg_loss(cls, gt_cls)
    seg = torch.where(defined, seg_i, torch.zeros_like(seg_i)).mean()
    trk = trk_loss if trk_loss is not None else torch.zeros((), device=pc1_warp.device)
    total = seg if pretrain else 0.5 * sf + 0.5 * trk + seg
    return total, {"Loss": total, "SceneFlowLoss": sf, "TrackingLoss": trk, "SegLoss": seg}


def track_4d_loss(objs1, objs2, mappings_prev, mappings_curr, mappings_inv, lbl1, lbl2, pc1, pc2, pc1_wrap, cls, gt_flow,
                  aff_list, gt_mov_pts, gt_cls, gt_objs, objs_idx, objs_centre, pretrain=False):
    """Reference signature (losses/loss.py:8-31).  Only pc1_wrap, cls, gt_flow, gt_cls, the two
    mappings and aff_list enter the value -- exactly the arguments the reference's body reads."""
    trk = affinity_loss(mappings_prev or {}, mappings_curr or {}, aff_list)
    trk = trk.to(pc1_wrap.device).float()
    if torch.isnan(trk):
        trk = torch.zeros((), device=pc1_wrap.device)
    # the reference evaluates batch element 0 only (:89) and a single gt_cls vector (:126-131)
    return backbone_loss(pc1_wrap[:1], cls[:1], gt_flow[:1], gt_cls if gt_cls.dim() == 1 else gt_cls[:1],
                         pretrain=pretrain, trk_loss=trk)
