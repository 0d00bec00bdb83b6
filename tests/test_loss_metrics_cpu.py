"""CPU: product loss / metrics (host logic) against the golden values captured from the reference."""
import numpy as np
import torch

from ratrack_amd import loss as L
from ratrack_amd import metrics as M

from _util import load_case


def test_metrics_match_reference():
    for name in ["eval_b2_n256", "eval_b1_n1024", "eval_b1_n256_dups"]:
        case = load_case(name)
        pc1 = torch.from_numpy(case["in_pc1"][:1])
        flow = torch.from_numpy(case["flow"][:1])
        gt = torch.from_numpy(case["in_gt_warp"][:1])
        mask = torch.from_numpy(~case["in_gt_cls"][:1]).float()
        sf = M.eval_scene_flow(pc1, pc1 + flow, gt, mask)
        for k, v in zip(case["metric_sf_keys"], case["metric_sf_vals"]):
            assert abs(sf[str(k)] - v) <= 1e-6 * max(1.0, abs(v)), (name, k, sf[str(k)], v)
        pre = (torch.from_numpy(case["cls"][:1]) > 0.5).float()
        seg = M.eval_motion_seg(pre, torch.from_numpy(case["in_gt_cls"][:1]).float())
        for k, v in zip(case["metric_seg_keys"], case["metric_seg_vals"]):
            assert abs(seg[str(k)] - v) <= 1e-9 + 1e-9 * abs(v), (name, k)


def test_loss_matches_reference_on_oracle_outputs():
    """Loss arithmetic incl. pretrain mode and the NaN->0 guard, on tensors from the CPU oracle."""
    from oracle import track4d_ref as R
    from _util import inputs_of, reference_state_dict
    case = load_case("train_b1_n256")
    sd = reference_state_dict()
    pc1, pc2, f1, f2 = inputs_of(case)
    with torch.no_grad():
        flow, h, cls, *_ = R.backbone(sd, pc1, pc2, f1, f2, None, training=True)
    gt = torch.from_numpy(case["in_gt_warp"])
    gt_cls = torch.from_numpy(case["in_gt_cls"])
    keys = [str(k) for k in case["loss_keys"]]
    for pretrain, gcls, ref in [(False, gt_cls, "loss_vals"), (True, gt_cls, "loss_vals_pretrain"),
                                (False, torch.zeros_like(gt_cls), "loss_vals_nopos")]:
        _, items = L.backbone_loss(pc1 + flow, cls, gt, gcls, pretrain=pretrain)
        np.testing.assert_allclose([float(items[k]) for k in keys], case[ref], rtol=1e-5, atol=1e-7)
        _, items = L.track_4d_loss(None, None, {}, {}, None, None, None, pc1, pc2, pc1 + flow, cls, gt, [], None,
                                   gcls[0], None, None, None, pretrain=pretrain)
        np.testing.assert_allclose([float(items[k]) for k in keys], case[ref], rtol=1e-5, atol=1e-7)


def test_batched_loss_is_mean_of_per_sample():
    torch.manual_seed(0)
    warp, gt = torch.randn(3, 3, 50), torch.randn(3, 3, 50)
    cls = torch.rand(3, 50)
    g = torch.rand(3, 50) > 0.7
    g[2] = False       # sample without positives: its seg term is NaN -> 0
    tot, it = L.backbone_loss(warp, cls, gt, g)
    per = [L.backbone_loss(warp[i:i + 1], cls[i:i + 1], gt[i:i + 1], g[i:i + 1])[1] for i in range(3)]
    assert abs(float(it["SegLoss"]) - np.mean([float(p["SegLoss"]) for p in per])) < 1e-6
    assert abs(float(it["SceneFlowLoss"]) - np.mean([float(p["SceneFlowLoss"]) for p in per])) < 1e-6
    assert float(per[2]["SegLoss"]) == 0.0
    # the zeroed sample must contribute a ZERO gradient, not NaN (0/0 under a nan->0 select poisons the weights)
    cls2 = cls.clone().requires_grad_(True)
    L.backbone_loss(warp, cls2, gt, g)[0].backward()
    assert torch.isfinite(cls2.grad).all() and float(cls2.grad[2].abs().sum()) == 0.0


def test_affinity_loss_matches_reference():
    """losses/loss.py:48-72: BCE between the flattened affinity list and the identity-match matrix of the GT mappings
    (values captured from the reference by tools/make_golden_gt.py --affinity), incl. the empty-mapping case."""
    import os
    from _util import GOLDEN
    from ratrack_amd import loss as L
    g = dict(np.load(os.path.join(GOLDEN, "affinity_loss.npz")))
    for ci in range(3):
        prev, curr = g["aff%d/prev" % ci].tolist(), g["aff%d/curr" % ci].tolist()
        mp = {k: 100 + i for i, k in enumerate(prev)}
        mc = {k: 200 + i for i, k in enumerate(curr)}
        v = L.affinity_loss(mp, mc, torch.from_numpy(g["aff%d/aff" % ci]))
        assert abs(float(v) - float(g["aff%d/val" % ci])) <= 1e-6 * max(1.0, float(g["aff%d/val" % ci])), ci
    assert float(L.affinity_loss({}, {1: 2}, torch.rand(0))) == float(g["aff_empty"]) == 0.0
    # the tracking term enters the total with weight 0.5 (losses/loss.py:22-24) through the 19-argument entry point
    pc1 = torch.randn(1, 3, 16)
    cls = torch.rand(1, 16) * 0.9 + 0.05
    gt_cls = torch.arange(16) % 3 == 0
    mp, mc = {5: 1, 7: 2}, {7: 3, 5: 4}
    aff = torch.tensor([0.2, 0.7, 0.9, 0.1])
    total, items = L.track_4d_loss(None, None, mp, mc, None, None, None, pc1, pc1, pc1 + 0.1, cls, pc1, aff, None, gt_cls, None, None, None)
    trk = float(L.affinity_loss(mp, mc, aff))
    assert abs(float(items["TrackingLoss"]) - trk) < 1e-7
    assert abs(float(total) - (0.5 * float(items["SceneFlowLoss"]) + 0.5 * trk + float(items["SegLoss"]))) < 1e-6
