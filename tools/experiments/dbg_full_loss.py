#!/usr/bin/env python3
"""Round 5 debug: per-tensor gradient error of the full-loss training iteration (tests/test_model_gpu.py::test_full_loss_train_step_
matches_reference) with and without the tracking term.  python tools/experiments/dbg_full_loss.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from _util import grad_sample, load_case, reference_state_dict
from ratrack_amd import loss as L
from ratrack_amd.track4d import Args, Track4D
DEV = "cuda"
case = load_case("train_full_b1_n256")
names = [str(k) for k in case["grad_names"]]
gmax = max(float(np.abs(case["grad/" + k]).max()) for k, n in zip(names, case["grad_norms"]) if n >= 0)

def run(with_trk, dedup=True, sf=0.5):
    sd = reference_state_dict(DEV)
    sd["fd_layer.cp.linear.bias"] = sd["fd_layer.cp.linear.bias"] + 0.09
    net = Track4D(Args()).to(DEV); net.load_state_dict(sd, strict=True); net.train()
    net._dedup_train = dedup
    g = lambda fi, k: torch.from_numpy(case["f%d_in_%s" % (fi, k)]).to(DEV)
    h = torch.zeros(5, 1, 128, device=DEV)
    h, _, _, _, _, _, _, objects, _, _ = net(g(0, "pc1"), g(0, "pc2"), g(0, "feature1"), g(0, "feature2"), h, dict())
    objects_prev = {k: v.clone().detach() for k, v in objects.items()}
    net.zero_grad()
    h1, pc1_warp, cls, aff_list, aff_mat, indices1, confs, objects, _, objects_curr = net(g(1, "pc1"), g(1, "pc2"), g(1, "feature1"), g(1, "feature2"), h.detach(), objects_prev)
    mp = {int(k): i for i, k in enumerate(case["prev_keys"])}; mc = {int(k): i for i, k in enumerate(case["curr_keys"])}
    gt, gt_cls = g(1, "gt_warp"), torch.from_numpy(case["f1_gt_cls_used"]).to(DEV)
    total, items = L.track_4d_loss(objects_prev, objects, mp if with_trk else {}, mc if with_trk else {}, None, None, None, g(1, "pc1"), g(1, "pc2"), pc1_warp, cls, gt,
                                   aff_list, None, gt_cls, None, None, None, pretrain=False)
    total.backward()
    out = {}
    for k, p in net.named_parameters():
        if p.grad is None: continue
        ref = case["grad/" + k].astype(np.float64)
        if np.abs(ref).max() <= 1e-5 * gmax: continue
        out[k] = float(np.abs(grad_sample(p.grad.detach().float().cpu().numpy()) - ref).max() / np.abs(ref).max())
    return out, {k: float(v) for k, v in items.items()}

a, ia = run(True)
b, ib = run(False)
c, ic = run(True, dedup=False)
print("losses", ia, "reference", dict(zip([str(k) for k in case["loss_keys"]], case["loss_vals"])))
print("%-60s %10s %10s %10s" % ("tensor", "full", "no trk", "module path"))
for k in a:
    print("%-60s %10.2e %10.2e %10.2e" % (k, a[k], b.get(k, float("nan")), c.get(k, float("nan"))))
