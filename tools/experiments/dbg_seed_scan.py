#!/usr/bin/env python3
"""Round 5: which synthetic frame pairs have no flipped decision between the hand-written training path and the module path (B = 1,
train mode)?  Error of the gradient of a dense functional of flow + prop; object counts of forward() with the shifted cls bias."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from _util import reference_state_dict
from ratrack_amd import synth
from ratrack_amd.track4d import Args, Track4D
DEV = "cuda"
def run(dedup, t):
    sd = reference_state_dict(DEV); sd["fd_layer.cp.linear.bias"] = sd["fd_layer.cp.linear.bias"] + 0.09
    net = Track4D(Args()).to(DEV); net.load_state_dict(sd, strict=True); net.train(); net._dedup_train = dedup
    h, pc1_warp, cls, aff_list, aff_mat, ind, confs, objects, _, oc = net(t["pc1"], t["pc2"], t["feature1"], t["feature2"], torch.zeros(5, 1, 128, device=DEV), dict())
    outs = net.backbone(t["pc1"], t["pc2"], t["feature1"], t["feature2"], torch.zeros(5, 1, 128, device=DEV))
    g = torch.Generator(DEV).manual_seed(1)
    ((outs[0] * torch.randn(outs[0].shape, device=DEV, generator=g)).sum() + (outs[6] * torch.randn(outs[6].shape, device=DEV, generator=g)).sum()).backward()
    return {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}, len(oc), int((cls > 0.5).sum()), float((cls - 0.5).abs().min())
for seed in range(20, 44):
    d = synth.make_frame_pairs(1, 256, seed)
    t = {k: torch.from_numpy(v).to(DEV) for k, v in d.items()}
    a, nobj, nmov, margin = run(True, t)
    b, _, _, _ = run(False, t)
    gmax = max(float(v.norm()) for v in b.values())
    rel = sorted(float((a[k] - b[k]).norm() / b[k].norm()) for k in b if k in a and float(b[k].norm()) > 1e-4 * gmax)
    print("seed %d: objects %2d movers %3d margin %.1e gt_pos %3d | median %.2e worst %.2e" % (seed, nobj, nmov, margin, int(d["gt_cls"].sum()), rel[len(rel) // 2], rel[-1]), flush=True)
