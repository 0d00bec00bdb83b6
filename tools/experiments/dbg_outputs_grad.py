#!/usr/bin/env python3
"""Round 5 debug: gradient of a random linear functional of EACH backbone output, hand-written training path vs module path."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from _util import reference_state_dict
from ratrack_amd import synth
from ratrack_amd.track4d import Args, Track4D
DEV = "cuda"
B, N = 1, 256
d = synth.make_frame_pairs(B, N, 21)
t = {k: torch.from_numpy(v).to(DEV) for k, v in d.items()}
names = ["flow", "h", "cls", "cor", "f1", "f2", "prop"]
def run(dedup, which, sparse=False):
    net = Track4D(Args()).to(DEV); net.load_state_dict(reference_state_dict(DEV), strict=True); net.train(); net._dedup_train = dedup
    outs = net.backbone(t["pc1"], t["pc2"], t["feature1"], t["feature2"], torch.zeros(5, B, 128, device=DEV))
    o = outs[which]
    g = torch.Generator(DEV).manual_seed(which)
    R = torch.randn(o.shape, device=DEV, generator=g)
    if sparse:      # as a max-pool over a few points sends it: most columns zero
        R = R * (torch.rand(o.shape, device=DEV, generator=g) < 0.05)
    (o * R).sum().backward()
    return {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
for which in range(7):
    for sparse in (False, True):
        a, b = run(True, which, sparse), run(False, which, sparse)
        gmax = max(float(v.norm()) for v in b.values())
        rel = sorted((float((a[k] - b[k]).norm() / b[k].norm()), k) for k in b if k in a and float(b[k].norm()) > 1e-4 * gmax)
        missing = [k for k in b if k not in a and float(b[k].norm()) > 1e-6 * gmax]
        print("%-5s %s: %3d tensors, median %.2e, worst %.2e (%s)  missing %s" % (names[which], "sparse" if sparse else "dense ", len(rel), rel[len(rel) // 2][0], rel[-1][0], rel[-1][1], missing[:3]))
