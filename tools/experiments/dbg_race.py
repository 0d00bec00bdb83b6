"""Stress: is the fused eval path on real frame pairs (N1 != N2, duplicates inside the clouds) reproducible run to run?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from _util import load_case, inputs_of, reference_state_dict, rel_err
from ratrack_amd.track4d import Args, Track4D
from ratrack_amd import fused
DEV = "cuda"
case = load_case("real_549_1047")
pc1, pc2, f1, f2 = inputs_of(case, DEV)
sd = reference_state_dict(DEV)
for side in (True, False):
    bad = 0
    errs = []
    for it in range(40):
        net = Track4D(Args()).to(DEV).eval()
        net.load_state_dict(sd, strict=True)
        junk = [torch.randn(np.random.randint(1000, 3000000), device=DEV) for _ in range(np.random.randint(1, 6))]
        with torch.no_grad():
            eng = net._fused_engine()
            eng.use_side_stream = side
            out = net.backbone(pc1, pc2, f1, f2, None)
        e = rel_err(out[0].cpu(), case["flow"])
        errs.append(e)
        bad += e > 1e-4
        del junk
    print("side stream", side, ": %d / 40 runs off the fixture; errors %s" % (bad, sorted(set("%.1e" % e for e in errs))))
