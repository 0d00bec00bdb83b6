#!/usr/bin/env python3
"""The forward cost-volume kernel alone at the bench shape (library from RTK_SO_PATH: ablation variants of fused_split.hip).
python tools/exp_cv_time.py [batch]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from ratrack_amd import synth
from ratrack_amd.track4d import Args, Track4D
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
net = Track4D(Args()).to("cuda").eval()
synth.fill_state_dict(net.state_dict()); net.invalidate_fused()
d = synth.make_frame_pairs(B, 256, 1)
t = [torch.from_numpy(d[k]).to("cuda") for k in ("pc1", "pc2", "feature1", "feature2")]
with torch.no_grad():
    net.backbone(*t, None)
    eng = net._fused
    eng.time_dominant_kernel(5)
    ev = eng.time_dominant_kernel(30)
ms = sorted(s.elapsed_time(e) for s, e in ev)
print("%s: cost volume forward B=%d: median %.1f us" % (os.environ.get("RTK_SO_PATH", "default")[-22:], B, ms[len(ms) // 2] * 1e3))
