"""The forward cost-volume kernel alone at the bench shape: microseconds per launch (median of 30)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ratrack_amd import synth
from ratrack_amd.track4d import Args, Track4D
dev = "cuda"
net = Track4D(Args()).to(dev).eval(); synth.fill_state_dict(net.state_dict())
d = synth.make_frame_pairs(64, 256, 1000)
t = [torch.from_numpy(d[k]).to(dev) for k in ("pc1", "pc2", "feature1", "feature2")]
with torch.no_grad():
    net.backbone(*t, None)
    ev = net._fused.time_dominant_kernel(40)
ms = sorted(s.elapsed_time(e) for s, e in ev)
print("cost volume forward alone: median %.1f us, min %.1f us" % (ms[len(ms) // 2] * 1e3, ms[0] * 1e3))
