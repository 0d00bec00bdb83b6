#!/usr/bin/env python3
"""Which Python lines launch the framework-native kernels (fill, add, cat, reduce, copy ...) of one eager train step.
python tools/prof_glue.py [batch] [pattern, default 'fill']"""
import collections, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from torch.profiler import ProfilerActivity, profile
from ratrack_amd import synth
from ratrack_amd.track4d import Args, Track4D
from ratrack_amd.train import Trainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
pat = sys.argv[2].lower() if len(sys.argv) > 2 else "fill"
net = Track4D(Args()).to("cuda"); synth.fill_state_dict(net.state_dict())
tr = Trainer(net, graph=False)
d = synth.make_frame_pairs(B, 256, 7)
t = {k: torch.from_numpy(v).to("cuda") for k, v in d.items()}
h = torch.zeros(5, B, 128, device="cuda")
step = lambda: tr.step(t["pc1"], t["pc2"], t["feature1"], t["feature2"], t["gt_warp"], t["gt_cls"], h)
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and pat in e.name.lower() and e.name.startswith("aten::"):
        own = [f for f in (e.stack or []) if "ratrack_amd" in f]
        cnt[(e.name[:40], tuple(own[:3]) if own else tuple((e.stack or ["<no python stack: autograd engine>"])[:2]))] += 1
for (name, st), n in cnt.most_common(30):
    print("%3d  %-40s %s" % (n, name, " <- ".join(s.split("/")[-1][:70] for s in st)))
