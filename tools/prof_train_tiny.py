#!/usr/bin/env python3
"""Where do the tiny kernels of the train step come from?  Counts of copy_/fill_/zeros/clone/sum by python call site."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from ratrack_amd import synth
from ratrack_amd.track4d import Track4D, Args
from ratrack_amd.train import Trainer
dev = "cuda"; B = 64
net = Track4D(Args()).to(dev); synth.fill_state_dict(net.state_dict())
d = synth.make_frame_pairs(B, 256, 0); t = {k: torch.from_numpy(v).to(dev) for k, v in d.items()}
h = torch.zeros(5, B, 128, device=dev); tr = Trainer(net)
step = lambda: tr.step(t["pc1"], t["pc2"], t["feature1"], t["feature2"], t["gt_warp"], t["gt_cls"], h)
for _ in range(4): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter(); tim = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::fill_", "aten::sum", "aten::mul", "aten::add", "aten::cat", "aten::index", "aten::where", "aten::div", "aten::sub"):
        site = "?"
        for fr in ev.stack:
            if "/root/repo" in fr or "ratrack_amd" in fr:
                site = fr.split("/")[-1]
                break
        else:
            site = (ev.stack[0].split("/")[-1] if ev.stack else "autograd/none")
        cnt[(ev.name, site)] += 1
        tim[(ev.name, site)] += ev.device_time
for (k, c) in cnt.most_common(45):
    print("%5d  %8.1f us  %-14s %s" % (c, tim[k], k[0], k[1][:110]))
