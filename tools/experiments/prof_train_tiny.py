#!/usr/bin/env python3
"""Which small framework ops does the train step issue?  Counts by op and input shapes."""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = []
for ev in prof.key_averages(group_by_input_shape=True):
    if ev.key.startswith("aten::") and ev.device_time_total > 0 and ev.self_device_time_total > 0:
        rows.append((ev.count, ev.self_device_time_total, ev.key, str(ev.input_shapes)[:150]))
rows.sort(key=lambda r: (-r[0], -r[1]))
for c, tm, k, sh in rows[:70]:
    print("%4d %8.1f us  %-28s %s" % (c, tm, k, sh))
