#!/usr/bin/env python3
"""Steady-state kernel breakdown of the module-path train step (torch.profiler, after MIOpen's find phase)."""
import sys, os
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
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=400, max_name_column_width=70))
