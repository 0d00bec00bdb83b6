#!/usr/bin/env python3
"""Workload for the PMC passes: a calibration copy of known size (torch.clone of 256 MiB: reads 256 MiB, writes 256 MiB)
followed by a few eager backbone steps.  Run under  rocprofv3 --pmc FETCH_SIZE --kernel-trace  and  --pmc WRITE_SIZE."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ratrack_amd import synth
from ratrack_amd.track4d import Track4D, Args
dev = "cuda"
x = torch.randn(64 * 1024 * 1024, device=dev)        # 256 MiB
for _ in range(3):
    y = x.clone()
torch.cuda.synchronize()
net = Track4D(Args()).to(dev).eval()
synth.fill_state_dict(net.state_dict())
d = synth.make_frame_pairs(64, 256, 1000)
t = {k: torch.from_numpy(v).to(dev) for k, v in d.items() if k != "gt_cls"}
h = torch.zeros(5, 64, 128, device=dev)
with torch.no_grad():
    for _ in range(4):
        net.backbone(t["pc1"], t["pc2"], t["feature1"], t["feature2"], h)
torch.cuda.synchronize()
