#!/usr/bin/env python3
"""Workload of the PMC passes (run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and `--pmc WRITE_SIZE --kernel-trace`,
separately, as MI355X_MICROARCH.md prescribes): a calibration copy of known size (torch.clone of 256 MiB: reads 256 MiB, writes
256 MiB), a few EAGER forward steps, a few eager train steps (B=64, N=256) and a few launches of every irregular kernel at
the bench shape (ratrack_amd.benchutil.irregular_ops).  Eager on purpose: a graph replay hides the kernels from the counters."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ratrack_amd import benchutil, synth
from ratrack_amd.track4d import Track4D, Args
from ratrack_amd.train import Trainer
ap = argparse.ArgumentParser()
ap.add_argument("--dominant", action="store_true", help="bench.py's in-run pass: calibration copy, two forward and two train steps only")
ap.add_argument("--train-steps", type=int, default=0, help="train-step pass (tools/pmc_train_total.py): calibration + this many train steps only")
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--npoints", type=int, default=256)
a = ap.parse_args()
dev = "cuda"
x = torch.randn(64 * 1024 * 1024, device=dev)        # 256 MiB
for _ in range(3):
    y = x.clone()
torch.cuda.synchronize()
del x, y
net = Track4D(Args()).to(dev).eval()
synth.fill_state_dict(net.state_dict())
d = synth.make_frame_pairs(a.batch, a.npoints, 1000)
t = {k: torch.from_numpy(v).to(dev) for k, v in d.items()}
h = torch.zeros(5, a.batch, 128, device=dev)
if a.dominant or a.train_steps:
    if a.dominant:
        with torch.no_grad():
            for _ in range(2):
                net.backbone(t["pc1"], t["pc2"], t["feature1"], t["feature2"], h)
    tr = Trainer(net)
    for _ in range(a.train_steps or 2):
        tr.step(t["pc1"], t["pc2"], t["feature1"], t["feature2"], t["gt_warp"], t["gt_cls"], h)
    torch.cuda.synchronize()
    sys.exit(0)
with torch.no_grad():
    for _ in range(4):
        net.backbone(t["pc1"], t["pc2"], t["feature1"], t["feature2"], h)
torch.cuda.synchronize()
benchutil.irregular_ops(64, 256, torch.device(dev), iters=3)
tr = Trainer(net)
for _ in range(3):
    tr.step(t["pc1"], t["pc2"], t["feature1"], t["feature2"], t["gt_warp"], t["gt_cls"], h)
torch.cuda.synchronize()
