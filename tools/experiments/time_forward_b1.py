#!/usr/bin/env python3
"""End-to-end latency of Track4D.forward at the reference's operating point (B = 1, consecutive frames): fused backbone +
moving-point clustering + Affinity/Sinkhorn association.  Not a bench.py metric; documents where the per-frame time goes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ratrack_amd import synth
from ratrack_amd.track4d import Track4D, Args
dev = "cuda"
net = Track4D(Args()).to(dev).eval()
synth.fill_state_dict(net.state_dict())
with torch.no_grad():
    net.fd_layer.cp.linear.bias.add_(0.09)        # as tools/make_golden.py: push some points over the 0.5 threshold
frames = [synth.make_frame_pairs(1, 256, 300 + i) for i in range(12)]
h, prev = None, None
stamps = []
with torch.no_grad():
    for i, d in enumerate(frames):
        t = {k: torch.from_numpy(v).to(dev) for k, v in d.items() if k != "gt_cls"}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = net.backbone(t["pc1"], t["pc2"], t["feature1"], t["feature2"], h)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        res = net.detect_and_associate(t["pc1"], t["feature1"], out[0], out[2], out[6], prev)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        h, prev = out[1], res[5]
        stamps.append((t1 - t0, t2 - t1, len(res[6]), len(res[5])))
for i, (a, b, nc, no) in enumerate(stamps):
    print("frame %2d: backbone %6.2f ms (eager, not graphed)  detect+associate %7.2f ms  clusters %d  tracked %d" % (i, a * 1e3, b * 1e3, nc, no))
