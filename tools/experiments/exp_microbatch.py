#!/usr/bin/env python3
"""Experiment: run the fused backbone as M micro-batches on M streams (latency-bound phases of one overlap dense phases of another)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ratrack_amd import synth
from ratrack_amd.track4d import Track4D, Args
from ratrack_amd import fused as F

dev = "cuda"
B, N = 64, 256
net = Track4D(Args()).to(dev).eval()
synth.fill_state_dict(net.state_dict())
d = synth.make_frame_pairs(B, N, 0)
t = {k: torch.from_numpy(v).to(dev) for k, v in d.items() if k != "gt_cls"}
h = torch.zeros(5, B, 128, device=dev)
for M in (1, 2, 4):
    engs = [F.FusedBackbone(net) for _ in range(M)]
    streams = [torch.cuda.Stream() for _ in range(M)]
    sl = [slice(i * B // M, (i + 1) * B // M) for i in range(M)]
    def step():
        cur = torch.cuda.current_stream()
        for e, s, q in zip(engs, streams, sl):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                e.backbone(t["pc1"][q], t["pc2"][q], t["feature1"][q], t["feature2"][q], h[:, q])
        for s in streams:
            cur.wait_stream(s)
    with torch.no_grad():
        for _ in range(5): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): step()
        torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 30
    print("micro-batches=%d: %.3f ms/step  %.0f pairs/s" % (M, el * 1e3, B / el), flush=True)
