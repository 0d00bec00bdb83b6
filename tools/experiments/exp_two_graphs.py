#!/usr/bin/env python3
"""Experiment: replay K captured backbone graphs round-robin on K streams (batch-level pipelining)."""
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
with torch.no_grad():
    for K in (1, 2, 3):
        engs = [F.FusedBackbone(net) for _ in range(K)]
        for e in engs:
            e.backbone(t["pc1"], t["pc2"], t["feature1"], t["feature2"], h)
        steps = [e.capture(t["pc1"], t["pc2"], t["feature1"], t["feature2"], h) for e in engs]
        streams = [torch.cuda.Stream() for _ in range(K)]
        def run(n):
            cur = torch.cuda.current_stream()
            for s in streams: s.wait_stream(cur)
            for i in range(n):
                with torch.cuda.stream(streams[i % K]):
                    steps[i % K].graph.replay()
            for s in streams: cur.wait_stream(s)
        run(6); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(60); torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 60
        print("graphs in flight=%d: %.3f ms/step  %.0f pairs/s" % (K, el * 1e3, B / el), flush=True)
