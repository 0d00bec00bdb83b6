"""The backbone at the reference's operating point (B = 1): latency of one captured batch alone, and per pair with four in flight.
python tools/experiments/time_b1.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ratrack_amd import fused, synth
from ratrack_amd.track4d import Args, Track4D

dev = torch.device("cuda")
net = Track4D(Args()).to(dev).eval()
synth.fill_state_dict(net.state_dict())
for B in (1, 8):
    d = synth.make_frame_pairs(B, 256, 1000)
    t = [torch.from_numpy(d[k]).to(dev) for k in ("pc1", "pc2", "feature1", "feature2")] + [torch.zeros(5, B, 128, device=dev)]
    with torch.no_grad():
        net.invalidate_fused()
        net.backbone(*t)
        for depth in (1, 4):
            pipe = fused.GraphPipeline(net._fused, tuple(t), depth=depth)
            for _ in range(200):
                pipe.submit(*t)
            pipe.drain(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 2000
            for _ in range(n):
                pipe.submit(*t)
            pipe.drain(); torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / n * 1e3
            print("B=%d, %d batch(es) in flight: %.3f ms per batch = %.1f k pairs/s" % (B, depth, ms, B / ms), flush=True)
            pipe = None
