"""Experiment: inference engine with / without the forked geometry stream, captured graphs, 1 and 2 batches in flight."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ratrack_amd import synth, fused
from ratrack_amd.track4d import Track4D, Args
dev = "cuda"; B = 64
net = Track4D(Args()).to(dev).eval(); synth.fill_state_dict(net.state_dict())
d = synth.make_frame_pairs(B, 256, 0); t = {k: torch.from_numpy(v).to(dev) for k, v in d.items()}
h = torch.zeros(5, B, 128, device=dev)
inp = (t["pc1"], t["pc2"], t["feature1"], t["feature2"], h)
for side in (True, False):
    for depth in (1, 2):
        net._fused = None
        with torch.no_grad():
            eng = fused.FusedBackbone(net)
            eng.use_side_stream = side
            pipe = fused.GraphPipeline(eng, inp, depth=depth)
            for e in pipe.engines: assert e.use_side_stream == side
            for _ in range(10): pipe.submit(*inp)
            pipe.drain(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100): pipe.submit(*inp)
            pipe.drain(); torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 10
        print("side stream %-5s depth %d: %.4f ms/step  %.0f pairs/s" % (side, depth, ms, B / ms * 1e3))
