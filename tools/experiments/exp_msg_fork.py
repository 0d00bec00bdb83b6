"""Pipelined forward at B=64, N=256 with the two scales of every MSG level on two streams (graph branches) or one after the other, at
several pipeline depths.  python tools/experiments/exp_msg_fork.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ratrack_amd import fused, synth
from ratrack_amd.track4d import Args, Track4D

dev = torch.device("cuda")
net = Track4D(Args()).to(dev).eval()
synth.fill_state_dict(net.state_dict())
batches = []
for i in range(8):
    d = synth.make_frame_pairs(64, 256, 1000 + 100 * i)
    batches.append([torch.from_numpy(d[k]).to(dev) for k in ("pc1", "pc2", "feature1", "feature2")] + [torch.zeros(5, 64, 128, device=dev)])


def run(fork, depth, steps=1500):
    fused.MSG_FORK = fork
    with torch.no_grad():
        net.invalidate_fused()
        net.backbone(*batches[0])
        pipe = fused.GraphPipeline(net._fused, tuple(batches[0]), depth=depth)
        for i in range(400):
            pipe.submit(*batches[i % 8])
        pipe.drain(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            pipe.submit(*batches[i % 8])
        pipe.drain(); torch.cuda.synchronize()
        el = time.perf_counter() - t0
    return el / steps * 1e3


for rep in range(2):
    for depth in (2, 3, 4):
        a, b = run(False, depth), run(True, depth)
        print("depth %d: one stream %.4f ms/batch (%.1f k pairs/s)   forked scales %.4f ms/batch (%.1f k pairs/s)" % (depth, a, 64 / a, b, 64 / b), flush=True)
