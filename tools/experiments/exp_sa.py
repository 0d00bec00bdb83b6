#!/usr/bin/env python3
"""Micro-benchmark of rtk_sa_scale (isolated, back-to-back) on real geometry at B'=128, N=256."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ratrack_amd import fused as F, synth
from ratrack_amd.track4d import Track4D, Args
dev = "cuda"
net = Track4D(Args()).to(dev).eval(); synth.fill_state_dict(net.state_dict())
eng = F.FusedBackbone(net)
d = synth.make_frame_pairs(64, 256, 0)
xyz = torch.cat([torch.from_numpy(d["pc1"]), torch.from_numpy(d["pc2"])]).permute(0, 2, 1).contiguous().to(dev)
geo = F.Geometry(xyz, 512)
torch.cuda.synchronize()
def timeit(fn, iters=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / iters
W = eng.enc
S_ = 128
for lvl, s, qc in [(0, 0, 32), (0, 1, 32), (1, 0, 96), (1, 1, 96), (2, 0, 192), (2, 1, 192)]:
    sc = W.scales[lvl][s]
    nsrc = geo.xyz[lvl].shape[1]
    q = torch.randn(S_ * nsrc, qc, device=dev)
    out = torch.empty(S_ * 512, 128, device=dev)
    t = timeit(lambda: F.sa_scale(geo, W, lvl, s, q, 0, out, 0))
    widths = [sc.c1] + [int(l.cout16) * 16 for l in sc.chain.arr]
    macs = 256 * S_ * sc.nsample * (4 * widths[0] + sum(a * b for a, b in zip(widths[:-1], widths[1:])))
    print("level %d scale %d ns=%2d widths %s: %6.1f us  %5.1f TF/s (live centroids only)" % (lvl, s, sc.nsample, widths, t, 2 * macs / t / 1e6), flush=True)
