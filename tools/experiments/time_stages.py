#!/usr/bin/env python3
"""Ad-hoc per-stage timing on the GPU (not part of the product): native ops at bench shapes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ratrack_amd import synth, pointnet2_utils as PU
from ratrack_amd.track4d import Track4D, Args

dev = "cuda"
B, N = int(os.environ.get("B", 64)), int(os.environ.get("N", 256))


def timeit(name, fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%-40s %9.1f us" % (name, e0.elapsed_time(e1) * 1000 / iters), flush=True)


d = synth.make_frame_pairs(B, N, 0)
pc1 = torch.from_numpy(d["pc1"]).to(dev); pc2 = torch.from_numpy(d["pc2"]).to(dev)
f1 = torch.from_numpy(d["feature1"]).to(dev); f2 = torch.from_numpy(d["feature2"]).to(dev)
xyz = pc1.permute(0, 2, 1).contiguous()
xyz2 = torch.cat([xyz, pc2.permute(0, 2, 1).contiguous()], 0)
timeit("fps %dx%d->512" % (2 * B, N), lambda: PU.furthest_point_sample(xyz2, 512))
idx = PU.furthest_point_sample(xyz2, 512)
l1 = PU.gather_operation(xyz2.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
timeit("fps %dx512->512" % (2 * B), lambda: PU.furthest_point_sample(l1, 512))
for r, ns in [(2, 4), (4, 8), (8, 16), (16, 32)]:
    timeit("ball_query r=%d ns=%d (n=512)" % (r, ns), lambda: PU.ball_query(float(r), ns, l1, l1))
timeit("three_nn 512x512", lambda: PU.three_nn(l1, l1))
timeit("knn_point k=16", lambda: PU.knn_point(16, xyz, xyz))
net = Track4D(Args()).to(dev).eval()
synth.fill_state_dict(net.state_dict())
net._use_fused = False
with torch.no_grad():
    timeit("backbone (module path) B=%d" % B, lambda: net.backbone(pc1, pc2, f1, f2, None), iters=5)
    t0 = time.time(); net.backbone(pc1, pc2, f1, f2, None); torch.cuda.synchronize(); print("one fwd wall %.1f ms" % ((time.time() - t0) * 1e3))
