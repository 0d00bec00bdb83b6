import sys, os
sys.path.insert(0, os.getcwd())
import torch
from ratrack_amd import synth, pointnet2_utils as PU
def timeit(name, fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    print("%-40s %9.1f us" % (name, e0.elapsed_time(e1) * 1000 / iters), flush=True)
for B, N in [(1, 256), (64, 256), (32, 1024)]:
    d = synth.make_frame_pairs(B, N, 0)
    xyz = torch.from_numpy(d["pc1"]).cuda().permute(0, 2, 1).contiguous()
    xyz2 = torch.cat([xyz, xyz], 0)
    timeit("fps %dx%d->512" % (2 * B, N), lambda: PU.furthest_point_sample(xyz2, 512))
