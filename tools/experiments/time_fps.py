"""rtk_fps_centroids at the bench shape (128 clouds of 256 points, 512 centroids): microseconds per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ratrack_amd import _lib, fused, synth
dev = "cuda"
d = synth.make_frame_pairs(64, 256, 1000)
xyz = torch.cat([torch.from_numpy(d["pc1"]), torch.from_numpy(d["pc2"])], 0).permute(0, 2, 1).contiguous().to(dev)
S_, n = xyz.shape[0], xyz.shape[1]
for npoint in (2, 64, 128, 256, 512):
  if True:
    idx = torch.zeros(S_, npoint, dtype=torch.int32, device=dev); new_xyz = torch.empty(S_, npoint, 3, device=dev)
    cnt = torch.zeros(S_, dtype=torch.int32, device=dev); tie = torch.zeros(S_, dtype=torch.int32, device=dev); ft = torch.zeros(S_, dtype=torch.int32, device=dev)
    snap = torch.empty(S_ * n + S_ * 2 * npoint, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    f = lambda: _lib.call("rtk_fps_centroids", S_, n, npoint, xyz.data_ptr(), idx.data_ptr(), new_xyz.data_ptr(), cnt.data_ptr(), tie.data_ptr(), None, snap.data_ptr(), ft.data_ptr(), st)
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    print("fps_centroids: %.1f us per launch (%d clouds x %d points -> %d)" % (e0.elapsed_time(e1) / 50 * 1e3, S_, n, npoint))

