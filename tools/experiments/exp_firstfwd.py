"""rtk_sa_first_layer (first SharedMLP layer straight from the per-point projection, + batch sums) at the train-step shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ratrack_amd import _lib, synth, train_ops as T
from ratrack_amd.train_path import TrainGeometry
from ratrack_amd.benchutil import time_graph
dev = "cuda"; B = 64
d = synth.make_frame_pairs(B, 256, 1000)
xyz = torch.cat([torch.from_numpy(d["pc1"]), torch.from_numpy(d["pc2"])]).permute(0, 2, 1).contiguous().to(dev)
tg = TrainGeometry(xyz, 512)
C1s = [[16, 16], [32, 32], [64, 64]]
tot = 0.0
for lvl in range(3):
    for s in range(2):
        idx, dxyz = tg.ball[lvl][s], tg.dxyz[lvl][s]
        S_, rows, ns = idx.shape
        C, n_src = C1s[lvl][s], 256
        proj = torch.randn(S_, C, n_src, device=dev); wx = torch.randn(C, 3, device=dev)
        z = torch.empty(S_, C, rows, ns, device=dev)
        sums = torch.zeros(8, 2, C, 2, dtype=torch.float64, device=dev)
        ms = time_graph(lambda: _lib.call("rtk_sa_first_layer", S_, C, rows, ns, 2, n_src, proj.data_ptr(), idx.data_ptr(), dxyz.data_ptr(),
                                          wx.data_ptr(), 3, tg.row_w[lvl].data_ptr(), z.data_ptr(), sums.data_ptr(), T._stream()), 10)
        tot += ms
        print("lvl %d scale %d: C=%d ns=%d z %.0f MB: %.1f us (%.2f TB/s)" % (lvl, s, C, ns, z.numel() * 4 / 1e6, ms * 1e3, z.numel() * 4 / ms / 1e9))
print("total %.1f us" % (tot * 1e3))
