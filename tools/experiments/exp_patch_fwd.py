#!/usr/bin/env python3
"""rtk_patch_cost alone at the bench shape (library from RTK_SO_PATH).  python tools/exp_patch_fwd.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from ratrack_amd import _lib, fused as F, benchutil as BU, synth, pointnet2_utils as PU
from ratrack_amd.track4d import Args, Track4D
net = Track4D(Args()).to("cuda").eval(); synth.fill_state_dict(net.state_dict())
eng = F.FusedBackbone(net)
B, N = 64, 256
torch.manual_seed(0)
xyz = torch.randn(B, N, 3, device="cuda"); knn = PU.knn_point(16, xyz, xyz)
feat = torch.randn(B * N, 256, device="cuda"); out = torch.empty(B * N, 256, device="cuda")
st = lambda: torch.cuda.current_stream().cuda_stream
f = lambda: _lib.call("rtk_patch_cost", B, N, xyz.data_ptr(), knn.data_ptr(), feat.data_ptr(), 256, eng.wn2.arr, out.data_ptr(), 256, 0, st())
print("%s: patch_cost %.1f us" % (os.environ.get("RTK_SO_PATH", "default")[-22:], BU.time_graph(f, 20) * 1e3))
