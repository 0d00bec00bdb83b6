#!/usr/bin/env python3
"""rtk_three_nn at the forward's shape (128 clouds, 512 unknown vs 512 known, ~256 distinct each) from a replayed graph.
python tools/exp_three_nn.py   (library from RTK_SO_PATH)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from ratrack_amd import _lib, benchutil as BU, pointnet2_hip  # noqa: F401
B, n, m = 128, 512, 512
torch.manual_seed(0)
base = torch.randn(B, 256, 3, device="cuda")
unk = torch.cat([base, base[:, :1].expand(B, 256, 3)], 1).contiguous()      # 256 distinct points + copies of point 0, as after FPS exhaustion
kn = unk.clone()
d2 = torch.empty(B, n, 3, device="cuda"); idx = torch.empty(B, n, 3, dtype=torch.int32, device="cuda")
nu = torch.full((B,), 256, dtype=torch.int32, device="cuda")
st = lambda: torch.cuda.current_stream().cuda_stream
_lib.SIGNATURES.setdefault("rtk_three_nn_masked", [_lib._c_int] * 3 + [_lib._c_void_p] * 6 + [_lib._c_void_p])
f = lambda: _lib.call("rtk_three_nn_masked", B, n, m, unk.data_ptr(), kn.data_ptr(), d2.data_ptr(), idx.data_ptr(), nu.data_ptr(), nu.data_ptr(), st())
print("%s: three_nn %.2f us" % (os.environ.get("RTK_SO_PATH", "default")[-20:], BU.time_graph(f, 20) * 1e3))
