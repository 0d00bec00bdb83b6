#!/usr/bin/env python3
"""Time of rtk_split_mlp2 over the positions of the B=64 cost volume (library from RTK_SO_PATH).  tools/exp_split_time.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from ratrack_amd import _lib, fused as F, benchutil as BU
dev = "cuda"
img = torch.cat([F.pack_layer_split(torch.randn(256, 256, device=dev) / 16) for _ in range(2)])
b = torch.zeros(256, device=dev)
npos = 64 * 256 * 16
x = torch.randn(npos, 256, device=dev); y = torch.empty_like(x)
st = lambda: torch.cuda.current_stream().cuda_stream
t = BU.time_graph(lambda: _lib.call("rtk_split_mlp2", npos, x.data_ptr(), img.data_ptr(), b.data_ptr(), b.data_ptr(), y.data_ptr(), st()), 10)
print("%s: %.1f us" % (os.environ.get("RTK_SO_PATH", "default")[-24:], t * 1e3))
