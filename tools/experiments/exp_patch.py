#!/usr/bin/env python3
"""Ad-hoc: standalone time of rtk_patch_cost at B=64, N=256."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ratrack_amd import _lib, train_ops, fused
dev = "cuda"; B, n = 64, 256
g = torch.Generator(dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g)
xyz = r(B, n, 3).contiguous(); knn = torch.randint(0, n, (B, n, 16), device=dev, generator=g); feat = r(B * n, 256)
wn, keep = train_ops._weightnet_images(r(8, 3), r(8), r(8, 8), r(8), r(256, 8), r(256))
out = torch.empty(B * n, 256, device=dev); st = torch.cuda.current_stream().cuda_stream
f = lambda: _lib.call("rtk_patch_cost", B, n, xyz.data_ptr(), knn.data_ptr(), feat.data_ptr(), 256, wn, out.data_ptr(), 256, 0, st)
for _ in range(5): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): f()
e1.record(); torch.cuda.synchronize()
print("patch_cost %.1f us" % (e0.elapsed_time(e1) * 20))
# backward
M = B * n * 16
dout = r(B * n, 256)
wct = keep[-1] if isinstance(keep, (list, tuple)) else None
outs, _k = wn, keep
import ctypes
big = torch.empty(2, M, 256, device=dev); dt2 = torch.empty(M, 8, device=dev); d4 = torch.empty(M, 4, device=dev)
wct_t = _k[0][5] if isinstance(_k, tuple) else None
fb = lambda: _lib.call("rtk_patch_cost_bwd", B, n, xyz.data_ptr(), knn.data_ptr(), feat.data_ptr(), 256, wn, wct_t.data_ptr(), dout.data_ptr(), 256,
                       big[0].data_ptr(), big[1].data_ptr(), dt2.data_ptr(), d4.data_ptr(), None, st)
for _ in range(5): fb()
torch.cuda.synchronize()
e0.record()
for _ in range(50): fb()
e1.record(); torch.cuda.synchronize()
print("patch_cost_bwd %.1f us" % (e0.elapsed_time(e1) * 20))
# gather form of the feature gradient
from ratrack_amd.benchutil import time_graph
t2 = torch.empty(M, 8, device=dev)
fb2 = lambda: _lib.call("rtk_patch_cost_bwd", B, n, xyz.data_ptr(), knn.data_ptr(), feat.data_ptr(), 256, wn, wct_t.data_ptr(), dout.data_ptr(), 256,
                        None, big[1].data_ptr(), dt2.data_ptr(), d4.data_ptr(), t2.data_ptr(), torch.cuda.current_stream().cuda_stream)
print("patch_cost_bwd (no dxg, +t2) %.1f us" % (time_graph(fb2, 10) * 1e3))
k32 = knn.to(torch.int32)
off = torch.empty(B, n + 1, dtype=torch.int32, device=dev); inv = torch.empty(B, 16 * n, dtype=torch.int16, device=dev)
fi = lambda: _lib.call("rtk_group_inverse_index", B, n, 16 * n, k32.data_ptr(), off.data_ptr(), inv.data_ptr(), torch.cuda.current_stream().cuda_stream)
print("inverse index %.1f us" % (time_graph(fi, 10) * 1e3))
wc, bc = r(256, 8), r(256)
dfeat = torch.empty(B * n, 256, device=dev)
fg = lambda: _lib.call("rtk_patch_dfeat_gather", B, n, off.data_ptr(), inv.data_ptr(), t2.data_ptr(), wc.data_ptr(), bc.data_ptr(), dout.data_ptr(), 256,
                       dfeat.data_ptr(), torch.cuda.current_stream().cuda_stream)
print("dfeat gather %.1f us" % (time_graph(fg, 10) * 1e3))
fs = lambda: _lib.call("rtk_scatter_add_rows", B, n * 16, n, 256, knn.data_ptr(), big[0].data_ptr(), dfeat.data_ptr(), torch.cuda.current_stream().cuda_stream)
print("scatter_add_rows %.1f us;  knn.to(int32) %.1f us" % (time_graph(fs, 10) * 1e3, time_graph(lambda: knn.to(torch.int32), 10) * 1e3))
