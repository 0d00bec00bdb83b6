#!/usr/bin/env python3
"""Split-bf16 matrix path (csrc/split_mfma.h): device packer against the host restatement, two 256x256 layers against float64,
and the time of the two layers over the positions of the B=64 cost volume.  python tools/exp_split.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from ratrack_amd import _lib, fused as F, benchutil as BU
dev = "cuda"
torch.manual_seed(0)
W = [torch.randn(256, 256, device=dev) / 16 for _ in range(2)]
b = [torch.randn(256, device=dev) * 0.1 for _ in range(2)]
img = torch.empty(2 * 256 * 256 * 3, dtype=torch.int16, device=dev)
st = lambda: torch.cuda.current_stream().cuda_stream
for l in range(2):
    _lib.call("rtk_pack_split_layer", 256, 256, W[l].data_ptr(), 0, img[l * 196608:].data_ptr(), st())
ref_img = torch.cat([F.pack_layer_split(w) for w in W])
print("device packer == host packer:", torch.equal(img, ref_img))
for npos in (77, 128, 4096):
    x = torch.randn(npos, 256, device=dev)
    y = torch.full((npos, 256), float("nan"), device=dev)
    _lib.call("rtk_split_mlp2", npos, x.data_ptr(), img.data_ptr(), b[0].data_ptr(), b[1].data_ptr(), y.data_ptr(), st())
    lk = torch.nn.functional.leaky_relu
    r64 = lk(lk(x.double() @ W[0].double().T + b[0].double(), 0.1) @ W[1].double().T + b[1].double(), 0.1)
    r32 = lk(lk(x @ W[0].T + b[0], 0.1) @ W[1].T + b[1], 0.1)
    e = lambda a: float((a.double() - r64).abs().max() / r64.abs().max())
    print("positions %5d: split err %.3e   torch fp32 err %.3e" % (npos, e(y), e(r32)))
npos = 64 * 256 * 16
x = torch.randn(npos, 256, device=dev); y = torch.empty_like(x)
t = BU.time_graph(lambda: _lib.call("rtk_split_mlp2", npos, x.data_ptr(), img.data_ptr(), b[0].data_ptr(), b[1].data_ptr(), y.data_ptr(), st()), 20)
fl = 2 * 2 * 256 * 256 * npos
print("two layers over %d positions: %.1f us  = %.1f TFLOP/s fp32-equivalent (%.1f bf16 TFLOP/s executed)" % (npos, t * 1e3, fl / t / 1e9, 6 * fl / t / 1e9))
