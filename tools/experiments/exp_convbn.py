#!/usr/bin/env python3
"""Ad-hoc timing of the fused conv+BN kernels at the largest set-abstraction shape (S=128, 64->64 channels, 256 rows x 32)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ratrack_amd import _lib, train_ops  # noqa
S_, C, rows, ns, groups = 128, 64, 256, 32, 2
dev = "cuda"
g = torch.Generator(dev).manual_seed(0)
x = torch.randn(S_, C, rows, ns, device=dev, generator=g); z = torch.empty_like(x); dz = torch.randn_like(x); dzp = torch.empty_like(x)
w = torch.randn(C, C, device=dev, generator=g) * 0.1
par = torch.rand(4, groups, C, device=dev, generator=g) + 0.5
rw = torch.ones(S_, rows, device=dev)
sums = torch.zeros(8, groups, C, 2, dtype=torch.float64, device=dev); dw = torch.zeros(C, C, device=dev); ws = torch.empty(1024 * C * C, device=dev)
st = torch.cuda.current_stream().cuda_stream
def t(name, fn, bytes_):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("%-28s %7.1f us  %5.2f TB/s" % (name, ms * 1e3, bytes_ / ms / 1e9))
T = x.numel() * 4
t("conv_bn_fwd (r x, w z)", lambda: _lib.call("rtk_conv_bn_fwd", S_, C, C, rows, ns, groups, x.data_ptr(), par.data_ptr(), w.data_ptr(), z.data_ptr(), None, rw.data_ptr(), sums.data_ptr(), st), 2 * T)
t("conv_bn_bwd stats (r dz, r z)", lambda: _lib.call("rtk_conv_bn_bwd", S_, C, C, rows, ns, groups, dz.data_ptr(), w.data_ptr(), x.data_ptr(), par.data_ptr(), rw.data_ptr(), sums.data_ptr(), 1e6, 0, None, None, st), 2 * T)
t("conv_bn_bwd apply (+w dz)", lambda: _lib.call("rtk_conv_bn_bwd", S_, C, C, rows, ns, groups, dz.data_ptr(), w.data_ptr(), x.data_ptr(), par.data_ptr(), rw.data_ptr(), sums.data_ptr(), 1e6, 1, dzp.data_ptr(), None, st), 3 * T)
t("conv_wgrad (r dz, r z)", lambda: _lib.call("rtk_conv_wgrad", S_, C, C, rows, ns, groups, dz.data_ptr(), x.data_ptr(), par.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel(), st), 2 * T)
