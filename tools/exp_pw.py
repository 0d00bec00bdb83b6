"""Experiment: per-point layer kernels (rtk_pw_conv forward / input gradient, rtk_pw_wgrad) at the train-step shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ratrack_amd import train_ops as T
from ratrack_amd.benchutil import _time
dev = "cuda"
for S, P, cins, co in [(128, 256, [64, 64], 128), (128, 256, [128, 32], 128), (128, 256, [128], 128), (128, 256, [32, 32], 32),
                       (128, 256, [64, 64], 64), (64, 256, [2, 256, 256], 16), (64, 256, [256], 256), (64, 256, [256], 128),
                       (64, 256, [32], 3), (128, 256, [2], 16)]:
    K = sum(cins)
    srcs = [torch.randn(S, c, P, device=dev) for c in cins]
    W = torch.randn(co, K, device=dev)
    out = torch.empty(S, co, P, device=dev)
    cols = T._pw_cols(srcs, None)
    t_f = _time(lambda: T._pw_forward(srcs, cols, W, None, out), 30)
    dz = torch.randn(S, co, P, device=dev)
    t_b = _time(lambda: T._pw_backward([True] * len(srcs), srcs, cols, W, dz, False), 30)
    gf = 2.0 * S * P * K * co / 1e9
    print("S=%d P=%d %s -> %d : fwd %.1f us (%.1f TFLOP/s)   wgrad+dgrad+fill %.1f us" % (S, P, cins, co, t_f * 1e3, gf / t_f, t_b * 1e3))
