"""Experiment: per-point layer kernels (rtk_pw_conv forward / input gradient, rtk_pw_wgrad) at the train-step shapes: microseconds
per launch next to the HBM floor (compulsory bytes / 8 TB/s) and the MFMA floor (flops / 157.3 TFLOP/s)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ratrack_amd import _lib, train_ops as T


from ratrack_amd.benchutil import time_graph as _time
dev = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for S, P, cins, co in [(2 * B, 256, [64, 64], 128), (2 * B, 256, [128, 32], 128), (2 * B, 256, [128], 128), (2 * B, 256, [64], 96),
                       (2 * B, 256, [96], 192), (2 * B, 256, [128], 64), (B, 256, [2, 256, 256], 32), (B, 256, [256], 256),
                       (B, 256, [128], 256), (B, 256, [128], 128), (B, 256, [128], 64), (B, 256, [64], 32), (B, 256, [32], 3),
                       (2 * B, 256, [2], 32)]:
    K = sum(cins)
    srcs = [torch.randn(S, c, P, device=dev) for c in cins]
    W = torch.randn(co, K, device=dev)
    out = torch.empty(S, co, P, device=dev)
    cols = T._pw_cols(srcs, None)
    sums = torch.zeros(2 * co * 16, dtype=torch.float64, device=dev)
    t_f = _time(lambda: T._pw_forward(srcs, cols, W, None, out), 30)
    t_s = _time(lambda: T._pw_forward(srcs, cols, W, None, out, sums=sums), 30)
    dz = torch.randn(S, co, P, device=dev)
    dW = torch.zeros(co, K, device=dev)
    ws = torch.empty(4 << 20, device=dev)
    t_w = _time(lambda: _lib.call("rtk_pw_wgrad", S, P, T._pw_operands([dz], [0]), len(srcs), T._pw_operands(srcs, cols), dW.data_ptr(),
                                  dW.stride(0), None, ws.data_ptr(), ws.numel(), T._stream()), 30)
    outs = [torch.empty_like(s) for s in srcs]
    t_d = _time(lambda: _lib.call("rtk_pw_conv", S, P, 1, T._pw_operands([dz], [0]), len(outs), T._pw_operands(outs, cols), W.data_ptr(),
                                  W.stride(0), 1, None, 0, None, 1, None, 0, T._stream()), 30)
    gf = 2.0 * S * P * K * co / 1e9
    by = 4.0 * S * P * (K + co)
    print("S=%3d P=%d %-14s -> %3d : floors hbm %5.1f mfma %5.1f us | fwd %5.1f  fwd+stats %5.1f  wgrad %5.1f  dgrad %5.1f us"
          % (S, P, cins, co, by / 8e6, gf / 157.3e-3, t_f * 1e3, t_s * 1e3, t_w * 1e3, t_d * 1e3))
