"""rtk_three_interpolate_grad_set at the train-step shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ratrack_amd import _lib, train_ops as T  # noqa
from ratrack_amd.benchutil import time_graph
dev = "cuda"
for S, C, n, m in [(128, 64, 256, 256), (128, 128, 256, 256), (64, 128, 256, 256), (64, 64, 256, 256)]:
    g = torch.Generator(dev).manual_seed(0)
    go = torch.randn(S, C, n, device=dev, generator=g)
    idx = torch.randint(0, m, (S, n, 3), device=dev, generator=g, dtype=torch.int32)
    w = torch.rand(S, n, 3, device=dev, generator=g)
    out = torch.empty(S, C, m, device=dev)
    ms = time_graph(lambda: _lib.call("rtk_three_interpolate_grad_set", S, C, n, m, go.data_ptr(), idx.data_ptr(), w.data_ptr(), out.data_ptr(),
                                      T._stream()), 10)
    ref = torch.zeros(S, C, m, device=dev, dtype=torch.float64)
    for k in range(3):
        ref.scatter_add_(2, idx[:, :, k].long().unsqueeze(1).expand(-1, C, -1), (go * w[:, :, k].unsqueeze(1)).double())
    off = torch.empty(S, m + 1, dtype=torch.int32, device=dev); inv = torch.empty(S, 3 * n, dtype=torch.int16, device=dev)
    ms_b = time_graph(lambda: _lib.call("rtk_group_inverse_index", S, m, 3 * n, idx.data_ptr(), off.data_ptr(), inv.data_ptr(), T._stream()), 10)
    out2 = torch.empty(S, C, m, device=dev)
    ms_g = time_graph(lambda: _lib.call("rtk_three_interpolate_grad_gather", S, C, n, m, go.data_ptr(), w.data_ptr(), off.data_ptr(),
                                        inv.data_ptr(), out2.data_ptr(), T._stream()), 10)
    print("S=%d C=%d n=%d m=%d: scatter (LDS atomics) %.1f us (%.2f TB/s) | inverse table %.1f us + gather %.1f us (%.2f TB/s)  max err %.1e / %.1e"
          % (S, C, n, m, ms * 1e3, (go.numel() + out.numel()) * 4 / ms / 1e9, ms_b * 1e3, ms_g * 1e3, (go.numel() + out.numel()) * 4 / ms_g / 1e9,
             float((out.double() - ref).abs().max()), float((out2.double() - ref).abs().max())))
