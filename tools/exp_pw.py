#!/usr/bin/env python3
"""Micro-benchmark of rtk_pointwise_mlp shapes used by the backbone (isolated, back-to-back launches)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ratrack_amd import fused as F
dev = "cuda"
torch.manual_seed(0)

def timeit(fn, iters=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / iters

def chain(cin, widths):
    layers, cur = [], cin
    for w in widths:
        layers.append((torch.randn(w, cur, dtype=torch.float64) / cur ** 0.5, torch.zeros(w, dtype=torch.float64), F.ACT_RELU))
        cur = w
    return F.Chain(layers, dev)

S, n = 128, 512
rows = S * n
for name, cin, widths in [("t1 64->96", 64, [96]), ("t2 96->192", 96, [192]), ("l3 128->64", 128, [64]), ("fp-like 128->128", 128, [128]),
                          ("p1 128->256 (16384 rows)", 128, [256]), ("cls head 256->128->64->32->1 (16384 rows)", 256, [128, 64, 32, 1])]:
    r = 16384 if "16384" in name else rows
    rps = 256 if "16384" in name else n
    x = torch.randn(r, cin, device=dev)
    ch = chain(cin, widths)
    out = torch.empty(r, F.ceil16(widths[-1]), device=dev)
    t_full = timeit(lambda: F.pointwise(r, rps, [(x, cin, False)], ch, out))
    nu = torch.full((r // rps,), rps // 2, dtype=torch.int32, device=dev)
    t_half = timeit(lambda: F.pointwise(r, rps, [(x, cin, False)], ch, out, row_nuniq=nu))
    flops = 2.0 * r * sum(a * b for a, b in zip([cin] + widths[:-1], widths))
    print("%-46s full %6.1f us (%5.1f TF/s)   half-skipped %6.1f us" % (name, t_full, flops / t_full / 1e6, t_half), flush=True)
