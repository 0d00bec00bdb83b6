#!/usr/bin/env python3
"""rtk_tn_gemm256_split (x^T y over m rows, split-bf16 matrix path) against float64 and against the batched library GEMM it replaces."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ratrack_amd import train_ops as T

dev = "cuda"


def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def bmm_way(big, acts, M):
    c = 64
    while M % c:
        c //= 2
    prod = torch.bmm(big.view(2 * c, M // c, 256).transpose(1, 2), acts.view(2 * c, M // c, 256))
    return prod.view(2, c, 256, 256).sum(1)


for M in (64 * 256 * 16, 8 * 256 * 16, 256 * 16, 322 * 16, 1000):
    g = torch.Generator(dev).manual_seed(M)
    big = torch.randn(2, M, 256, device=dev, generator=g) * torch.rand(2, M, 1, device=dev, generator=g)
    acts = torch.relu(torch.randn(2, M, 256, device=dev, generator=g))
    out = T.tn_gemm256([(big[0], acts[0]), (big[1], acts[1])])
    ref = torch.stack([big[k].double().t() @ acts[k].double() for k in range(2)])
    lib = bmm_way(big, acts, M) if M % 16 == 0 else torch.stack([big[k].t() @ acts[k] for k in range(2)])
    sc = ref.abs().max().item()
    print("M=%7d  split: max err %.3g of max|out|   library fp32: %.3g" % (M, (out - ref).abs().max().item() / sc, (lib - ref).abs().max().item() / sc))
    t_s = timeit(lambda: T.tn_gemm256([(big[0], acts[0]), (big[1], acts[1])]))
    t_l = timeit(lambda: bmm_way(big, acts, M)) if M % 16 == 0 else float("nan")
    fl = 2 * 2 * 256 * 256 * M
    print("           split %7.1f us (%.0f TFLOP/s fp32-equivalent)   library %7.1f us (%.0f)" % (t_s, fl / t_s / 1e6, t_l, fl / t_l / 1e6))
