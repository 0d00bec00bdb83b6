#!/usr/bin/env python3
"""Ad-hoc timing of the host-side pieces of the cost-volume backward at bench shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch


def _tall_tn(a, b):
    """a^T b for tall-skinny a (M,p), b (M,q) with M >> p, q: the (p,q) output is a handful of GEMM tiles, so a plain mm
    walks the M dimension in one or two workgroups.  Split M into up to 256 slabs (batched GEMM), then add the slabs."""
    M = a.shape[0]
    c = 256
    while M % c:
        c //= 2
    if c == 1:
        return torch.mm(a.t(), b)
    return torch.bmm(a.view(c, M // c, -1).transpose(1, 2), b.view(c, M // c, -1)).sum(0)


M = 64 * 256 * 16
dev = "cuda"
def timeit(name, fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    print("%-50s %9.1f us" % (name, e0.elapsed_time(e1) * 1000 / iters), flush=True)
dz = torch.randn(M, 256, device=dev); a = torch.randn(M, 256, device=dev)
ap = torch.randn(M, 272, device=dev)
t2p = torch.randn(M, 16, device=dev); t8 = torch.randn(M, 8, device=dev); d3 = torch.randn(M, 3, device=dev)
wc = torch.randn(256, 8, device=dev); wb = torch.randn(8, 8, device=dev)
timeit("mm dz^T a (256 x M x 256)", lambda: torch.mm(dz.t(), a))
timeit("mm dz^T ap (256 x M x 272)", lambda: torch.mm(dz.t(), ap))
timeit("tall_tn dz, a (slabs)", lambda: _tall_tn(dz, a))
timeit("dz.sum(0)", lambda: dz.sum(0))
timeit("tall_tn dq3, t2p (N=16)", lambda: _tall_tn(dz, t2p))
timeit("mm dq3^T t2p", lambda: torch.mm(dz.t(), t2p))
timeit("tall_tn dt2, t1 (8x8)", lambda: _tall_tn(t8, t8))
timeit("tall_tn dt1, d3 (8x3)", lambda: _tall_tn(t8, d3))
timeit("mm dt2 wb (M x 8 x 8)", lambda: torch.mm(t8, wb))
timeit("dt2 * (t2>0)", lambda: t8 * (t8 > 0))
timeit("addmm relu t1", lambda: torch.relu(torch.addmm(wb[0], t8, wb.t())))
timeit("cat t2p", lambda: torch.cat([t8, torch.ones(M, 1, device=dev), torch.zeros(M, 7, device=dev)], 1))
timeit("empty big", lambda: torch.empty(6, M, 256, device=dev))
