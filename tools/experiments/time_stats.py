#!/usr/bin/env python3
"""Round 5: what do the order-independent batch sums cost?  Producer (rtk_bn_train_stats) and consumer (rtk_bn_relu_fwd_fin) of a
per-point BatchNorm layer timed alone, at the train step's typical shapes.  Run from a tree with the float64-atomic sums and from one
with the fixed-point sums on the same box:  python tools/experiments/time_stats.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

from ratrack_amd import _lib, train_ops as T


def main():
    dev = "cuda"
    st = lambda: torch.cuda.current_stream().cuda_stream
    for (S_, C, rows, ns) in ((128, 64, 256, 1), (128, 128, 256, 1), (64, 32, 256, 1), (128, 64, 256, 32), (128, 16, 256, 8)):
        z = torch.randn(S_, C, rows, ns, device=dev)
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        par = torch.empty(4, 1, C, device=dev)
        y = torch.empty_like(z)
        sums = torch.zeros(16, 1, C, 2, dtype=torch.float64, device=dev)
        fin = T._BnFin(sums.data_ptr(), float(S_ * rows * ns), g.data_ptr(), b.data_ptr(), 1e-5, 0.1, rm.data_ptr(), rv.data_ptr(), None, None)
        prod = lambda: _lib.call("rtk_bn_train_stats", S_, C, rows, ns, 1, z.data_ptr(), None, sums.data_ptr(), st())
        cons = lambda: _lib.call("rtk_bn_relu_fwd_fin", S_, C, rows, ns, 1, z.data_ptr(), ctypes.byref(fin), par.data_ptr(), 0, y.data_ptr(), st())
        res = []
        for fn in (prod, cons):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) * 1e3 / 200)
        print("(%d, %d, %d, %d): stats %.2f us, normalise+relu %.2f us" % (S_, C, rows, ns, res[0], res[1]), flush=True)


if __name__ == "__main__":
    main()
