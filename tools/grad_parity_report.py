#!/usr/bin/env python3
"""Per-parameter gradient parity of the hand-written training path against the reference-generated train-step fixtures
(tests/golden/train_b8_n256.npz, real_*.npz): error vs the reference's fp32 gradients, vs the float64 arbiter, and the fp32
reference's own distance from the arbiter.  Runs on the GPU box:  python tools/grad_parity_report.py [case ...] [--fp32-cv]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import test_varn_train_gpu as T  # noqa: E402
from _util import grad_report, load_case  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--fp32-cv" in sys.argv:
        from ratrack_amd import train_ops
        train_ops.CV_SPLIT = False
    for name in args or ["train_b8_n256", "real_549_1047"]:
        case = load_case(name)
        if name.startswith("real"):
            sub = {k[len("train/"):]: v for k, v in case.items() if k.startswith("train/")}
            sub.update({k: v for k, v in case.items() if k.startswith("in_")})
        else:
            sub = case
        items, flow, cls, grads, sd = T.train_step(T.make_net(), sub)
        rows = grad_report(sub, grads)
        print("== %s: loss %s (reference %s)" % (name, [round(items[str(k)], 6) for k in sub["loss_keys"]], np.round(sub["loss_vals"], 6).tolist()))
        print("%-52s %9s %9s %9s %10s" % ("parameter", "vs ref32", "vs f64", "ref32-f64", "norm"))
        for r in sorted(rows, key=lambda r: -r["e_ref"] / (1.0 if not r["zero"] else 1e3)):
            print("%-52s %9.2e %9s %9s %10.3e%s" % (r["name"], r["e_ref"], "%.2e" % r["e_arb"] if r["e_arb"] is not None else "-",
                                                    "%.2e" % r["floor"] if r["floor"] is not None else "-", r["ref_norm"], "  zero" if r["zero"] else ""))


if __name__ == "__main__":
    main()
