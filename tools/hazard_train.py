#!/usr/bin/env python3
"""Is the hand-written TRAIN step reproducible bit for bit?  (Round 4: the selection kernel's irreproducibility was traced to packed
fp32 instructions that broadcast the odd half of a register pair through op_sel -- DESIGN section 8 -- and hipcc's SLP vectoriser
emits the same form in several training kernels.)  The same forward + loss + backward from the same weights, `--iters` times; every
gradient tensor, the loss and the outputs compared bitwise with the first run.

    python tools/hazard_train.py --batch 64 --iters 40
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from ratrack_amd import synth, train_ops  # noqa: E402
from ratrack_amd.track4d import Args, Track4D  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--npoints", type=int, default=256)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--deterministic", action="store_true", help="train_ops.set_deterministic(): order-independent sums")
    a = ap.parse_args()
    from _util import reference_state_dict
    dev = "cuda"
    net = Track4D(Args()).to(dev)
    net.load_state_dict(reference_state_dict(dev), strict=True)
    net.train()
    d = synth.make_frame_pairs(a.batch, a.npoints, 2030)
    g = {k: torch.from_numpy(v).to(dev) for k, v in d.items()}
    train_ops.enable_zero_arena(torch.device(dev))
    train_ops.set_deterministic(a.deterministic)

    def step():
        net.zero_grad(set_to_none=True)
        train_ops.arena_begin_step(torch.device(dev))
        flow, h, cls, *_ = net.backbone(g["pc1"], g["pc2"], g["feature1"], g["feature2"], None)
        total, items = train_ops.backbone_loss(g["pc1"], flow, cls, g["gt_warp"], g["gt_cls"], pretrain=False)
        total.backward()
        train_ops.arena_end_step(torch.device(dev))
        out = {"loss": total.detach().clone().reshape(1), "flow": flow.detach().clone(), "cls": cls.detach().clone()}
        for k, p in net.named_parameters():
            if p.grad is not None:
                out["grad/" + k] = p.grad.detach().clone()
        return out
    step()
    step()                      # (the first steps build caches: packed weight images, the zero arena's extent)
    ref = step()
    torch.cuda.synchronize()
    bad, worst = {}, {}
    for it in range(a.iters):
        cur = step()
        for k, v in cur.items():
            if not torch.equal(v.view(torch.int32), ref[k].view(torch.int32)):
                bad[k] = bad.get(k, 0) + 1
                worst[k] = max(worst.get(k, 0.0), float((v - ref[k]).abs().max() / ref[k].abs().max().clamp_min(1e-30)))
    print("train step B=%d N=%d%s, %d repetitions: %s" % (a.batch, a.npoints, ", deterministic mode" if a.deterministic else "", a.iters,
          "every gradient tensor, the loss and the outputs bit-identical" if not bad else
          "%d of %d tensors differed at least once; outputs: loss %s flow %s cls %s; largest relative difference of any tensor in any "
          "repetition %.1e; tensors ever beyond 1e-4: %s; the five largest: %s"
          % (len(bad), len(ref), bad.get("loss", 0), bad.get("flow", 0), bad.get("cls", 0), max(worst.values()),
             {k: "%.1e" % v for k, v in worst.items() if v > 1e-4},
             {k: "%.1e" % v for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:5]})))
    if bad:
        print("tensors that never differed: %s" % ", ".join(sorted(k for k in ref if k not in bad)))
        print("tensors that differed (repetitions of %d): %s" % (a.iters, ", ".join("%s %d" % (k, v) for k, v in sorted(bad.items()))))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
