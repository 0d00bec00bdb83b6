#!/usr/bin/env python3
"""Round 5: does the cost volume start sooner from a high-priority stream?  Pipelined forward (B = 64, N = 256, depth 4) with the
capture cut around the cost volume (two graphs + one eager launch per step) and that launch on: the batch's own stream / one shared
normal-priority stream / one shared high-priority stream / a high-priority stream per batch; with the kernel on 192 and on all 256
workgroups.  One process: only numbers of one call compare.  python tools/experiments/exp_priority.py [--seconds 1.5]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from ratrack_amd import fused, synth
from ratrack_amd.track4d import Args, Track4D


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=1.5)
    a = ap.parse_args()
    dev = "cuda"
    net = Track4D(Args()).to(dev).eval()
    synth.fill_state_dict(net.state_dict())
    net.invalidate_fused()
    B, N = 64, 256
    batches = []
    for k in range(8):
        d = synth.make_frame_pairs(B, N, 100 + k)
        batches.append([torch.from_numpy(d[x]).to(dev) for x in ("pc1", "pc2", "feature1", "feature2")])
    h = torch.zeros(5, B, 128, device=dev)
    lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
    print("stream priority range (least, greatest): %s %s" % (lo, hi))
    with torch.no_grad():
        net.backbone(*batches[0], h)
        eng = net._fused_engine()

        def run(split, mode, share):
            saved = fused.cv_shared_workgroups
            if share is not None:
                fused.cv_shared_workgroups = lambda *args: share
            try:
                pipe = fused.GraphPipeline(eng, (*batches[0], h), depth=4, split_cost_volume=split)
            finally:
                fused.cv_shared_workgroups = saved
            if mode == "shared":
                st = torch.cuda.Stream()
                for e in pipe.engines:
                    e.cv_stream = st
            elif mode == "shared-high":
                st = torch.cuda.Stream(priority=-1)
                for e in pipe.engines:
                    e.cv_stream = st
            elif mode == "own-high":
                for e in pipe.engines:
                    e.cv_stream = torch.cuda.Stream(priority=-1)
            i = 0
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.4:
                pipe.submit(*batches[i % 8], h); i += 1
            pipe.drain(); torch.cuda.synchronize()
            n = 0
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < a.seconds:
                for _ in range(32):
                    pipe.submit(*batches[i % 8], h); i += 1; n += 1
            pipe.drain(); torch.cuda.synchronize()
            el = time.perf_counter() - t0
            return B * n / el, el / n * 1e3

        for share in (None, 0, 224):
            for split, mode in ((False, "-"), (True, "batch stream"), (True, "shared"), (True, "shared-high"), (True, "own-high")):
                v, ms = run(split, mode, share)
                print("cost volume on %-7s workgroups, %-22s %-14s %8.0f pairs/s  %.4f ms/step"
                      % ("default" if share is None else (share or "all"), "one graph" if not split else "two graphs + launch:", mode, v, ms), flush=True)


if __name__ == "__main__":
    main()
