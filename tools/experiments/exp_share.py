#!/usr/bin/env python3
"""Round 5: pipelined forward throughput (B = 64, N = 256) against the cost volume's share of the CUs and the pipeline depth, all in
one process (boxes differ by +-3 %: only numbers of one call compare).  python tools/experiments/exp_share.py [--seconds 1.5]"""
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
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--shares", default="0,224,208,192,176,160,144,128")
    ap.add_argument("--depths", default="4,3,5,6")
    a = ap.parse_args()
    dev = "cuda"
    net = Track4D(Args()).to(dev).eval()
    synth.fill_state_dict(net.state_dict())
    net.invalidate_fused()
    B, N = a.batch, 256
    batches = []
    for k in range(8):
        d = synth.make_frame_pairs(B, N, 100 + k)
        batches.append([torch.from_numpy(d[x]).to(dev) for x in ("pc1", "pc2", "feature1", "feature2")])
    h = torch.zeros(5, B, 128, device=dev)
    with torch.no_grad():
        net.backbone(*batches[0], h)
        eng = net._fused_engine()

        def run(depth, share):
            saved = fused.cv_shared_workgroups
            if share is not None:
                fused.cv_shared_workgroups = lambda *args: share
            try:
                pipe = fused.GraphPipeline(eng, (*batches[0], h), depth=depth)
            finally:
                fused.cv_shared_workgroups = saved
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

        for s in [int(x) for x in a.shares.split(",")]:
            v, ms = run(4, s)
            print("depth 4, cost volume on %3s workgroups: %8.0f pairs/s  %.4f ms/step" % (s if s else "all", v, ms), flush=True)
        for dp in [int(x) for x in a.depths.split(",")]:
            v, ms = run(dp, None)
            print("depth %d, default share:                  %8.0f pairs/s  %.4f ms/step" % (dp, v, ms), flush=True)


if __name__ == "__main__":
    main()
