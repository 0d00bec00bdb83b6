#!/usr/bin/env python3
"""Which KERNEL is not reproducible?  Companion of tools/hazard_harness.py.

Records the launch list of one eager fused backbone() pass (every rtk_* call with its arguments; every buffer the pass
allocated is pinned, so the pointers stay valid), then re-issues exactly that list:
  (a) alone, on one stream, `--iters` times  -- every pinned buffer must come out bit-identical each time;
  (b) two recorded passes (different batches, disjoint buffers) replayed CONCURRENTLY on two streams -- what a GraphPipeline
      does -- and every pinned buffer of both compared with its own baseline.
A buffer that differs is attributed to the first recorded call that received a pointer into it.

    python tools/hazard_stages.py --iters 300
"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from ratrack_amd import _lib  # noqa: E402
from hazard_harness import DEV, make_net, tie_batch  # noqa: E402


class Recorder:
    """Pins every tensor allocated while active and records every _lib.call."""

    def __init__(self):
        self.kept, self.calls, self.zeroed = [], [], []

    def __enter__(self):
        self._saved = (torch.empty, torch.zeros, torch.empty_like, torch.zeros_like, torch.full, _lib.call)
        rec = self

        def keep(fn, zero=False):
            def inner(*a, **k):
                t = fn(*a, **k)
                if torch.is_tensor(t) and t.is_cuda:
                    rec.kept.append(t)
                    if zero:
                        rec.zeroed.append(t)
                return t
            return inner
        torch.empty, torch.zeros, torch.empty_like, torch.zeros_like, torch.full = (keep(f, f in (self._saved[1], self._saved[3])) for f in self._saved[:5])
        real_call = self._saved[5]

        def call(name, *args):
            rec.calls.append((name, args))
            return real_call(name, *args)
        _lib.call = call
        return self

    def __exit__(self, *a):
        torch.empty, torch.zeros, torch.empty_like, torch.zeros_like, torch.full, _lib.call = self._saved
        return False

    def replay(self, stream):
        h = stream.cuda_stream
        with torch.cuda.stream(stream):
            for t in self.zeroed:          # the zero-initialised workspaces (index tables, the global max-pool's atomic-max target)
                t.zero_()
        for name, args in self.calls:
            _lib.call(name, *(args[:-1] + (h,)))

    def owner(self, t):
        """Name of the first recorded call that got a pointer into tensor t."""
        lo = t.data_ptr()
        hi = lo + t.numel() * t.element_size()
        for k, (name, args) in enumerate(self.calls):
            for a in args[:-1]:
                if isinstance(a, int) and lo <= a < hi:
                    return "%s (call %d)" % (name, k)
        return "?"


def record(net, batch, h):
    eng = net._fused_engine()
    eng.use_side_stream = False
    with torch.no_grad():
        net.backbone(*batch, h)              # warm: weight images, lazily built split chains
        torch.cuda.synchronize()
        with Recorder() as rec:
            out = net.backbone(*batch, h)
        torch.cuda.synchronize()
    rec.out, rec.inputs = out, (batch, h)      # (the recorded pointers include the inputs': keep them alive)
    rec.base = [t.clone() for t in rec.kept]
    return rec


DETAIL = []      # (kept index, shape, dtype, differing elements, first flat indices) of the first few differing buffers


def diffs(rec):
    bad = {}
    for i, (t, b) in enumerate(zip(rec.kept, rec.base)):
        a = t.view(torch.int32) if t.dtype == torch.float32 else t
        r = b.view(torch.int32) if b.dtype == torch.float32 else b
        ne = (a != r)
        n = int(ne.sum())
        if n:
            key = rec.owner(t)
            bad[key] = bad.get(key, 0) + n
            if len(DETAIL) < 60:
                idx = ne.flatten().nonzero().flatten()
                DETAIL.append((i, tuple(t.shape), str(t.dtype), n, idx[:6].tolist(), idx[-1].item(), key,
                               t.flatten()[idx[:4]].tolist(), b.flatten()[idx[:4]].tolist()))
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    a = ap.parse_args()
    net = make_net()
    h8 = torch.randn(5, 8, 128, device=DEV, generator=torch.Generator(DEV).manual_seed(18)) * 0.1
    recs = [record(net, tie_batch(8, 256, 4300 + i), h8) for i in range(2)]
    print("recorded %d launches, %d pinned buffers per pass" % (len(recs[0].calls), len(recs[0].kept)), flush=True)
    s = [torch.cuda.Stream(), torch.cuda.Stream()]
    # (a) alone
    total = {}
    for it in range(a.iters):
        recs[0].replay(s[0])
        torch.cuda.synchronize()
        for k, v in diffs(recs[0]).items():
            total[k] = total.get(k, 0) + 1
    print("(a) one pass alone on one stream, %d replays: %s" % (a.iters, total or "bit-identical"), flush=True)
    # (b) two passes concurrently
    total = {}
    for it in range(a.iters):
        recs[0].replay(s[0])
        recs[1].replay(s[1])
        torch.cuda.synchronize()
        for r in recs:
            for k, v in diffs(r).items():
                total[k] = total.get(k, 0) + 1
    print("(b) two passes concurrently on two streams, %d replays: iterations with a difference, by producing launch:" % a.iters, flush=True)
    for k, v in sorted(total.items(), key=lambda kv: int(kv[0].split("call ")[1].rstrip(")")) if "call" in kv[0] else 1 << 30):
        print("    %-50s %d" % (k, v))
    if not total:
        print("    bit-identical")
    print("pinned buffers in allocation order:")
    for i, t in enumerate(recs[0].kept):
        print("    [%d] %s %s -> %s" % (i, tuple(t.shape), t.dtype, recs[0].owner(t)))
    print("first differing buffers (kept index, shape, dtype, #diff, first flat indices, last index, owner, got, expected):")
    for d in DETAIL:
        print("   ", d)


if __name__ == "__main__":
    main()
