#!/usr/bin/env python3
"""Stream / lifetime / uninitialised-read hazard harness for the fused eval path (round-3 review, item 1).

The reference's forward is deterministic (models/track4d.py:67-106), and so is the fused path by construction
(no float atomics; the one atomic is an order-independent max).  Any run-to-run BIT difference is therefore a bug: a race
between the geometry stream and the main stream, a buffer recycled under a pending kernel, or a read of memory nobody wrote.
The harness makes each of those loud instead of waiting for a 1-in-90 tolerance failure:

  poison     every torch.empty / empty_like handed out while a case runs is filled first -- floats with NaN, integers with 1
             (an in-range, wrong index) -- so a read of uninitialised bytes changes the result deterministically;
  churn      between iterations random-sized NaN-filled blocks are allocated and dropped on the main stream (warm pools whose
             free blocks hold poison), `torch.cuda.empty_cache()` now and then;
  no-sync    iterations are enqueued back to back, comparisons are device-side flags read every `--check-every` iterations,
             so iteration k+1's allocations and kernels overlap iteration k's side-stream work;
  stages     the geometry tables (FPS indices, centroids, exhausted-cloud counters, ball / three-NN / kNN tables) are compared
             as well as the seven outputs, so a difference is attributed to the stage that produced it.

Cases: the three real frames as one padded batch (n_valid), every real pair at B = 1 with N1 != N2 (internal padding), a
synthetic batch with exact duplicate points and lattice ties, the captured GraphPipeline at depth 1 / 2 / 4 with rotating inputs.

    python tools/hazard_harness.py --iters 2000 --out profiles/r04_hazard_harness.txt
"""
import argparse
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from ratrack_amd import fused as F  # noqa: E402
from ratrack_amd import synth, vod_gt, vod_io  # noqa: E402
from ratrack_amd.track4d import Args, Track4D  # noqa: E402

DEV = "cuda"
NAMES = ["flow", "h", "cls", "cor", "pc1_features", "pc2_features", "prop"]


# ---- poison allocator ---------------------------------------------------------------------------------------------------
class poison:
    """While active, torch.empty / torch.empty_like / Tensor.new_empty return poisoned device tensors."""
    active = False
    _saved = None

    @staticmethod
    def _fill(t):
        if t.is_cuda and t.numel():
            if t.is_floating_point():
                t.fill_(float("nan"))
            elif t.dtype in (torch.int32, torch.int64, torch.int16, torch.uint8, torch.int8):
                t.fill_(1)
        return t

    def __enter__(self):
        e, el = torch.empty, torch.empty_like
        poison._saved = (e, el)
        torch.empty = lambda *a, **k: poison._fill(e(*a, **k))
        torch.empty_like = lambda *a, **k: poison._fill(el(*a, **k))
        poison.active = True
        return self

    def __exit__(self, *a):
        torch.empty, torch.empty_like = poison._saved
        poison.active = False
        return False


def churn(rng, blocks):
    """Allocate and drop random-sized NaN blocks on the current stream: the pools' free blocks now hold poison."""
    keep = []
    for _ in range(blocks):
        n = int(2 ** rng.uniform(8, 24))
        t = torch.empty(n // 4 + 1, dtype=torch.float32, device=DEV)
        t.fill_(float("nan"))
        keep.append(t)
    del keep


# ---- cases ----------------------------------------------------------------------------------------------------------------
def real_pairs():
    from _util import GOLDEN
    ex = os.path.join(GOLDEN, "vod_example")
    scans = [vod_io.load_radar_bin(os.path.join(ex, "radar_%s.bin" % f)) for f in ("00549", "01047", "01201")]
    return [vod_io.frame_pair_tensors(scans[i], scans[(i + 1) % 3], device=DEV) for i in range(3)]


def make_net():
    from _util import reference_state_dict
    net = Track4D(Args()).to(DEV).eval()
    net.load_state_dict(reference_state_dict(DEV), strict=True)
    return net


def tie_batch(b, n, case_id):
    """Synthetic pairs with exact duplicate points, symmetric equidistant points and a lattice sample (FPS ties between
    distinct points: the re-levelling kernel's resume path)."""
    d = synth.make_frame_pairs(b, n, case_id)
    for k in ("pc1", "pc2"):
        x = d[k]
        x[0, :, 40:56] = x[0, :, 8:9]
        x[1 % b, :, 100] = x[1 % b, :, 101]
        x[2 % b, :, 200] = x[2 % b, :, 0] + np.array([300., 0, 0], np.float32)
        x[2 % b, :, 201] = x[2 % b, :, 0] - np.array([300., 0, 0], np.float32)
        side = 7
        g = np.stack(np.meshgrid(*[np.arange(float(side))] * 3, indexing="ij"), -1).reshape(-1, 3)[:n].T
        x[3 % b] = g.astype(np.float32)
    return [torch.from_numpy(d[k]).to(DEV) for k in ("pc1", "pc2", "feature1", "feature2")]


def geometry_tensors(geo):
    """Every table of a geometry, restricted to what a consumer reads: since round 6 the eval path's index workspace is not zero-filled
    and the rows of duplicate centroids (>= the level's exhausted-cloud counter) are written by nothing (ball tables: every consumer
    skips them; three-NN tables of the centroid levels: likewise) -- they are masked to zero here, on the device, so that a difference
    that is left IS a defect."""
    out = {"tie": geo.tie}
    rows = torch.arange(geo.npoint, device=geo.tie.device)

    def live(t, nu):      # (S, npoint, k) table, (S) counters -> rows >= the counter zeroed
        return torch.where((rows[None, :] < nu[:, None].to(rows.dtype))[:, :, None], t, torch.zeros_like(t))
    for l in range(3):
        out["fps_idx%d" % l] = geo.fps_idx[l]
        out["xyz%d" % (l + 1)] = geo.xyz[l + 1]
        out["nuniq%d" % l] = geo.nuniq[l]
        for s in range(2):
            out["ball%d%d" % (l, s)] = live(geo.ball[l][s], geo.nuniq[l])
    for k, (d2, idx, m) in geo.nn.items():
        u = {"fp3": 2, "fp2": 1, "fp1": 0}[k]
        out["nn_idx_" + k] = live(idx, geo.nuniq[u - 1]) if u > 0 else idx
    if geo.knn is not None:
        out["knn0"], out["knn1"] = geo.knn
    return out


def masked_equal(a, b, mask_rows=None):
    return (a != b)


class Flags:
    """Device-side mismatch counters, one per named tensor; read (synchronise) only when asked."""

    def __init__(self):
        self.names, self.counts = [], None
        self.nonfinite = torch.zeros((), dtype=torch.int64, device=DEV)

    def add(self, items):
        if self.counts is None:
            self.names = [k for k, _, _ in items]
            self.counts = torch.zeros(len(items), dtype=torch.int64, device=DEV)
        for i, (k, a, ref) in enumerate(items):
            # bit comparison (NaN-safe): outputs are compared as integers
            ai = a.contiguous().view(torch.int32) if a.dtype == torch.float32 else a
            ri = ref.view(torch.int32) if ref.dtype == torch.float32 else ref
            self.counts[i] += (ai != ri).sum()

    def report(self):
        if self.counts is None:
            return {}
        c = self.counts.cpu().tolist()
        return {k: v for k, v in zip(self.names, c) if v}


def run_case(name, fn, iters, rng, check_every, log, use_poison_every=3):
    """fn() -> list of (name, tensor) ; first call (synchronised, no poison) is the reference."""
    torch.cuda.synchronize()
    ref = [(k, t.clone()) for k, t in fn()]
    torch.cuda.synchronize()
    flags = Flags()
    bad_iters = []
    t0 = time.time()
    for it in range(iters):
        if rng.random() < 0.02:
            torch.cuda.empty_cache()
        if rng.random() < 0.5:
            churn(rng, rng.randint(1, 12))
        if it % use_poison_every == 0:
            with poison():
                cur = fn()
        else:
            cur = fn()
        assert [k for k, _ in cur] == [k for k, _ in ref]
        flags.add([(k, a, r) for (k, a), (_, r) in zip(cur, ref)])
        del cur
        if (it + 1) % check_every == 0 or it == iters - 1:
            rep = flags.report()
            if rep:
                bad_iters.append((it + 1, dict(rep)))
                log("   !! %s: bit differences accumulated up to iteration %d: %s" % (name, it + 1, rep))
                flags = Flags()
    torch.cuda.synchronize()
    dt = time.time() - t0
    ok = not bad_iters
    log("%-44s %5d iterations  %6.1f s  %s" % (name, iters, dt, "bit-identical every time" if ok else "DIFFERENCES in %d windows" % len(bad_iters)))
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=2000)
    ap.add_argument("--check-every", type=int, default=50)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--cases", default="all")
    a = ap.parse_args()
    rng = random.Random(a.seed)
    lines = []

    def log(s):
        print(s, flush=True)
        lines.append(s)

    log("# tools/hazard_harness.py --iters %d --seed %d on %s (torch %s)" % (a.iters, a.seed, torch.cuda.get_device_name(0), torch.__version__))
    log("# every iteration compared BIT for BIT with the first (synchronised, unpoisoned) run; poison allocator every 3rd iteration,")
    log("# allocator churn with NaN blocks, empty_cache() at random, comparisons read every %d iterations (no sync in between)" % a.check_every)
    net = make_net()
    eng = net._fused_engine()
    pairs = real_pairs()
    pc1, pc2, f1, f2, nv = vod_gt.pad_frame_pairs(pairs, device=DEV)
    h3 = torch.randn(5, 3, 128, device=DEV, generator=torch.Generator(DEV).manual_seed(17)) * 0.1
    want = (lambda n: True) if a.cases == "all" else (lambda n: any(c in n for c in a.cases.split(",")))
    ok = True

    with torch.no_grad():
        # -- 1. the padded batch of the three real frames (the test that failed once): outputs + geometry tables -------------
        def padded():
            out = net.backbone(pc1, pc2, f1, f2, h3, n_valid=nv)
            return list(zip(NAMES, out))
        if want("padded_real_batch"):
            ok &= run_case("padded_real_batch (B=3, n_valid)", padded, a.iters, rng, a.check_every, log)

        def padded_geometry():
            xyz = torch.cat([pc1, pc2], 0).permute(0, 2, 1).contiguous()
            geo = F.Geometry(xyz, 512, side=eng.side, knn_frames=3, n_valid=nv.reshape(6).contiguous())
            geo.wait("knn")
            return sorted(geometry_tensors(geo).items())
        if want("padded_real_geometry"):
            if eng.side is None:
                eng.side = torch.cuda.Stream()
            ok &= run_case("padded_real_geometry (side stream)", padded_geometry, a.iters, rng, a.check_every, log)

        # -- 2. each real pair at B = 1, N1 != N2 (internal padding in Track4D.backbone) ----------------------------------------
        for b, p in enumerate(pairs):
            hb = h3[:, b:b + 1].contiguous()

            def single(p=p, hb=hb):
                return list(zip(NAMES, net.backbone(*p, hb)))
            if want("real_pair"):
                ok &= run_case("real_pair_%d (B=1, N1=%d, N2=%d)" % (b, p[0].shape[2], p[1].shape[2]), single, a.iters, rng, a.check_every, log)

        # -- 3. synthetic batch with duplicates / lattice ties --------------------------------------------------------------------
        tb = tie_batch(8, 256, 4242)
        h8 = torch.randn(5, 8, 128, device=DEV, generator=torch.Generator(DEV).manual_seed(18)) * 0.1

        def ties():
            return list(zip(NAMES, net.backbone(*tb, h8)))
        if want("tie_batch"):
            ok &= run_case("tie_batch (B=8, N=256, duplicates + lattice)", ties, a.iters, rng, a.check_every, log)

        def tie_geometry():
            xyz = torch.cat([tb[0], tb[1]], 0).permute(0, 2, 1).contiguous()
            geo = F.Geometry(xyz, 512, side=eng.side, knn_frames=8)
            geo.wait("knn")
            return sorted(geometry_tensors(geo).items())
        if want("tie_geometry"):
            ok &= run_case("tie_geometry (side stream)", tie_geometry, a.iters, rng, a.check_every, log)

        # -- 4. captured pipelines, rotating inputs ---------------------------------------------------------------------------------
        batches = [tie_batch(8, 256, 4300 + i) for i in range(3)]
        refs = []
        for t in batches:
            refs.append([x.clone() for x in net.backbone(*t, h8)])
        torch.cuda.synchronize()
        for depth in (1, 2, 4):
            if not want("pipeline"):
                continue
            pipe = F.GraphPipeline(eng, (*batches[0], h8), depth=depth)
            state = {"i": 0}
            flags_extra = []

            def piped(pipe=pipe, state=state):
                i = state["i"] % 3
                state["i"] += 1
                out = pipe.submit(*batches[i], h8)
                # the comparison below runs on the current stream: join only this slot's stream
                torch.cuda.current_stream().wait_stream(pipe.streams[(pipe.i - 1) % pipe.depth])
                return [("%s" % n, o) for n, o in zip(NAMES, out)], i

            # custom loop: the reference depends on which batch went in
            flags = Flags()
            bad = []
            t0 = time.time()
            for it in range(a.iters):
                if rng.random() < 0.5:
                    churn(rng, rng.randint(1, 12))
                cur, i = piped()
                flags.add([(k, o, r) for (k, o), r in zip(cur, refs[i])])
                if (it + 1) % a.check_every == 0 or it == a.iters - 1:
                    rep = flags.report()
                    if rep:
                        bad.append((it + 1, rep))
                        log("   !! pipeline depth %d: differences up to iteration %d: %s" % (depth, it + 1, rep))
                        flags = Flags()
            pipe.drain()
            torch.cuda.synchronize()
            log("%-44s %5d iterations  %6.1f s  %s" % ("graph_pipeline depth %d (3 rotating batches)" % depth, a.iters, time.time() - t0,
                                                       "bit-identical every time" if not bad else "DIFFERENCES in %d windows" % len(bad)))
            ok &= not bad
            del pipe
    log("RESULT: %s" % ("all cases bit-identical on every iteration" if ok else "DIFFERENCES FOUND"))
    if a.out:
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
