#!/usr/bin/env python3
"""Focused reproducer for the non-reproducible re-levelling launch that tools/hazard_stages.py pointed at (round 4):
the geometry launches (level-1 selection + rtk_fps_relevel) on stream A, bit-compared with their own first result and with the
full selection kernel run level after level, while stream B replays kernels of a recorded backbone pass.

    python tools/hazard_fps.py --iters 3000                          quiet GPU / FPS noise
    python tools/hazard_fps.py --iters 1500 --study [--only rtk_pointwise_mlp,rtk_sa_scale_split]
                                                                      which kernels, as noise, make which launches irreproducible

History: the one-launch resume + settle kernel of rounds 2-3 (fps_relevel_kernel) failed this with rtk_pointwise_mlp or
rtk_sa_scale_split as noise (2-8 % of the iterations per tied cloud; profiles/r04_hazard_fps_before.txt), the level-1 kernel never
did -- until resume was added to it.  The ISA diff of those two builds named the cause (a packed fp32 instruction reading a
broadcast operand from the odd half of a register pair through op_sel: DESIGN section 8); with the broadcasts pinned in pairs of
their own every victim is bit-identical (profiles/r04_hazard_fps_after.txt)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from ratrack_amd import _lib  # noqa: E402
from ratrack_amd import fused as F  # noqa: E402
from hazard_harness import DEV, tie_batch  # noqa: E402


def run_levels(xyz, npoint, stream, relevel=True, front=False):
    """-> (idx (3,S,npoint), xyz (3,S,npoint,3), nuniq (3,S)) through the selection + re-levelling launches, or (relevel=False) through
    the full selection kernel level after level, or (front=True) through rtk_geometry_front -- the three levels of a cloud by one wave in
    one launch, what the product path issues since round 6."""
    S_, n, _ = xyz.shape
    h = stream.cuda_stream
    idx = torch.zeros(3, S_, npoint, dtype=torch.int32, device=DEV)
    out = torch.empty(3, S_, npoint, 3, dtype=torch.float32, device=DEV)
    cnt = torch.zeros(3, S_, dtype=torch.int32, device=DEV)
    tie = torch.zeros(3, S_, dtype=torch.int32, device=DEV)
    first = torch.zeros(S_, dtype=torch.int32, device=DEV)
    snap = torch.empty(S_, n, dtype=torch.float32, device=DEV)
    if front:
        _lib.call("rtk_geometry_front", S_, S_, n, npoint, xyz.data_ptr(), None, 0, None, None, None, None, idx.data_ptr(), out.data_ptr(), cnt.data_ptr(),
                  tie.data_ptr(), first.data_ptr(), snap.data_ptr(), None, None, None, None, None, 0, h)
    elif relevel:
        _lib.call("rtk_fps_centroids", S_, n, npoint, xyz.data_ptr(), idx[0].data_ptr(), out[0].data_ptr(), cnt[0].data_ptr(), tie[0].data_ptr(), None,
                  snap.data_ptr(), first.data_ptr(), h)
        _lib.call("rtk_fps_relevel", S_, npoint, 2, out[0].data_ptr(), cnt[0].data_ptr(), tie[0].data_ptr(), idx[1].data_ptr(), out[1].data_ptr(),
                  cnt[1].data_ptr(), tie[1].data_ptr(), idx[0].data_ptr(), snap.data_ptr(), n, first.data_ptr(), h)
    else:
        src, ns = xyz, n
        for l in range(3):
            _lib.call("rtk_fps_centroids", S_, ns, npoint, src.data_ptr(), idx[l].data_ptr(), out[l].data_ptr(), cnt[l].data_ptr(), None, None, None,
                      None, h)
            src, ns = out[l], npoint
    return idx, out, cnt, (tie[0], first, snap)


def relevel_only(ref, npoint, stream, resume=True):
    """Only the re-levelling launches, on the level-1 results of `ref`."""
    idx0, out0, cnt0 = ref[0][0], ref[1][0], ref[2][0]
    tie, first, snap = ref[3]
    S_ = idx0.shape[0]
    n = snap.shape[1]
    h = stream.cuda_stream
    idx = torch.zeros(2, S_, npoint, dtype=torch.int32, device=DEV)
    out = torch.empty(2, S_, npoint, 3, dtype=torch.float32, device=DEV)
    cnt = torch.zeros(2, S_, dtype=torch.int32, device=DEV)
    tie23 = torch.zeros(2, S_, dtype=torch.int32, device=DEV)
    extra = (idx0.data_ptr(), snap.data_ptr(), n, first.data_ptr()) if resume else (None, None, 0, None)
    _lib.call("rtk_fps_relevel", S_, npoint, 2, out0.data_ptr(), cnt0.data_ptr(), tie.data_ptr(), idx.data_ptr(), out.data_ptr(),
              cnt.data_ptr(), tie23.data_ptr(), *extra, h)
    return idx, out, cnt, None


DUMPS = [0]


def noise_study(a, xyz, ref):
    """Which kernels of a full backbone pass, running on a second stream, make the geometry launches irreproducible?"""
    from hazard_harness import make_net
    from hazard_stages import record
    net = make_net()
    h8 = torch.randn(5, 8, 128, device=DEV, generator=torch.Generator(DEV).manual_seed(18)) * 0.1
    rec = record(net, tie_batch(8, 256, 4301), h8)
    names = []
    for nm, _ in rec.calls:
        if nm not in names:
            names.append(nm)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    subsets = [("whole pass", None)] + [(nm, nm) for nm in names]
    if a.only:
        subsets = [(nm, nm) for nm in a.only.split(",")]
    for label, only in subsets:
        calls = [c for c in rec.calls if only is None or c[0] == only]
        for what in a.victims.split(","):
            bad = {}
            for it in range(a.iters):
                hb = sb.cuda_stream
                for _ in range(1 if only is None else max(1, 40 // len(calls))):
                    for nm, args in calls:
                        _lib.call(nm, *(args[:-1] + (hb,)))
                with torch.cuda.stream(sa):
                    if what in ("both launches", "full selection"):
                        cur = run_levels(xyz, 512, sa, what == "both launches")
                        lv = [(l, cur[0][l], cur[1][l], cur[2][l], ref[0][l], ref[1][l], ref[2][l]) for l in range(3)]
                    else:
                        cur = relevel_only(ref, 512, sa, resume=(what == "relevel only"))
                        lv = [(l + 1, cur[0][l], cur[1][l], cur[2][l], ref[0][l + 1], ref[1][l + 1], ref[2][l + 1]) for l in range(2)]
                sa.synchronize()
                for l, i_, x_, c_, ri, rx, rc in lv:
                    ne = (i_ != ri).any(dim=1) | (x_ != rx).flatten(1).any(dim=1) | (c_ != rc)
                    for s in ne.nonzero().flatten().tolist():
                        key = "level %d sample %d" % (l + 1, s)
                        bad[key] = bad.get(key, 0) + 1
                        if DUMPS[0] < 14:
                            DUMPS[0] += 1
                            d = (i_[s] != ri[s]).nonzero().flatten()
                            dx = (x_[s] != rx[s]).any(dim=1).nonzero().flatten()
                            p0 = int(d[0]) if d.numel() else -1
                            lo = max(p0 - 2, 0)
                            srcx = ref[1][l - 1][s] if l >= 1 else None          # the cloud this level selected from (reference values)
                            cons = None
                            if srcx is not None:
                                cons = int((x_[s] != srcx[i_[s].long()]).any(dim=1).sum())      # rows whose coordinates are not those of the stored index
                            vals, cnts = torch.unique(i_[s][:int(rc[s])], return_counts=True)
                            print("   DUMP %s (%s): T=%d | idx differs at %d positions [%d..%d], xyz at %d positions [%s..%s] | rows with xyz != cloud[idx]: %s | "
                                  "got idx %s expected %s | duplicates among the first nuniq picks: %s" %
                                  (key, what, int(ref[3][0][s]), d.numel(), p0, int(d[-1]) if d.numel() else -1, dx.numel(),
                                   int(dx[0]) if dx.numel() else -1, int(dx[-1]) if dx.numel() else -1, cons,
                                   i_[s][lo:p0 + 5].tolist(), ri[s][lo:p0 + 5].tolist(), vals[cnts > 1][:6].tolist()), flush=True)
                if it % 8 == 7:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            print("noise = %-26s victim = %-14s %d iterations: %s" % (label, what, a.iters, bad or "bit-identical"), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=3000)
    ap.add_argument("--study", action="store_true")
    ap.add_argument("--only", default="")
    ap.add_argument("--victims", default="both launches,relevel only,relevel no resume,full selection")
    a = ap.parse_args()
    tb = tie_batch(8, 256, 4300)
    xyz = torch.cat([tb[0], tb[1]], 0).permute(0, 2, 1).contiguous()
    big = torch.cat([xyz] * 8, 0).contiguous()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(sa):
        ref = run_levels(xyz, 512, sa, True)
        full = run_levels(xyz, 512, sa, False)
    torch.cuda.synchronize()
    for name, x, y in zip(("idx", "xyz", "nuniq"), ref[:3], full[:3]):
        print("relevel == full selection (quiet GPU): %s %s" % (name, bool(torch.equal(x, y))))
    print("tie rounds per sample:", ref[3][0].tolist(), " first tied round:", ref[3][1].tolist())
    if a.study:
        return noise_study(a, xyz, ref)
    for mode in ("quiet", "noise"):
        for relevel in (True, False):
            bad = {}
            for it in range(a.iters):
                if mode == "noise":
                    with torch.cuda.stream(sb):
                        run_levels(big, 512, sb, relevel=False)
                with torch.cuda.stream(sa):
                    cur = run_levels(xyz, 512, sa, relevel)
                if it % 16 == 15 or it == a.iters - 1:
                    torch.cuda.synchronize()
                sa.synchronize()
                for l in range(3):
                    ne = (cur[0][l] != ref[0][l]).any(dim=1) | (cur[1][l] != ref[1][l]).flatten(1).any(dim=1) | (cur[2][l] != ref[2][l])
                    for s in ne.nonzero().flatten().tolist():
                        key = "level %d sample %d" % (l + 1, s)
                        if key not in bad:
                            d = (cur[0][l][s] != ref[0][l][s]).nonzero().flatten()
                            bad[key] = [0, int(d[0]) if d.numel() else -1]
                        bad[key][0] += 1
            torch.cuda.synchronize()
            print("%-6s %-28s %d iterations: %s" % (mode, "fps_centroids + fps_relevel" if relevel else "full selection x3", a.iters,
                                                    {k: "%d times, first differing pick %d" % tuple(v) for k, v in bad.items()} or "bit-identical"), flush=True)


if __name__ == "__main__":
    main()
