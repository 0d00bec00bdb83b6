"""GPU: the HIP path against the CPU ORACLE at BASELINE.json's full sizes (round-3 review, "full-size parity leans on
self-comparison").  oracle/track4d_ref.py runs a B = 64, N = 256 eval forward in a few seconds and a train step with CPU autograd
in under a minute, so every single-GPU configuration meets the oracle at its own size:

  config 2   B = 32, N = 256    eval forward  (models/track4d.py:67-106)
  config 3   B = 64, N = 256    eval forward + one TRAIN step (multi-task loss losses/loss.py:8-31, backward, BatchNorm running statistics)
  config 5   B = 32, N = 1024   eval forward  (radar_5frames clouds: FPS really down-samples)

Indices (FPS, ball query, three-NN, kNN) are compared bit for bit for EVERY sample, floats within the north-star tolerance
max|a - b| <= 1e-4 max|b| per tensor.  The train step's gradients are compared per tensor against the oracle's fp32 gradients and
against the same step evaluated by the oracle in float64 (the arbiter; index ops see the fp32 coordinates, so the geometry is that
of the fp32 run), with the fp32 oracle's own distance from float64 as the per-tensor noise floor.
"""
import time

import numpy as np
import pytest
import torch

from oracle import track4d_ref as R
from ratrack_amd import fused as F
from ratrack_amd import synth, train_ops
from ratrack_amd.track4d import Args, Track4D

from _util import RTOL, reference_state_dict, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
NAMES = ["flow", "h", "cls", "cor", "pc1_features", "pc2_features", "prop"]


def _eval_net():
    net = Track4D(Args()).to(DEV).eval()
    net.load_state_dict(reference_state_dict(DEV), strict=True)
    return net


def _check_geometry(geo, tr, B, S=512):
    """Every index table of the fused path's geometry (both frames stacked: samples [0,B) = pc1, [B,2B) = pc2) against the
    oracle's trace, all samples.  Centroid-level rows at or beyond a sample's exhausted-cloud counter `nuniq` are duplicates of its
    row 0 in the reference (FPS re-picks point 0 once the cloud is exhausted) and are skipped by the fused path: for them the
    ORACLE's rows are checked to be copies of its row 0 (the property the skip relies on)."""
    torch.cuda.synchronize()
    halves = [tr, tr["pc2"]]
    nu = [geo.nuniq[l].cpu().numpy() for l in range(3)]
    checked = 0
    for half, t in enumerate(halves):
        sl = slice(half * B, (half + 1) * B)
        for l in range(3):
            ref = t["fps_idx"][l].numpy()
            got = geo.fps_idx[l][sl].cpu().numpy()
            assert np.array_equal(got, ref), "FPS level %d, frame %d" % (l + 1, half + 1)
            checked += ref.size
            for s in range(2):
                ref = t["ball_idx"][2 * l + s].numpy()
                got = geo.ball[l][s][sl].cpu().numpy()
                for b in range(B):
                    k = int(nu[l][half * B + b])
                    assert np.array_equal(got[b, :k], ref[b, :k]), "ball query level %d scale %d, sample %d" % (l + 1, s, half * B + b)
                    assert (ref[b, k:] == ref[b, :1]).all(), "oracle: rows of duplicate centroids are not copies of row 0"
                checked += ref.size
        for i, name in enumerate(("fp3", "fp2", "fp1")):
            d2r, ir = t["three_nn"][i]
            d2, idx, m = geo.nn[name]
            d2, idx = d2[sl].cpu().numpy(), idx[sl].cpu().numpy()
            lvl_unknown = {"fp3": 1, "fp2": 0, "fp1": None}[name]
            for b in range(B):
                k = d2r.shape[1] if lvl_unknown is None else int(nu[lvl_unknown][half * B + b])
                assert np.array_equal(idx[b, :k], ir[b, :k].numpy()), "three_nn %s indices, sample %d" % (name, half * B + b)
                assert np.array_equal(d2[b, :k], d2r[b, :k].numpy()), "three_nn %s squared distances, sample %d" % (name, half * B + b)
            checked += ir.numel()
    for i in range(2):
        assert torch.equal(geo.knn[i].cpu(), tr["knn_idx"][i]), "kNN table %d" % i
        checked += tr["knn_idx"][i].numel()
    return checked


@pytest.mark.parametrize("B,N,case", [(32, 256, 2010), (64, 256, 2000), (32, 1024, 2020)])
def test_eval_forward_matches_oracle_at_full_size(B, N, case):
    net = _eval_net()
    d = synth.make_frame_pairs(B, N, case)
    t = {k: torch.from_numpy(v) for k, v in d.items() if k != "gt_cls"}
    h0 = torch.randn(5, B, 128, generator=torch.Generator().manual_seed(case)) * 0.1
    g = [t[k].to(DEV) for k in ("pc1", "pc2", "feature1", "feature2")]
    with torch.no_grad():
        out = net.backbone(*g, h0.to(DEV))
        xyz = torch.cat([g[0], g[1]], 0).permute(0, 2, 1).contiguous()
        geo = F.Geometry(xyz, 512, knn_frames=B)
        torch.cuda.synchronize()
        sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
        tr = {}
        t0 = time.time()
        torch.set_num_threads(min(32, torch.get_num_threads()))
        ref = R.backbone(sd, t["pc1"], t["pc2"], t["feature1"], t["feature2"], h0, training=False, trace=tr)
        print("\noracle eval forward B=%d N=%d: %.1f s on the host" % (B, N, time.time() - t0))
    n_idx = _check_geometry(geo, tr, B)
    print("%d index / distance entries bit-identical" % n_idx)
    for name, a, b in zip(NAMES, out, ref):
        e = rel_err(a.float().cpu().numpy(), b.numpy())
        print("  %-14s rel-to-scale error %.2e" % (name, e))
        assert e <= RTOL, "%s: %.3e vs the CPU oracle at B=%d, N=%d" % (name, e, B, N)
    # the headline metric's second half: scene-flow EPE of the two implementations against the synthetic ground truth
    gt = t["gt_warp"]
    epe = [float(R.epe(t["pc1"] + f.float().cpu(), gt)) for f in (out[0], ref[0])]
    assert abs(epe[0] - epe[1]) <= 1e-4 * max(epe[1], 1.0), epe


def _oracle_train_step(d, dtype):
    sd = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in reference_state_dict().items()}
    names = [k for k in sd if not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"))]
    for k in names:
        sd[k].requires_grad_(True)
    t = {k: torch.from_numpy(v).to(dtype) for k, v in d.items() if k != "gt_cls"}
    flow, h, cls, *_ = R.backbone(sd, t["pc1"], t["pc2"], t["feature1"], t["feature2"], None, training=True)
    warp = t["pc1"] + flow
    gcls = torch.from_numpy(d["gt_cls"])
    B = warp.shape[0]
    total, items = 0.0, {}
    for b in range(B):      # the reference's loss is B = 1 code (losses/loss.py:89,131-142): called per sample and averaged
        tb, it = R.track_4d_loss(warp[b:b + 1], cls[b:b + 1], t["gt_warp"][b:b + 1], gcls[b])
        total = total + tb / B
        for k, v in it.items():
            items[k] = items.get(k, 0.0) + float(v) / B
    total.backward()
    grads = {k: (None if sd[k].grad is None else sd[k].grad.detach().double().numpy()) for k in names}
    stats = {k: v.detach().double().numpy() for k, v in sd.items() if k.endswith("running_mean") or k.endswith("running_var")}
    return items, flow.detach(), cls.detach(), grads, stats


def _train_step_parity(seed, B=64, N=256, copies=1):
    """One train step of the hand-written path at (B, N) on synthetic batch `seed` (copies > 1: B / copies distinct frame pairs, each
    `copies` times -- the same activations, BatchNorm statistics and decisions as the small batch, at the full batch's launch shapes) against the oracle's CPU autograd in fp32 and float64:
    asserts losses, train-mode flow / cls and BatchNorm running statistics within 1e-4; returns the per-tensor gradient rows
    (name, exact-zero?, error vs the fp32 oracle, error vs float64, the fp32 oracle's own error vs float64), each max|a - b| / max|b|."""
    assert B % copies == 0
    d = synth.make_frame_pairs(B // copies, N, seed)
    if copies > 1:
        d = {k: np.ascontiguousarray(np.concatenate([v] * copies, 0)) for k, v in d.items()}
    net = Track4D(Args()).to(DEV)
    net.load_state_dict(reference_state_dict(DEV), strict=True)
    net.train()
    g = {k: torch.from_numpy(v).to(DEV) for k, v in d.items()}
    flow, h, cls, *_ = net.backbone(g["pc1"], g["pc2"], g["feature1"], g["feature2"], None)
    total, items = train_ops.backbone_loss(g["pc1"], flow, cls, g["gt_warp"], g["gt_cls"], pretrain=False)
    total.backward()
    torch.cuda.synchronize()
    mine = {k: (None if p.grad is None else p.grad.detach().double().cpu().numpy()) for k, p in net.named_parameters()}
    sd = {k: v.detach().double().cpu().numpy() for k, v in net.state_dict().items()}
    torch.set_num_threads(min(32, torch.get_num_threads()))
    t0 = time.time()
    it32, flow32, cls32, g32, st32 = _oracle_train_step(d, torch.float32)
    it64, flow64, cls64, g64, _ = _oracle_train_step(d, torch.float64)
    print("\noracle train step B=%d N=%d seed %d in fp32 and float64: %.0f s on the host" % (B, N, seed, time.time() - t0))
    for k in ("Loss", "SceneFlowLoss", "SegLoss"):
        assert abs(float(items[k]) - it32[k]) <= 1e-4 * abs(it32[k]) + 1e-6, (k, float(items[k]), it32[k], it64[k])
    assert rel_err(flow.detach().cpu().numpy(), flow32.numpy()) <= RTOL and rel_err(cls.detach().cpu().numpy(), cls32.numpy()) <= RTOL
    for k, v in st32.items():
        assert rel_err(sd[k], v) <= 1e-4, k
    # ---- gradients ---------------------------------------------------------------------------------------------------------
    gmax = max(float(np.abs(v).max()) for v in g64.values() if v is not None)
    nmax = max(float(np.sqrt((v * v).sum())) for v in g64.values() if v is not None)
    rows = []
    for k, a64 in g64.items():
        m = mine.get(k)
        if a64 is None:
            assert m is None or float(np.abs(m).max()) == 0.0, "%s: dead parameter has a gradient" % k
            continue
        assert m is not None, "%s: no gradient" % k
        zero = float(np.sqrt((a64 * a64).sum())) <= 1e-8 * nmax      # exact gradient 0 (a bias in front of a BatchNorm): rounding noise only
        scale = 1e-4 * gmax if zero else float(np.abs(a64).max())
        e_arb = float(np.abs(m - a64).max()) / scale
        e_ref = float(np.abs(m - g32[k]).max()) / (1e-4 * gmax if zero else float(np.abs(g32[k]).max()))
        floor = float(np.abs(g32[k] - a64).max()) / scale
        rows.append((k, zero, e_ref, e_arb, floor))
    live = [r for r in rows if not r[1]]
    e_arb = np.array([r[3] for r in live])
    floor = np.array([r[4] for r in live])
    print("seed %d: %d gradient tensors, error vs float64: median %.1e, 90th percentile %.1e, max %.1e;  the fp32 oracle's own: median %.1e, "
          "90th %.1e, max %.1e" % (seed, len(live), np.median(e_arb), np.quantile(e_arb, 0.9), e_arb.max(), np.median(floor),
                                   np.quantile(floor, 0.9), floor.max()))
    for r in sorted(live, key=lambda r: -r[3])[:8]:
        print("   %-50s vs fp32 oracle %.2e | vs float64 %.2e | fp32 oracle vs float64 %.2e" % (r[0], r[2], r[3], r[4]))
    assert len(rows) > 100
    for k, zero, e_ref, e_a, fl in rows:
        if zero:
            assert e_a <= 1.0, (k, "exact-zero gradient carries more than rounding noise", e_a)
    return rows, e_arb, floor


def test_train_step_matches_oracle_at_full_size():
    """Config 3, B = 64, N = 256: one train step of the hand-written path (forward in train mode, multi-task loss, backward) against
    the oracle's CPU autograd in fp32 and in float64.  Losses, train-mode flow / cls, BatchNorm running statistics within 1e-4;
    gradients per tensor (ALL elements, not a sample)."""
    rows, e_arb, floor = _train_step_parity(2030)
    # Bounds.  At B = 64 the two fp32 evaluations (the oracle's and this path's) differ from float64 by DISCRETE events -- ReLU /
    # max-pool decisions within rounding of a tie -- and the oracle's own fp32 run is the yardstick for how much that is on this
    # batch (measured, round 4: this path median 2.2e-4 / 90th percentile 7.6e-4 / max 3.3e-3 from float64, the fp32 oracle
    # 9.0e-4 / 1.6e-3 / 1.1e-2).  Required: no further from float64 than the reference arithmetic itself, with absolute floors for
    # batches on which the oracle happens to have no flipped decision.
    for k, zero, e_ref, e_a, fl in rows:
        if not zero:
            assert e_a <= max(GRAD_MAX, 3.0 * fl), (k, e_ref, e_a, fl)
    assert np.median(e_arb) <= max(GRAD_MEDIAN, 1.5 * np.median(floor)), (np.median(e_arb), np.median(floor))
    assert np.quantile(e_arb, 0.9) <= max(GRAD_P90, 2.0 * np.quantile(floor, 0.9)), (np.quantile(e_arb, 0.9), np.quantile(floor, 0.9))
    assert e_arb.max() <= max(GRAD_MAX, 1.5 * floor.max()), (e_arb.max(), floor.max())


def test_train_step_on_a_full_size_batch_without_flipped_decisions():
    """The same step at the same launch shapes (B = 64, N = 256) on a batch where nothing is decided within rounding of a tie, so that
    the bound is the arithmetic's and not a statistic of flipped decisions: <= 1e-3 of the largest element for EVERY gradient tensor,
    median <= 2e-4, against float64.
    No batch of 64 distinct synthetic pairs is free of flips (tools/experiments/scan_fullsize_seed.py, round 6: on eight of eight the fp32
    ORACLE is 5e-3 ... 1.9e-2 from float64), and of 24 batches of 8 pairs most are not either -- where neither evaluation flips, this
    path sits at median 5e-6 ... 8e-6.  Batch 3012 (8 pairs: this path median 7.5e-6 / max 3.5e-4, the fp32 oracle 1.0e-5 / 6.1e-4) taken
    EIGHT TIMES is a B = 64 batch with the small batch's activations, BatchNorm statistics and decisions -- every kernel runs its
    full-size grid, every reduction over the batch sums 64 samples."""
    rows, e_arb, floor = _train_step_parity(3012, B=64, N=256, copies=8)
    assert e_arb.max() <= 1e-3 and np.median(e_arb) <= 2e-4, (e_arb.max(), np.median(e_arb), floor.max(), np.median(floor))


# Absolute floors of the per-tensor bounds at B = 64 (every tensor; median; 90th percentile), as max|a - b| / max|b| against float64.
# Round 5 (profiles/r05_fullsize_grad_parity.txt, tools/experiments/dbg_fullsize_grad.py): WHICH decisions flip depends on every rounding
# upstream of them -- this round's changed summation orders (neighbour sums by transposing reductions) drew a different set than round
# 4's (then: median 2.2e-4 / 90th 7.6e-4 / max 3.3e-3) on the same batch: median 7.2e-4 / 2.6e-3 / 7.6e-3 with the cost volume on the
# fp16 split path, 1.5e-3 / 3.0e-3 / 1.3e-2 with it on the fp32-input MFMA kernels (bit-exact fp32 fmaf chains) -- the split is the
# CLOSER of the two -- and 9.0e-4 / 1.6e-3 / 1.1e-2 for the fp32 oracle itself.  The bounds are therefore those of "one more fp32
# evaluation": within a small factor of the oracle's own distance per tensor and in distribution, with absolute floors.
GRAD_MAX, GRAD_MEDIAN, GRAD_P90 = 1e-2, 1e-3, 3e-3
