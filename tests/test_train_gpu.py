"""GPU: training-mode kernels (include/rtk_train.h) and the de-duplicated training path (ratrack_amd/train_path.py).

* bn_relu against nn.BatchNorm2d(train) + ReLU (+ max_pool2d) in float64 on the EXPANDED tensor (duplicate rows
  materialised as the reference has them): outputs, running statistics, dz (copies summed), dgamma, dbeta.
* one training step of Track4D through the de-duplicated path against the same step through the module path
  (reference structure: every duplicate row computed, MIOpen BatchNorm) -- and, in test_model_gpu.py, against the
  golden train step captured from the reference.
"""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from ratrack_amd import synth
from ratrack_amd.track4d import Args, Track4D
from ratrack_amd.train_ops import bn_relu

from _util import reference_state_dict, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _expand_rows(t, nuniq, npoint):
    """(S,C,U,ns) -> (S,C,npoint,ns): rows >= nuniq[b] are copies of row 0."""
    S_, C, U, ns = t.shape
    idx = torch.arange(npoint, device=t.device).view(1, npoint).repeat(S_, 1)
    idx = torch.where(idx >= nuniq.view(S_, 1), torch.zeros_like(idx), idx)
    return torch.gather(t, 2, idx.view(S_, 1, npoint, 1).expand(S_, C, npoint, ns)), idx


@pytest.mark.parametrize("ns,pool,groups,dedup", [(1, False, 1, True), (4, True, 2, True), (8, False, 2, True), (16, True, 1, True),
                                                  (32, True, 2, False), (1, False, 2, False), (32, False, 1, True)])
def test_bn_relu_matches_expanded_batchnorm(ns, pool, groups, dedup):
    torch.manual_seed(ns * 7 + pool + groups)
    S_, C, npoint = 4, 19, 40
    U = 24 if dedup else npoint
    nuniq = torch.tensor([24, 17, 24, 9] if dedup else [npoint] * 4, device=DEV)
    z = (torch.randn(S_, C, U, ns, device=DEV) * 1.7 + 0.3).requires_grad_(True)
    bn = nn.BatchNorm2d(C).to(DEV)
    with torch.no_grad():
        bn.weight.uniform_(-1.0, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
        bn.running_mean.normal_()
        bn.running_var.uniform_(0.5, 2.0)
    ref = nn.BatchNorm2d(C).to(DEV).double()
    ref.load_state_dict({k: v.double() if v.is_floating_point() else v.clone() for k, v in bn.state_dict().items()})
    ar = torch.arange(U, device=DEV).view(1, U)
    w = (ar < nuniq.view(S_, 1)).float()
    w[:, 0] += (npoint - nuniq).float()
    y = bn_relu(z, bn, w.contiguous() if dedup else None, (S_ // groups) * npoint * ns, groups, pool)

    zd = z.detach().double().requires_grad_(True)
    full, idx = _expand_rows(zd, nuniq, npoint)
    outs = []
    for g in range(groups):                                        # sequential calls, as the reference's per-frame passes
        sl = slice(g * S_ // groups, (g + 1) * S_ // groups)
        o = F.relu(ref(full[sl]))
        outs.append(F.max_pool2d(o, kernel_size=[1, ns]).squeeze(-1) if pool else o)
    yref_full = torch.cat(outs, 0)
    live = (ar < nuniq.view(S_, 1))                                # rows that consumers read
    yref = yref_full[:, :, :U]
    m = live.view(S_, 1, U, *([1] if not pool else [])).expand_as(y)
    assert torch.allclose(y[m].double(), yref[m], rtol=1e-5, atol=1e-5)
    for k in ("running_mean", "running_var"):
        assert torch.allclose(getattr(bn, k).double(), getattr(ref, k), rtol=1e-5, atol=1e-6), k
    assert int(bn.num_batches_tracked) == groups

    # gradient: random cotangent on the live de-duplicated rows; the expanded graph receives it on the first copy only
    # (consumers of a duplicate are redirected to row 0, so the copies have no consumers of their own)
    ct = torch.randn_like(y) * m.float()
    y.backward(ct)
    ct_full = torch.zeros_like(yref_full)
    ct_full[:, :, :U] = ct.double()
    yref_full.backward(ct_full)
    scale = float(zd.grad.abs().max())
    assert torch.allclose(z.grad.double()[:, :, :][live.view(S_, 1, U, 1).expand_as(z)],
                          zd.grad[live.view(S_, 1, U, 1).expand_as(z)], rtol=1e-4, atol=2e-5 * scale)
    assert torch.allclose(bn.weight.grad.double(), ref.weight.grad, rtol=1e-4, atol=1e-5 * float(ref.weight.grad.abs().max()))
    assert torch.allclose(bn.bias.grad.double(), ref.bias.grad, rtol=1e-4, atol=1e-5 * float(ref.bias.grad.abs().max()))


@pytest.mark.parametrize("det", [True, False])
@pytest.mark.parametrize("scale,shift,cot", [(1e-7, 0.0, 1.0), (1e-3, 5e-3, 1e-9), (1.0, 0.0, 1.0), (3e3, -1e4, 1e4), (1e6, 1e6, 1e-12)])
def test_batch_statistics_hold_at_any_scale(scale, shift, cot, det, monkeypatch):
    """det: the batch sums are accumulated in FIXED POINT (csrc/rtk_common.h rtk_stat_add: exact, order-independent float64 limbs; forward
    sums in units of 2^-36, backward sums in units of 2^-66): activations from 1e-7 to 1e6 -- with means far from zero -- and
    cotangents from 1e-12 to 1e4 against nn.BatchNorm2d in float64."""
    from ratrack_amd import train_ops
    monkeypatch.setattr(train_ops, "DETERMINISTIC", det)
    torch.manual_seed(3)
    S_, C, U, ns = 4, 24, 40, 8
    z = (torch.randn(S_, C, U, ns, device=DEV) * scale + shift).requires_grad_(True)
    bn = nn.BatchNorm2d(C).to(DEV)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    ref = nn.BatchNorm2d(C).to(DEV).double()
    ref.load_state_dict({k: v.double() if v.is_floating_point() else v.clone() for k, v in bn.state_dict().items()})
    y = bn_relu(z, bn, None, S_ * U * ns, 1, False)
    zd = z.detach().double().requires_grad_(True)
    yref = F.relu(ref(zd))
    assert torch.allclose(y.double(), yref, rtol=2e-5, atol=2e-5)
    for k in ("running_mean", "running_var"):
        assert torch.allclose(getattr(bn, k).double(), getattr(ref, k), rtol=1e-5, atol=1e-6 * max(1.0, scale * scale)), k
    ct = torch.randn_like(y) * cot
    y.backward(ct)
    yref.backward(ct.double())
    gs = float(zd.grad.abs().max())
    assert torch.allclose(z.grad.double(), zd.grad, rtol=1e-3, atol=2e-4 * gs), float((z.grad.double() - zd.grad).abs().max() / gs)
    assert torch.allclose(bn.weight.grad.double(), ref.weight.grad, rtol=1e-4, atol=1e-5 * float(ref.weight.grad.abs().max()))
    assert torch.allclose(bn.bias.grad.double(), ref.bias.grad, rtol=1e-4, atol=1e-5 * float(ref.bias.grad.abs().max()))


@pytest.mark.parametrize("bad", [float("nan"), float("inf"), 1e30])
def test_batch_statistics_propagate_non_finite_and_out_of_range_inputs(bad, monkeypatch):
    """Order-independent sums: an addend that is not finite, or too large for the fixed-point window (a partial sum of squares of 1e60), marks the sum: the
    channel's statistics read as NaN -- as the float sums this replaces did for non-finite inputs -- and nothing leaks into the other
    channels, whose outputs and running statistics are those of the clean batch."""
    from ratrack_amd import train_ops
    monkeypatch.setattr(train_ops, "DETERMINISTIC", True)
    torch.manual_seed(4)
    S_, C, U, ns = 2, 16, 32, 4
    z = torch.randn(S_, C, U, ns, device=DEV)
    z[1, 5, 7, 2] = bad
    bn = nn.BatchNorm2d(C).to(DEV)
    y = bn_relu(z, bn, None, S_ * U * ns, 1, False)
    assert torch.isnan(bn.running_mean[5])
    keep = [c for c in range(C) if c != 5]
    bn2 = nn.BatchNorm2d(C - 1).to(DEV)
    clean = bn_relu(z[:, keep].contiguous(), bn2, None, S_ * U * ns, 1, False)
    assert torch.equal(y[:, keep], clean)
    assert torch.equal(bn.running_mean[keep], bn2.running_mean) and torch.equal(bn.running_var[keep], bn2.running_var)


def test_train_step_gradients_are_reproducible_bit_for_bit(monkeypatch):
    """With train_ops.set_deterministic() / Trainer(deterministic=True): the same train step (forward, loss, backward) from the same weights ten times: every gradient tensor, the loss and the outputs
    bit-identical.  Batch sums are exact fixed-point accumulations (rtk_stat_add), the first layer's gather sums per-plane fixed point
    in LDS, the offset-weight gradients and the loss per-sample shares added in a fixed order, weight gradients per-workgroup partials
    added in a fixed order: no sum in the step depends on the order in which workgroups or waves arrive.  (tools/hazard_train.py is
    the same check at B = 64 over any number of repetitions.)"""
    from ratrack_amd import train_ops
    monkeypatch.setattr(train_ops, "DETERMINISTIC", True)
    net = Track4D(Args()).to(DEV)
    net.load_state_dict(reference_state_dict(DEV), strict=True)
    net.train()
    d = synth.make_frame_pairs(16, 256, 2031)
    g = {k: torch.from_numpy(v).to(DEV) for k, v in d.items()}

    def step():
        net.zero_grad(set_to_none=True)
        flow, h, cls, *_ = net.backbone(g["pc1"], g["pc2"], g["feature1"], g["feature2"], None)
        total, items = train_ops.backbone_loss(g["pc1"], flow, cls, g["gt_warp"], g["gt_cls"], pretrain=False)
        total.backward()
        out = {"loss": total.detach().clone().reshape(1), "flow": flow.detach().clone(), "cls": cls.detach().clone(), "h": h.detach().clone()}
        out.update({"grad/" + k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None})
        return out
    step()
    ref = step()
    assert len(ref) > 150
    for _ in range(10):
        cur = step()
        bad = [k for k, v in cur.items() if not torch.equal(v.view(torch.int32), ref[k].view(torch.int32))]
        assert not bad, bad[:8]


def _train_once(dedup, B, N, pretrain=False):
    from ratrack_amd import loss as L
    net = Track4D(Args()).to(DEV)
    net.load_state_dict(reference_state_dict(DEV), strict=True)
    net.train()
    net._dedup_train = dedup
    d = synth.make_frame_pairs(B, N, 5)
    t = {k: torch.from_numpy(v).to(DEV) for k, v in d.items()}
    flow, h, cls, cor, f1, f2, prop = net.backbone(t["pc1"], t["pc2"], t["feature1"], t["feature2"], None)
    total, items = L.backbone_loss(t["pc1"] + flow, cls, t["gt_warp"], t["gt_cls"], pretrain=pretrain)
    total.backward()
    grads = {k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in net.named_parameters()}
    stats = {k: v.detach().clone() for k, v in net.state_dict().items() if "running_" in k or "num_batches" in k}
    return dict(flow=flow.detach(), cls=cls.detach(), prop=prop.detach(), h=h.detach(), f1=f1.detach(), f2=f2.detach(),
                total=float(total.detach())), grads, stats


@pytest.mark.parametrize("B,N", [(2, 256), (3, 200), (2, 640), (64, 256)])
def test_dedup_train_step_matches_module_path(B, N):
    """Same weights, same batch: forward outputs, every parameter gradient and every BatchNorm running statistic of the
    de-duplicated path agree with the module path that computes all 512 centroid rows.  (64, 256) is BASELINE config 3 at
    its full size."""
    out_d, g_d, s_d = _train_once(True, B, N)
    out_m, g_m, s_m = _train_once(False, B, N)
    for k in ("flow", "cls", "prop", "h", "f1", "f2"):
        a, b = out_d[k], out_m[k]
        assert float((a - b).abs().max()) <= 2e-4 * float(b.abs().max()) + 1e-6, k
    assert abs(out_d["total"] - out_m["total"]) <= 1e-4 * abs(out_m["total"])
    gmax = max(float(g.norm()) for g in g_m.values() if g is not None)
    rels = []
    for k, gm in g_m.items():
        gd = g_d[k]
        if gm is None:
            assert gd is None or float(gd.norm()) == 0.0, k
            continue
        err = float((gd - gm).norm())
        if float(gm.norm()) > 1e-4 * gmax:
            rels.append(err / float(gm.norm()))
        # Two fp32 evaluations of a ReLU / max-pool network differ by discrete events -- an activation within rounding of 0 flips its
        # mask and that element's upstream gradient enters or leaves a sum (profiles/r03_grad_parity.txt: the float64
        # arbiter puts the fp32 REFERENCE up to 1.5e-3 from the truth for the same reason).  Observed worst tensors here: 3e-3 (B=2),
        # 1.3e-2 (B=3, N=200), 6e-4 (N=640), 1.7e-2 (B=64) in relative L2 -- flips, not a bias: the MEDIAN over the 153 parameter tensors is 1e-5 where
        # nothing flips (N=640), 3e-4 ... 6e-3 where a decision near the output flips and everything upstream feels it (B=3: 2.1e-3 in most
        # runs, 6.0e-3 in one -- the order of the LDS float atomics differs from run to run); a wrong term, count or weight moves it to
        # O(1).  Asserted below at 1e-2, under the per-tensor bound.
        assert err <= (1.5e-2 if B < 32 else 4e-2) * float(gm.norm()) + (1e-5 if B < 32 else 3e-5) * gmax, (k, err, float(gm.norm()))
    rels.sort()
    print("\nB=%d N=%d: relative L2 gradient difference over %d parameter tensors: median %.2e, worst %.2e" % (B, N, len(rels), rels[len(rels) // 2], rels[-1]))
    assert rels[len(rels) // 2] <= 1e-2, rels[len(rels) // 2]
    for k, v in s_m.items():
        if v.is_floating_point():
            assert float((s_d[k] - v).abs().max()) <= 1e-4 * float(v.abs().max()) + 1e-7, k
        else:
            assert int(s_d[k]) == int(v), k


def test_dedup_train_handles_duplicate_input_points():
    """Clouds with repeated points (padded frames): FPS exhausts before n picks, so some of the min(n, npoint) rows are
    dead (weight 0).  Forward and gradients still match the module path."""
    from ratrack_amd import loss as L
    B, N = 2, 256
    d = synth.make_frame_pairs(B, N, 9)
    for k in ("pc1", "pc2"):
        d[k][:, :, 200:] = d[k][:, :, :56]                         # 56 exact duplicates per cloud
    d["feature1"][:, :, 200:] = d["feature1"][:, :, :56]
    d["feature2"][:, :, 200:] = d["feature2"][:, :, :56]
    t = {k: torch.from_numpy(v).to(DEV) for k, v in d.items()}
    res = []
    for dedup in (True, False):
        net = Track4D(Args()).to(DEV)
        net.load_state_dict(reference_state_dict(DEV), strict=True)
        net.train()
        net._dedup_train = dedup
        flow, h, cls, *_ = net.backbone(t["pc1"], t["pc2"], t["feature1"], t["feature2"], None)
        total, _ = L.backbone_loss(t["pc1"] + flow, cls, t["gt_warp"], t["gt_cls"], pretrain=False)
        total.backward()
        res.append((flow.detach(), {k: p.grad for k, p in net.named_parameters() if p.grad is not None}))
    (fa, ga), (fb, gb) = res
    assert torch.isfinite(fa).all()
    assert float((fa - fb).abs().max()) <= 2e-4 * float(fb.abs().max())
    gmax = max(float(g.norm()) for g in gb.values())
    for k in gb:
        assert torch.isfinite(ga[k]).all(), k
        assert float((ga[k] - gb[k]).norm()) <= 1.5e-2 * float(gb[k].norm()) + 1e-5 * gmax, k


@pytest.mark.parametrize("B,N", [(2, 256), (3, 77), (1, 1024)])
def test_cost_volume_operator_matches_module_autograd(B, N):
    """Fused cost-volume forward + backward kernels against FeatureCorrelator.forward under framework autograd: output,
    gradients of both feature tensors and of every fc_layer parameter on the path."""
    from ratrack_amd import train_path as TP
    torch.manual_seed(3)
    net = Track4D(Args()).to(DEV)
    net.load_state_dict(reference_state_dict(DEV), strict=True)
    fc = net.fc_layer
    d = synth.make_frame_pairs(B, N, 11)
    pc1, pc2 = torch.from_numpy(d["pc1"]).to(DEV), torch.from_numpy(d["pc2"]).to(DEV)
    res = []
    for fused_op in (True, False):
        torch.manual_seed(5)
        f1 = (torch.randn(B, 256, N, device=DEV) * 0.5).requires_grad_(True)
        f2 = (torch.randn(B, 256, N, device=DEV) * 0.5).requires_grad_(True)
        fc.zero_grad(set_to_none=True)
        out = TP.correlator_train(fc, pc1, pc2, f1, f2) if fused_op else fc(pc1, pc2, f1, f2)
        ct = torch.randn(out.shape, device=DEV, generator=torch.Generator(DEV).manual_seed(7))
        out.backward(ct)
        grads = {k: p.grad.detach().clone() for k, p in fc.named_parameters() if p.grad is not None}
        grads["f1"], grads["f2"] = f1.grad.clone(), f2.grad.clone()
        res.append((out.detach(), grads))
    (oa, ga), (ob, gb) = res
    assert float((oa - ob).abs().max()) <= 1e-4 * float(ob.abs().max())
    assert set(ga) == set(gb)
    for k in gb:
        err, ref = float((ga[k] - gb[k]).norm()), float(gb[k].norm())
        assert err <= 2e-4 * ref + 1e-7, (k, err, ref)


@pytest.mark.parametrize("m,n,C", [(1000, 77, 64), (4096, 256, 256), (1111, 300, 256), (16384, 1024, 256)])
def test_scatter_add_rows_matches_index_add(m, n, C):
    from ratrack_amd import _lib, train_ops  # noqa: F401
    B = 3
    g = torch.Generator(DEV).manual_seed(1)
    idx = torch.randint(0, n, (B, m), device=DEV, generator=g)
    src = torch.randn(B, m, C, device=DEV, generator=g)
    dst = torch.full((B, n, C), float("nan"), device=DEV)
    _lib.call("rtk_scatter_add_rows", B, m, n, C, idx.data_ptr(), src.data_ptr(), dst.data_ptr(), torch.cuda.current_stream().cuda_stream)
    ref = torch.zeros(B, n, C, device=DEV, dtype=torch.float64)
    ref.scatter_add_(1, idx.unsqueeze(-1).expand(-1, -1, C), src.double())
    assert torch.allclose(dst.double(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("graph", [False, True])
def test_deterministic_trainer_repeats_its_trajectory_bit_for_bit(graph):
    """Trainer(deterministic=True): six optimisation steps (lr = 1e-3, Adam) from the same weights on the same batches, twice: the
    losses of every step, every parameter, every optimizer moment and every BatchNorm running statistic end up bit-identical --
    eagerly and as a captured hipGraph (warm-up steps, capture, replays).  (Without the mode two runs differ by 1e-3 after a few
    steps: test_graphed_trainer_matches_eager.)"""
    from ratrack_amd.train import Trainer
    B, N = 4, 256
    batches = []
    for i in range(6):
        d = synth.make_frame_pairs(B, N, 40 + i)
        batches.append({k: torch.from_numpy(v).to(DEV) for k, v in d.items()})

    def run():
        net = Track4D(Args()).to(DEV)
        net.load_state_dict(reference_state_dict(DEV), strict=True)
        tr = Trainer(net, graph=graph, lr=1e-3, deterministic=True)
        h = torch.zeros(5, B, 128, device=DEV)
        losses = []
        for t in batches:
            items, _ = tr.step(t["pc1"], t["pc2"], t["feature1"], t["feature2"], t["gt_warp"], t["gt_cls"], h)
            losses.append(items["Loss"].detach().clone())
        torch.cuda.synchronize()
        state = {k: v.detach().clone() for k, v in net.state_dict().items()}
        for gi, grp in enumerate(tr.opt.state_dict()["state"].items()):
            for k, v in grp[1].items():
                if torch.is_tensor(v):
                    state["opt/%d/%s" % (grp[0], k)] = v.detach().clone()
        return torch.stack(losses), state

    la, sa = run()
    lb, sb = run()
    assert torch.equal(la, lb)
    assert float((la[0] - la[-1]).abs()) > 0                      # the parameters did move
    assert sa.keys() == sb.keys() and len(sa) > 400
    bad = [k for k in sa if not torch.equal(sa[k], sb[k])]
    assert not bad, bad[:8]


def test_graphed_trainer_matches_eager():
    """Trainer(graph=True): 3 eager warm-up steps, capture, replays.  With lr = 0 (parameters frozen, everything else live)
    the captured step must reproduce the eager losses and BatchNorm running statistics on every batch -- the static input
    buffers, the geometry and every kernel are replayed correctly.  With lr = 1e-3 the replayed optimizer moves the
    parameters as far as the eager one (the trajectories themselves are chaotic: Adam's first steps are sign-like, and
    float atomics make even two eager runs differ by 1e-3 after a few steps)."""
    from ratrack_amd.train import Trainer
    B, N = 2, 256
    batches = []
    for i in range(6):
        d = synth.make_frame_pairs(B, N, 20 + i)
        batches.append({k: torch.from_numpy(v).to(DEV) for k, v in d.items()})

    def run(graph, lr):
        net = Track4D(Args()).to(DEV)
        net.load_state_dict(reference_state_dict(DEV), strict=True)
        init = {k: v.detach().clone() for k, v in net.named_parameters()}
        tr = Trainer(net, graph=graph, lr=lr)
        h = torch.zeros(5, B, 128, device=DEV)
        losses = []
        for t in batches:
            items, _ = tr.step(t["pc1"], t["pc2"], t["feature1"], t["feature2"], t["gt_warp"], t["gt_cls"], h)
            losses.append(float(items["Loss"]))
        moved = float(torch.sqrt(sum(((p.detach() - init[k]) ** 2).sum() for k, p in net.named_parameters())))
        return losses, {k: v.detach().clone() for k, v in net.state_dict().items()}, moved

    la, sa, _ = run(False, 0.0)
    lb, sb, _ = run(True, 0.0)
    np.testing.assert_allclose(lb, la, rtol=1e-5)
    for k, v in sa.items():
        if "running_" in k:
            assert float((sb[k] - v).abs().max()) <= 1e-5 * float(v.abs().max()) + 1e-7, k
        elif "num_batches" in k:
            assert torch.equal(sb[k], v), k
    lc, _, moved_e = run(False, 1e-3)
    ld, _, moved_g = run(True, 1e-3)
    assert np.isfinite(ld).all() and moved_g > 0
    assert abs(moved_g - moved_e) <= 0.05 * moved_e, (moved_g, moved_e)
    np.testing.assert_allclose(ld[:3], lc[:3], rtol=1e-3)          # the eager warm-up steps


def test_data_parallel_step_structure_on_one_gpu():
    """The world > 1 step structure, exercised on ONE GPU through a 1-rank RCCL process group: gradients packed into the flat
    bucket, a real RCCL all-reduce, Adam on gradients aliasing the bucket --
      (a) eager, (b) graph A / eager all-reduce / graph B (the default for world > 1), (c) one graph with the collective
    captured (opt-in).  With lr = 0 all three must reproduce the plain single-process losses batch by batch; with
    lr = 1e-3 the optimizer must move the parameters as far as the plain trainer does."""
    import os
    import socket
    import torch.distributed as dist
    from ratrack_amd.train import Trainer
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", torch.cuda.current_device()))
    try:
        B, N = 2, 256
        batches = []
        for i in range(7):
            d = synth.make_frame_pairs(B, N, 40 + i)
            batches.append({k: torch.from_numpy(v).to(DEV) for k, v in d.items()})

        def run(lr, **kw):
            net = Track4D(Args()).to(DEV)
            net.load_state_dict(reference_state_dict(DEV), strict=True)
            init = {k: v.detach().clone() for k, v in net.named_parameters()}
            tr = Trainer(net, lr=lr, **kw)
            if kw:
                tr.reducer.always_pack = tr.reducer.always_reduce = True
            h = torch.zeros(5, B, 128, device=DEV)
            losses = []
            for t in batches:
                items, _ = tr.step(t["pc1"], t["pc2"], t["feature1"], t["feature2"], t["gt_warp"], t["gt_cls"], h)
                losses.append(float(items["Loss"]))
            moved = float(torch.sqrt(sum(((p.detach() - init[k]) ** 2).sum() for k, p in net.named_parameters())))
            if kw:
                assert tr.reducer.payload_bytes == 4 * 1058196      # the live set of SURVEY fact 8 (of a 1 816 031-slot bucket)
                if kw.get("graph"):
                    assert (tr._g_opt is not None) == bool(kw.get("split_graph"))
            return losses, moved

        ref, _ = run(0.0)
        for kw in (dict(graph=False, split_graph=False), dict(graph=True, split_graph=True), dict(graph=True, split_graph=False)):
            got, _ = run(0.0, **kw)
            np.testing.assert_allclose(got, ref, rtol=1e-5, err_msg=str(kw))
        _, moved_ref = run(1e-3)
        for kw in (dict(graph=True, split_graph=True), dict(graph=True, split_graph=False)):
            got, moved = run(1e-3, **kw)
            assert np.isfinite(got).all() and abs(moved - moved_ref) <= 0.05 * moved_ref, (kw, moved, moved_ref)
    finally:
        dist.destroy_process_group()


def test_captured_collective_survives_twenty_replays():
    """bench.py --graph-collective / Trainer(graph_collective=True): the RCCL all-reduce CAPTURED inside the step's one hipGraph, on a
    1-rank RCCL group (the only world a 1-GPU lease offers), for 20 replays after the capture: every replay's losses equal the eager
    data-parallel step's on the same batch and weights (lr = 0), and the bucket's five guard words -- the digest of the live-parameter
    set that rides in the same collective (ratrack_amd/ddp.py) -- still hold world x the rank's stamp after every replay: the captured
    pack -> all-reduce -> unpack sequence keeps writing and summing them."""
    import os
    import socket
    import torch.distributed as dist
    from ratrack_amd.train import Trainer
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", torch.cuda.current_device()))
    try:
        B, N = 2, 256
        batches = []
        for i in range(5):
            d = synth.make_frame_pairs(B, N, 60 + i)
            batches.append({k: torch.from_numpy(v).to(DEV) for k, v in d.items()})
        h = torch.zeros(5, B, 128, device=DEV)

        def make(**kw):
            net = Track4D(Args()).to(DEV)
            net.load_state_dict(reference_state_dict(DEV), strict=True)
            tr = Trainer(net, lr=0.0, **kw)
            tr.reducer.always_pack = tr.reducer.always_reduce = True
            return tr

        def one(tr, t):
            items, _ = tr.step(t["pc1"], t["pc2"], t["feature1"], t["feature2"], t["gt_warp"], t["gt_cls"], h)
            return float(items["Loss"])

        eager = make(graph=False, split_graph=False)
        want = [one(eager, batches[i % 5]) for i in range(24)]
        tr = make(graph=True, graph_collective=True)
        assert not tr.split, "graph_collective = one graph, no eager collective between two graphs"
        got = []
        for i in range(24):                                              # 3 eager warm-ups, the capture (+ its replay), 20 more replays
            got.append(one(tr, batches[i % 5]))
            if tr._g is not None:
                assert tr._g_opt is None
                R = tr.reducer
                guard = R._buf[R.flat.numel():].cpu()
                assert torch.equal(guard, R._guard_host * dist.get_world_size()), (i, guard.tolist(), R._guard_host.tolist())
                assert float(guard[4]) == float(len(R._live)) and len(R._live) > 100
        assert tr._g is not None and tr._count >= 3
        np.testing.assert_allclose(got, want, rtol=1e-5)
        assert tr.reducer.payload_bytes == 4 * 1058196
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("chans,ns,U,n_src,groups", [([16, 16, 32], 4, 200, 200, 2), ([32, 32], 8, 57, 100, 1), ([32, 64], 16, 242, 242, 2),
                                                      ([64, 64], 32, 100, 256, 1)])
def test_fused_sa_chain_matches_unfused_operators(chans, ns, U, n_src, groups):
    """The fused set-abstraction chain (first-layer kernel, conv+BN kernels, MFMA weight gradient, two-pass backward) against
    the same chain built from bn_relu + framework convolutions, on odd sizes (positions not a multiple of 64, dead rows,
    two statistics groups): output, gradients of the projection input and of every parameter, running statistics."""
    import copy
    import types
    from ratrack_amd import train_path as TP
    from ratrack_amd.pytorch_utils import SharedMLP
    torch.manual_seed(sum(chans) + ns)
    S_, npoint, Cf = 4, 512, 24
    mlp0 = SharedMLP([3 + Cf] + chans, bn=True).to(DEV)
    with torch.no_grad():
        for p in mlp0.parameters():
            p.mul_(1.5).add_(torch.randn_like(p) * 0.05)
    nuniq = torch.tensor([U, max(U - 13, 1), U, max(U // 2, 1)], device=DEV)
    ar = torch.arange(U, device=DEV).view(1, U)
    row_w = (ar < nuniq.view(S_, 1)).float()
    row_w[:, 0] += (npoint - nuniq).float()
    g = torch.Generator(DEV).manual_seed(3)
    tg = types.SimpleNamespace(samples=S_, npoint=npoint, row_w=[row_w.contiguous()],
                               ball=[[torch.randint(0, n_src, (S_, U, ns), device=DEV, generator=g, dtype=torch.int32)]],
                               dxyz=[[torch.randn(S_, 3, U, ns, device=DEV, generator=g)]])
    if ns % 4 == 0:       # the gather-form first-layer backward (train_path.TrainGeometry builds this table per level and scale)
        from ratrack_amd import _lib
        off = torch.empty(S_, n_src + 1, dtype=torch.int32, device=DEV)
        inv = torch.empty(S_, U * ns, dtype=torch.int16, device=DEV)
        _lib.call("rtk_group_inverse_index", S_, n_src, U * ns, tg.ball[0][0].data_ptr(), off.data_ptr(), inv.data_ptr(),
                  torch.cuda.current_stream().cuda_stream)
        tg.inv = [[(off, inv)]]
    res = []
    for fused_chain in (True, False):
        mlp = copy.deepcopy(mlp0)
        feats = torch.randn(S_, Cf, n_src, device=DEV, generator=torch.Generator(DEV).manual_seed(5)).requires_grad_(True)
        TP.FUSED_SA_CHAIN = fused_chain
        try:
            out = TP._sa_scale(mlp, tg, 0, 0, feats, groups)
        finally:
            TP.FUSED_SA_CHAIN = True
        live = (ar < nuniq.view(S_, 1)).view(S_, 1, U).expand_as(out)
        ct = torch.randn(out.shape, device=DEV, generator=torch.Generator(DEV).manual_seed(9)) * live
        out.backward(ct)
        res.append((out.detach()[live], feats.grad.clone(), {k: p.grad.clone() for k, p in mlp.named_parameters()},
                    {k: v.clone() for k, v in mlp.state_dict().items() if "running" in k}))
    (oa, fa, ga, sa), (ob, fb, gb, sb) = res
    assert float((oa - ob).abs().max()) <= 1e-4 * float(ob.abs().max())
    assert float((fa - fb).norm()) <= 2e-4 * float(fb.norm())
    for k in gb:
        assert float((ga[k] - gb[k]).norm()) <= 2e-4 * float(gb[k].norm()) + 1e-6, k
    for k in sb:
        assert float((sa[k] - sb[k]).abs().max()) <= 1e-5 * float(sb[k].abs().max()) + 1e-7, k


@pytest.mark.parametrize("B", [1, 5, 64])
def test_gru_step_operator_matches_nn_gru(B):
    """rtk_gru_step + rtk_gru_step_bwd against nn.GRU(128,128,5) on a length-1 sequence: outputs, input / state gradients and
    all 20 parameter gradients, with cotangents on both outputs."""
    from ratrack_amd.train_ops import gru_step
    torch.manual_seed(B)
    gru = nn.GRU(128, 128, 5).to(DEV)
    res = []
    for custom in (True, False):
        gru.zero_grad(set_to_none=True)
        g = torch.Generator(DEV).manual_seed(2)
        x = torch.randn(B, 128, device=DEV, generator=g).requires_grad_(True)
        h0 = (torch.randn(5, B, 128, device=DEV, generator=g) * 0.5).requires_grad_(True)
        if custom:
            y, h1 = gru_step(x, h0, gru)
        else:
            o, h1 = gru(x.unsqueeze(0), h0)
            y = o[0]
        cy, ch = torch.randn(B, 128, device=DEV, generator=g), torch.randn(5, B, 128, device=DEV, generator=g) * 0.3
        (y * cy).sum().add((h1 * ch).sum()).backward()
        res.append((y.detach(), h1.detach(), x.grad.clone(), h0.grad.clone(), {k: p.grad.clone() for k, p in gru.named_parameters()}))
    a, b = res
    for i in range(4):
        assert float((a[i] - b[i]).abs().max()) <= 2e-5 * float(b[i].abs().max()) + 1e-6, i
    for k in b[4]:
        assert float((a[4][k] - b[4][k]).norm()) <= 1e-4 * float(b[4][k].norm()) + 1e-6, k


def test_training_reduces_the_loss_on_a_fixed_batch():
    """End-to-end sanity of the training path + optimizer: 25 captured steps on one fixed batch lower the multi-task loss
    substantially and keep every parameter finite."""
    from ratrack_amd.train import Trainer
    torch.manual_seed(0)
    net = Track4D(Args()).to(DEV)
    net.load_state_dict(reference_state_dict(DEV), strict=True)
    d = synth.make_frame_pairs(4, 256, 77)
    t = {k: torch.from_numpy(v).to(DEV) for k, v in d.items()}
    tr = Trainer(net, graph=True)
    h = torch.zeros(5, 4, 128, device=DEV)
    losses = []
    for _ in range(25):
        items, _ = tr.step(t["pc1"], t["pc2"], t["feature1"], t["feature2"], t["gt_warp"], t["gt_cls"], h)
        losses.append(float(items["Loss"]))
    assert all(np.isfinite(losses)), losses
    assert losses[-1] < 0.7 * losses[0], losses
    assert all(torch.isfinite(p).all() for p in net.parameters())


def _pm(t):
    """Same values as the (S,C,P) tensor t, stored point-major (a permuted view of an (S,P,C) tensor)."""
    return t.permute(0, 2, 1).contiguous().permute(0, 2, 1)


@pytest.mark.parametrize("S,P,cins,cout,layouts,bias,out_pm", [
    (4, 256, [128], 128, "c", False, False), (3, 242, [64, 64], 128, "cc", False, False), (2, 77, [2, 256, 256], 16, "ccp", False, False),
    (4, 256, [64, 32], 32, "cp", True, False), (2, 300, [256], 256, "p", True, True), (5, 64, [32], 3, "c", False, False),
    (2, 1024, [128, 32], 128, "pc", False, False), (3, 50, [3], 1, "p", True, False), (2, 256, [96], 64, "c", True, True)])
def test_pw_linear_matches_torch(S, P, cins, cout, layouts, bias, out_pm):
    """rtk_pw_conv / rtk_pw_wgrad behind pw_linear (virtual concatenation, channel- and point-major operands, odd channel and
    position counts, column-sliced weights) against cat + einsum under autograd: output, weight / bias / input gradients."""
    from ratrack_amd.train_ops import pw_linear
    g = torch.Generator(DEV).manual_seed(S * 1000 + P)
    K = sum(cins)
    Wfull = (torch.randn(cout, K + 5, device=DEV, generator=g) / K ** 0.5).requires_grad_(True)      # the layer uses columns 3 .. 3+K of a wider matrix
    b = torch.randn(cout, device=DEV, generator=g).requires_grad_(True) if bias else None
    base = [torch.randn(S, c, P, device=DEV, generator=g) for c in cins]
    cot = torch.randn(S, cout, P, device=DEV, generator=g)
    res = []
    for ours in (True, False):
        Wfull.grad = None
        if b is not None:
            b.grad = None
        srcs = [(_pm(t) if l == "p" else t.clone()).requires_grad_(True) for t, l in zip(base, layouts)]
        if ours:
            cols, c = [], 3
            for ci in cins:
                cols.append(c)
                c += ci
            z = pw_linear(srcs, Wfull, b, cols=cols, out_point_major=out_pm)
            assert z.shape == (S, cout, P) and (cout == 1 or (z.stride(1) == 1) == out_pm)
        else:
            z = torch.einsum("ok,skp->sop", Wfull[:, 3:3 + K].double(), torch.cat(srcs, 1).double())
            if b is not None:
                z = z + b.double().view(1, -1, 1)
        (z * cot).sum().backward()
        res.append((z.detach().double(), Wfull.grad.clone().double(), None if b is None else b.grad.clone().double(), [t.grad.double() for t in srcs]))
    (za, wa, ba, xa), (zb, wb, bb, xb) = res
    tol = lambda r: 2e-5 * float(r.abs().max()) + 1e-6
    assert float((za - zb).abs().max()) <= tol(zb)
    assert float((wa - wb).abs().max()) <= 5e-5 * float(wb.abs().max()) + 1e-5
    assert float(wa[:, :3].abs().max()) == 0 and float(wa[:, 3 + K:].abs().max()) == 0      # columns outside the layer stay zero
    if b is not None:
        assert float((ba - bb).abs().max()) <= 5e-5 * float(bb.abs().max()) + 1e-5
    for a, r in zip(xa, xb):
        assert float((a - r).abs().max()) <= tol(r)


@pytest.mark.parametrize("S,P,cins,cout,groups,weighted", [(4, 256, [64, 64], 128, 2, True), (2, 242, [128, 32], 128, 1, True),
                                                           (4, 256, [256], 128, 1, False), (6, 100, [128], 64, 2, False),
                                                           (2, 256, [64], 32, 1, False)])
def test_pw_bn_relu_matches_torch(S, P, cins, cout, groups, weighted):
    """pw_bn_relu = conv1x1(cat) -> BatchNorm2d(train, per-group statistics, row weights) -> ReLU against the float64 framework
    formulation on the EXPANDED tensor (rows repeated by their integer weight): outputs, running statistics, all gradients."""
    from ratrack_amd.train_ops import pw_bn_relu
    g = torch.Generator(DEV).manual_seed(P + cout)
    K = sum(cins)
    base = [torch.randn(S, c, P, device=DEV, generator=g) for c in cins]
    W0 = torch.randn(cout, K, 1, 1, device=DEV, generator=g) / K ** 0.5
    rw = None
    if weighted:
        rw = torch.ones(S, P, device=DEV)
        rw[:, 0] = 5.0                                   # row 0 stands for 5 identical rows
        rw[:, P - 7:] = 0.0                              # dead rows
    cot = torch.randn(S, cout, P, device=DEV, generator=g)
    count = (S // groups) * (P if rw is None else int(rw[0].sum()))
    res = []
    for ours in (True, False):
        bn = nn.BatchNorm2d(cout).to(DEV)
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5, generator=g) if False else bn.weight.copy_(torch.linspace(0.5, 1.5, cout))
            bn.bias.copy_(torch.linspace(-0.3, 0.3, cout))
        W = W0.clone().requires_grad_(True)
        srcs = [t.clone().requires_grad_(True) for t in base]
        if ours:
            y = pw_bn_relu(srcs, W, bn, rw, count, groups)
            live = torch.ones(S, P, dtype=torch.bool, device=DEV) if rw is None else rw > 0
            (y * cot * live.unsqueeze(1)).sum().backward()
            res.append((y.detach().double(), W.grad.double(), bn.weight.grad.double(), bn.bias.grad.double(), [t.grad.double() for t in srcs],
                        bn.running_mean.clone().double(), bn.running_var.clone().double(), live))
        else:
            bn = bn.double()
            x = torch.cat([t.double() for t in srcs], 1)
            Wd = W.double()
            outs = []
            for gi in range(groups):
                sl = slice(gi * (S // groups), (gi + 1) * (S // groups))
                z = torch.einsum("ok,skp->sop", Wd[:, :, 0, 0], x[sl])
                if rw is None:
                    outs.append(torch.relu(bn(z.unsqueeze(-1))).squeeze(-1))
                else:       # expand: row r repeated rw[r] times
                    rep = rw[0].long()
                    ze = torch.repeat_interleave(z, rep, dim=2)
                    ye = torch.relu(bn(ze.unsqueeze(-1))).squeeze(-1)
                    first = torch.cumsum(rep, 0) - rep
                    yfull = torch.zeros_like(z)
                    live0 = rep > 0
                    yfull[:, :, live0] = ye[:, :, first[live0]]
                    # the gradient of the de-duplicated row is the SUM over its copies: give every copy the same cotangent
                    outs.append((yfull, ye, first, rep))
            if rw is None:
                y = torch.cat(outs, 0)
                (y * cot.double()).sum().backward()
            else:
                tot = 0
                ys = []
                for gi, (yfull, ye, first, rep) in enumerate(outs):
                    sl = slice(gi * (S // groups), (gi + 1) * (S // groups))
                    live0 = rep > 0
                    c = torch.zeros_like(ye)
                    c[:, :, first[live0]] = cot[sl].double()[:, :, live0]          # cotangent on the first copy only == on the row
                    tot = tot + (ye * c).sum()
                    ys.append(yfull)
                tot.backward()
                y = torch.cat(ys, 0)
            res.append((y.detach(), W.grad.double(), bn.weight.grad, bn.bias.grad, [t.grad.double() for t in srcs], bn.running_mean.clone(),
                        bn.running_var.clone(), None))
    a, r = res
    live = a[7].unsqueeze(1)
    assert float(((a[0] - r[0]) * live).abs().max()) <= 1e-4 * float(r[0].abs().max())
    for i in (1, 2, 3, 5, 6):
        assert float((a[i] - r[i]).abs().max()) <= 2e-4 * float(r[i].abs().max()) + 1e-6, i
    for x, y in zip(a[4], r[4]):
        assert float(((x - y) * live).abs().max()) <= 2e-4 * float(y.abs().max()) + 1e-6


@pytest.mark.parametrize("S,C,rows,ns,n_src", [(4, 16, 256, 4, 256), (3, 32, 242, 8, 242), (2, 64, 256, 32, 256), (2, 64, 512, 32, 512),
                                               (5, 16, 77, 4, 100), (2, 32, 300, 16, 1024), (8, 16, 256, 8, 256)])
def test_first_layer_backward_gather_form(S, C, rows, ns, n_src):
    """rtk_group_inverse_index + rtk_sa_first_layer_bwd (scatter turned into a gather, dWx fused) against the LDS-atomic scatter
    kernel + batched GEMM they replace; the inverse table itself against a stable argsort; run-to-run determinism."""
    from ratrack_amd import _lib
    from ratrack_amd import train_ops as T
    g = torch.Generator(DEV).manual_seed(rows + ns)
    P = rows * ns
    idx = torch.randint(0, n_src, (S, rows, ns), device=DEV, generator=g, dtype=torch.int32)
    idx[:, :, 1:] = torch.where(torch.rand(S, rows, ns - 1, device=DEV, generator=g) < 0.5, idx[:, :, :1].expand(-1, -1, ns - 1), idx[:, :, 1:])
    idx[0, : rows // 2] = 0                                                    # a heavily shared source point
    dz = torch.randn(S, C, rows, ns, device=DEV, generator=g)
    dxyz = torch.randn(S, 3, rows, ns, device=DEV, generator=g)
    off = torch.empty(S, n_src + 1, dtype=torch.int32, device=DEV)
    inv = torch.empty(S, P, dtype=torch.int16, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    _lib.call("rtk_group_inverse_index", S, n_src, P, idx.data_ptr(), off.data_ptr(), inv.data_ptr(), st)
    order = torch.sort(idx.view(S, P).long(), dim=1, stable=True).indices
    assert torch.equal(inv.long() & 0xffff, order)
    cnt = torch.zeros(S, n_src, dtype=torch.long, device=DEV).scatter_add_(1, idx.view(S, P).long(), torch.ones(S, P, dtype=torch.long, device=DEV))
    assert torch.equal(off[:, 1:].long(), cnt.cumsum(1)) and (off[:, 0] == 0).all()
    outs = []
    for _ in range(2):
        dproj = torch.empty(S, C, n_src, device=DEV)
        dwx = torch.zeros(C, 3, device=DEV)
        _lib.call("rtk_sa_first_layer_bwd", S, C, rows, ns, n_src, dz.data_ptr(), dxyz.data_ptr(), off.data_ptr(), inv.data_ptr(),
                  dproj.data_ptr(), dwx.data_ptr(), 3, torch.empty(S * C * 3, device=DEV).data_ptr(), st)
        outs.append((dproj, dwx))
    ref = torch.zeros(S, C, n_src, dtype=torch.float64, device=DEV).scatter_add_(2, idx.view(S, 1, P).long().expand(-1, C, -1), dz.view(S, C, P).double())
    assert float((outs[0][0].double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    ref_w = torch.einsum("scp,sap->ca", dz.view(S, C, P).double(), dxyz.view(S, 3, P).double())
    assert float((outs[0][1].double() - ref_w).abs().max()) <= 2e-5 * float(ref_w.abs().max()) + 1e-4
    # bit for bit from run to run: the gather sums are accumulated in per-plane fixed point, the samples' shares of dwx in a fixed order
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # without the workspace the shares of dwx go through float atomics: the same values up to the order of addition
    dproj, dwx = torch.empty(S, C, n_src, device=DEV), torch.zeros(C, 3, device=DEV)
    _lib.call("rtk_sa_first_layer_bwd", S, C, rows, ns, n_src, dz.data_ptr(), dxyz.data_ptr(), off.data_ptr(), inv.data_ptr(),
              dproj.data_ptr(), dwx.data_ptr(), 3, None, st)
    assert torch.equal(dproj, outs[0][0])
    assert float((dwx - outs[0][1]).abs().max()) <= 1e-5 * float(outs[0][1].abs().max())
    # the fixed-point unit follows every plane's own largest element: planes of zeros, of 1e-20, of 1e20 and a plane with a NaN
    dz2 = dz.clone()
    dz2[:, 0] = 0.0
    dz2[:, 1] *= 1e-20
    dz2[:, 2] *= 1e20
    dz2[0, 3, rows // 2, 1] = float("nan")
    _lib.call("rtk_sa_first_layer_bwd", S, C, rows, ns, n_src, dz2.data_ptr(), dxyz.data_ptr(), off.data_ptr(), inv.data_ptr(),
              dproj.data_ptr(), dwx.data_ptr(), 3, torch.empty(S * C * 3, device=DEV).data_ptr(), st)
    assert float(dproj[:, 0].abs().max()) == 0.0
    for c, f in ((1, 1e-20), (2, 1e20)):
        assert float((dproj[:, c].double() - ref[:, c] * f).abs().max()) <= 1e-5 * float(ref[:, c].abs().max()) * f
    assert torch.isnan(dproj[0, 3]).all() and torch.isfinite(dproj[1:, 3]).all() and torch.isfinite(dproj[:, 4:]).all()
    assert float((dproj[1:, 3].double() - ref[1:, 3]).abs().max()) <= 1e-5 * float(ref.abs().max())


@pytest.mark.parametrize("B,N,pretrain,vec", [(4, 256, False, False), (3, 242, True, False), (1, 256, False, True), (64, 256, False, False)])
def test_fused_backbone_loss_matches_framework_formulation(B, N, pretrain, vec):
    """rtk_backbone_loss (values + gradients, one launch) against ratrack_amd.loss.backbone_loss under autograd -- itself pinned
    to the reference's losses in tests/test_loss_metrics_cpu.py -- incl. a sample without positives (its segmentation term and
    gradient are zero) and the pre-training configuration (no gradient reaches the flow)."""
    from ratrack_amd import loss as L
    from ratrack_amd.train_ops import backbone_loss
    g = torch.Generator(DEV).manual_seed(B + N)
    pc1 = torch.randn(B, 3, N, device=DEV, generator=g) * 10
    gt = pc1 + torch.randn(B, 3, N, device=DEV, generator=g) * 0.3
    gt_cls = torch.rand(N if vec else B * N, device=DEV, generator=g).view((N,) if vec else (B, N)) > 0.7
    if not vec and B > 1:
        gt_cls[1] = False                                   # no positives in sample 1
    res = []
    for ours in (True, False):
        flow = (torch.randn(B, 3, N, device=DEV, generator=torch.Generator(DEV).manual_seed(1)) * 0.2).requires_grad_(True)
        cls = torch.rand(B, N, device=DEV, generator=torch.Generator(DEV).manual_seed(2)).clamp(1e-4, 1 - 1e-4).requires_grad_(True)
        total, items = backbone_loss(pc1, flow, cls, gt, gt_cls, pretrain) if ours else L.backbone_loss(pc1 + flow, cls, gt, gt_cls, pretrain)
        total.backward()
        res.append(([float(items[k]) for k in ("Loss", "SceneFlowLoss", "TrackingLoss", "SegLoss")], flow.grad, cls.grad))
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=2e-5, atol=1e-7)
    if pretrain:
        assert res[0][1] is None and (res[1][1] is None or float(res[1][1].abs().max()) == 0)
    else:
        assert float((res[0][1] - res[1][1]).abs().max()) <= 1e-5 * float(res[1][1].abs().max())
    assert float((res[0][2] - res[1][2]).abs().max()) <= 2e-5 * float(res[1][2].abs().max())
    if not vec and B > 1:
        assert float(res[0][2][1].abs().max()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("split", [True, False])
@pytest.mark.parametrize("B,N", [(2, 256), (3, 77)])
def test_cost_volume_train_forward_keeps_what_the_backward_needs(B, N, split, monkeypatch):
    """rtk_cost_volume_train = rtk_cost_volume (bit for bit) + the three activations and the sign masks of the first two, in the
    backward kernel's lane order: bit 4v + r of word (position, g) <-> channel 16v + 4g + r.  The same for the split-bf16 pair
    (rtk_cost_volume_split_train / rtk_cost_volume_split), which writes the same formats from a different tile."""
    from ratrack_amd import _lib, train_ops as T
    from ratrack_amd.model_utils import knn_point
    monkeypatch.setattr(T, "CV_SPLIT", split)
    g = torch.Generator(DEV).manual_seed(4)
    r = lambda *s: torch.randn(*s, device=DEV, generator=g)
    d = synth.make_frame_pairs(B, N, 5)
    x1 = torch.from_numpy(d["pc1"]).to(DEV).permute(0, 2, 1).contiguous()
    x2 = torch.from_numpy(d["pc2"]).to(DEV).permute(0, 2, 1).contiguous()
    knn = knn_point(16, x2, x1).contiguous()
    p1, p2 = r(B * N, 256) * 0.5, r(B * N, 256) * 0.5
    W = T._CvWeights(r(256, 3), r(256, 256) * 0.06, r(256) * 0.1, r(256, 256) * 0.06, r(256) * 0.1, r(8, 3), r(8), r(8, 8) * 0.3, r(8),
                     r(256, 8) * 0.3, r(256), backward=True)
    M = B * N * 16
    st = torch.cuda.current_stream().cuda_stream
    out_a, out_b = torch.empty(B * N, 256, device=DEV), torch.empty(B * N, 256, device=DEV)
    acts = torch.full((3, M, 256), float("nan"), device=DEV)
    masks = torch.zeros(2, M, 4, dtype=torch.int64, device=DEV)
    common = (B, N, N, x1.data_ptr(), x2.data_ptr(), knn.data_ptr(), p1.data_ptr(), p2.data_ptr(), W.wd.data_ptr())
    common += (W.split.data_ptr(), W.split_scales.data_ptr(), W.b2.data_ptr(), W.b3.data_ptr(), W.wn) if split else (W.layers, W.wn)
    _lib.call("rtk_cost_volume_split" if split else "rtk_cost_volume", *common, out_a.data_ptr(), 256, st)
    amax = torch.zeros(2, device=DEV)
    _lib.call("rtk_cost_volume_split_train" if split else "rtk_cost_volume_train", *common, out_b.data_ptr(), 256, acts[0].data_ptr(),
              acts[1].data_ptr(), acts[2].data_ptr(), masks[0].data_ptr(), masks[1].data_ptr(), *((amax.data_ptr(),) if split else ()), st)
    torch.cuda.synchronize()
    assert torch.equal(out_a, out_b)
    assert torch.isfinite(acts).all()
    if split:      # the tensors' largest |element|, folded in by the kernel for the weight-gradient contraction's scales
        assert float(amax[0]) == float(acts[0].abs().max()) and float(amax[1]) == float(acts[1].abs().max())
    # layer 1 from its definition: leaky(p1[i] + p2[nbr] + Wd d)   (the kernel's offset product is an MFMA: compare with a tolerance)
    nbr = (knn + (torch.arange(B, device=DEV) * N).view(B, 1, 1)).view(-1)
    dvec = (x2.reshape(B * N, 3)[nbr] - x1.reshape(B * N, 3).repeat_interleave(16, 0))
    a1 = torch.nn.functional.leaky_relu(p1.repeat_interleave(16, 0) + p2[nbr] + dvec @ r_wd(W, DEV).t(), 0.1)
    assert float((acts[0] - a1).abs().max()) <= 1e-4 * float(a1.abs().max())
    for a, m in ((acts[0], masks[0]), (acts[1], masks[1])):
        words = m.view(M, 4, 1)                                                    # (position, g): 64 bits
        bits = (words >> torch.arange(64, device=DEV).view(1, 1, 64)) & 1          # bit 4v + r
        got = bits.view(M, 4, 16, 4).permute(0, 2, 1, 3).reshape(M, 256).bool()    # channel 16v + 4g + r
        assert torch.equal(got, a > 0)


def r_wd(W, dev):
    """The (256, 3) offset weights back from the packed [16][64] image of _CvWeights (fragment v, lane (g, i): row 16v + i, column g)."""
    img = W.wd.view(16, 4, 16)                                                     # [v][g][i]
    return img.permute(0, 2, 1).reshape(256, 4)[:, :3].contiguous()


@pytest.mark.gpu
def test_weight_gradient_operators_are_deterministic():
    """Round 2 replaced the float atomics of the weight-gradient operators by workgroup partials added in a fixed order: the same
    inputs give the same bits."""
    from ratrack_amd import train_ops as T
    g = torch.Generator(DEV).manual_seed(8)
    r = lambda *s: torch.randn(*s, device=DEV, generator=g)
    srcs = [r(24, 64, 200), r(24, 35, 200)]
    W, dz = r(128, 99), r(24, 128, 200)
    runs = [T._pw_backward([False, False], srcs, [0, 64], W, dz, True)[:3] for _ in range(3)]
    for dW, db, _ in runs[1:]:
        assert torch.equal(dW, runs[0][0]) and torch.equal(db, runs[0][1])
    M, C = 5000, 256
    d4, dq3, dt2 = r(M, 4), r(M, C), r(M, 8)
    wa, ba, wb, bb, wc = r(8, 3), r(8), r(8, 8), r(8), r(C, 8)
    runs = [T._weightnet_backward(d4, dq3, dt2, wa, ba, wb, bb, wc) for _ in range(3)]
    for res in runs[1:]:
        for a, b in zip(res, runs[0]):
            assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("S,C,N", [(4, 128, 256), (2, 37, 100), (1, 5, 1)])
def test_gmax_cat_matches_max_expand_cat(S, C, N):
    """rtk_gmax_cat (models/track4d.py:92-95 as one kernel each way) against torch.max / expand / cat and their autograd: values exactly,
    gradients to summation order; ties (repeated columns, as in padded clouds) send the gradient to the FIRST maximum like torch.max."""
    from ratrack_amd import train_ops as T
    g = torch.Generator(DEV).manual_seed(S * 1000 + N)
    f = torch.randn(S, C, N, device=DEV, generator=g)
    if N > 8:
        f[:, :, N - 4:] = f[:, :, :1]                      # padding columns: copies of point 0
        f[:, ::3, 0] = f.max(-1)[0][:, ::3] + 1.0          # ... which is the maximum of every third channel: a tie of five columns
    w = torch.randn(S, 2 * C, N, device=DEV, generator=g)
    a = f.clone().requires_grad_()
    ref = torch.cat((a, a.max(-1)[0].unsqueeze(2).expand(-1, -1, N)), dim=1)
    (ref * w).sum().backward()
    b = f.clone().requires_grad_()
    out = T.gmax_cat(b)
    (out * w).sum().backward()
    assert torch.equal(out, ref)
    assert torch.allclose(b.grad, a.grad, rtol=1e-5, atol=1e-5 * float(a.grad.abs().max()))


@pytest.mark.gpu
def test_gru_parameter_pack_is_stack_and_transpose():
    from ratrack_amd import _lib, train_ops as T
    import ctypes
    gru = torch.nn.GRU(128, 128, num_layers=5).to(DEV)
    L, H = 5, 128
    params = []
    for l in range(L):
        params += [getattr(gru, "weight_ih_l%d" % l), getattr(gru, "weight_hh_l%d" % l), getattr(gru, "bias_ih_l%d" % l), getattr(gru, "bias_hh_l%d" % l)]
    outs = [torch.empty(L, 3 * H, H, device=DEV), torch.empty(L, H, 3 * H, device=DEV), torch.empty(L, 3 * H, H, device=DEV),
            torch.empty(L, H, 3 * H, device=DEV), torch.empty(L, 3 * H, device=DEV), torch.empty(L, 3 * H, device=DEV)]
    ptrs = (ctypes.c_void_p * (4 * L))(*[p.data_ptr() for p in params])
    _lib.call("rtk_gru_pack_params", L, H, ptrs, *[o.data_ptr() for o in outs], torch.cuda.current_stream().cuda_stream)
    w_ih, w_hh = torch.stack(params[0::4]), torch.stack(params[1::4])
    for got, want in zip(outs, [w_ih, w_ih.transpose(1, 2), w_hh, w_hh.transpose(1, 2), torch.stack(params[2::4]), torch.stack(params[3::4])]):
        assert torch.equal(got, want.contiguous())


@pytest.mark.gpu
def test_deferred_weight_gradients_equal_immediate_ones():
    """Trainer queues the per-point layers' weight gradients during the backward and issues them eight per launch at its end
    (rtk_pw_wgrad_multi), delivering them to .grad itself.  Same kernels, same partial sums: bit-identical gradients -- for a parameter
    used once, a parameter used twice (the second delivery adds), a layer with bias, and input gradients untouched."""
    from ratrack_amd import train_ops as T
    g = torch.Generator(DEV).manual_seed(21)
    r = lambda *s: torch.randn(*s, device=DEV, generator=g)
    x1, x2 = r(6, 70, 130).requires_grad_(), r(6, 33, 130).requires_grad_()
    lin = torch.nn.Conv1d(103, 96, 1).to(DEV)
    conv = torch.nn.Conv2d(96, 48, 1, bias=False).to(DEV)
    bn = torch.nn.BatchNorm2d(48).to(DEV)
    many = [torch.nn.Conv1d(48, 20 + 7 * k, 1).to(DEV) for k in range(11)]      # more jobs than one launch takes

    def run(deferred):
        for p in [x1, x2] + list(lin.parameters()) + list(conv.parameters()) + list(bn.parameters()) + [q for m in many for q in m.parameters()]:
            p.grad = None
        bn.running_mean.zero_(); bn.running_var.fill_(1.0); bn.num_batches_tracked.zero_()
        y = T.pw_linear([x1, x2], lin.weight, lin.bias)
        z = T.pw_bn_relu([y], conv.weight, bn) + T.pw_bn_relu([y * 0.5], conv.weight, bn)      # conv.weight twice
        loss = sum((T.pw_linear([z], m.weight, m.bias) ** 2).mean() for m in many)
        if deferred:
            T.begin_deferred_wgrads()
        try:
            loss.backward()
            if deferred:
                assert lin.weight.grad is None and conv.weight.grad is None, "queued gradients must not travel through autograd"
        finally:
            if deferred:
                T.flush_deferred_wgrads()
        names = [x1, x2, lin.weight, lin.bias, conv.weight, bn.weight, bn.bias] + [q for m in many for q in m.parameters()]
        return [p.grad.clone() for p in names]

    a, b = run(False), run(True)
    for k, (u, v) in enumerate(zip(a, b)):
        assert torch.equal(u, v), (k, float((u - v).abs().max()))
    assert T._DEFERRED is None


@pytest.mark.gpu
def test_weight_gradient_workspaces_of_any_size_give_the_same_sums():
    """The partial-block workspaces only set how far the position axis is split: no workspace (one workgroup per block), a
    one-block workspace and the full one agree to rounding; an undersized workspace of rtk_conv_wgrad / rtk_weightnet_bwd is refused."""
    from ratrack_amd import _lib, train_ops as T
    g = torch.Generator(DEV).manual_seed(9)
    r = lambda *s: torch.randn(*s, device=DEV, generator=g)
    st = torch.cuda.current_stream().cuda_stream
    S, P, ci, co = 16, 200, 70, 96
    x, dz = r(S, ci, P), r(S, co, P)
    ref = torch.einsum("sop,skp->ok", dz.double(), x.double()).float()
    for nfloats in (0, 4096 * 4, 4 << 20):
        ws = torch.empty(max(nfloats, 1), device=DEV)
        dW, db = torch.zeros(co, ci, device=DEV), torch.zeros(co, device=DEV)
        _lib.call("rtk_pw_wgrad", S, P, T._pw_operands([dz], [0]), 1, T._pw_operands([x], [0]), dW.data_ptr(), dW.stride(0), db.data_ptr(),
                  ws.data_ptr() if nfloats else None, nfloats, st)
        assert float((dW - ref).norm()) <= 1e-5 * float(ref.norm()), nfloats
        assert torch.allclose(db, dz.sum((0, 2)), rtol=1e-4, atol=1e-3)
    # rtk_weightnet_bwd: one partial vector is enough; less is an error, not a silent fallback
    M, C = 3000, 64
    d4, dq3, dt2 = r(M, 4), r(M, C), r(M, 8)
    wa, ba, wb, bb = r(8, 3), r(8), r(8, 8), r(8)
    outs = []
    for vectors in (1, 1024):
        ws = torch.empty(vectors * ((9 * C + 107) & ~3), device=DEV)
        z = [torch.zeros(n, device=DEV) for n in (24, 8, 64, 8, C * 8, C)]
        _lib.call("rtk_weightnet_bwd", M, C, d4.data_ptr(), dq3.data_ptr(), dt2.data_ptr(), wa.data_ptr(), ba.data_ptr(), wb.data_ptr(),
                  bb.data_ptr(), *[t.data_ptr() for t in z], ws.data_ptr(), ws.numel(), st)
        outs.append(z)
    for a, b in zip(*outs):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-3 * float(b.abs().max()))
    with pytest.raises(_lib.RtkError):
        ws = torch.empty(16, device=DEV)
        _lib.call("rtk_weightnet_bwd", M, C, d4.data_ptr(), dq3.data_ptr(), dt2.data_ptr(), wa.data_ptr(), ba.data_ptr(), wb.data_ptr(),
                  bb.data_ptr(), *[t.data_ptr() for t in outs[0]], ws.data_ptr(), ws.numel(), st)


@pytest.mark.gpu
@pytest.mark.parametrize("m", [1, 15, 16, 17, 1000, 4096, 5152, 40000])
def test_position_contraction_on_the_split_path_carries_fp32_accuracy(m):
    """rtk_tn_gemm256_split (the cost volume's two weight gradients, x^T y over m rows, three fp16 MFMA products per fp32 product under one
    power-of-two scale per tensor) against float64, with torch's fp32 GEMM as the yardstick; row counts that end inside a 16-row step, inside a thread's four rows,
    and slabs that are rounded up with empty steps (out-of-range rows must read as zero)."""
    from ratrack_amd import _lib, train_ops as T
    g = torch.Generator(DEV).manual_seed(m)
    x = torch.randn(2, m, 256, device=DEV, generator=g) * torch.rand(2, m, 1, device=DEV, generator=g) * 3
    y = torch.relu(torch.randn(2, m, 256, device=DEV, generator=g)) + 0.25      # same-sign operand: no cancellation to hide behind
    ref = torch.stack([x[k].double().t() @ y[k].double() for k in range(2)])
    lib = torch.stack([x[k].t() @ y[k] for k in range(2)])
    out = T.tn_gemm256([(x[0], y[0]), (x[1], y[1])])
    scale = float(ref.abs().max())
    err, err_lib = float((out - ref).abs().max()) / scale, float((lib - ref).abs().max()) / scale
    assert err <= max(2.0 * err_lib, 3e-7), (err, err_lib)
    # one job, the smallest legal workspace (one slab per job), a pitched output that must be left alone outside its 256 columns
    big = torch.full((256, 300), 7.0, device=DEV)
    ws = torch.empty(65536, device=DEV)
    job = (T._TnJob * 1)()
    job[0].x, job[0].y, job[0].out, job[0].out_pitch = x[1].data_ptr(), y[1].data_ptr(), big.data_ptr(), 300
    am = torch.stack([x[1].abs().max(), y[1].abs().max()]).contiguous()
    job[0].x_amax, job[0].y_amax = am[0:].data_ptr(), am[1:].data_ptr()
    _lib.call("rtk_tn_gemm256_split", 1, job, m, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    assert float((big[:, :256] - ref[1]).abs().max()) / scale <= max(2.0 * err_lib, 3e-7)
    assert bool((big[:, 256:] == 7.0).all())
    assert torch.equal(T.tn_gemm256([(x[0], y[0]), (x[1], y[1])]), out)      # no float atomics: the same bits every time
    # the scales are powers of two taken from the data: operands scaled by 2^k give the same bits times 2^k (1e-30 .. 1e+30 operands)
    for k in (-100.0, 60.0):
        out2 = T.tn_gemm256([(x[0] * 2.0 ** k, y[0] * 2.0 ** -k), (x[1] * 2.0 ** k, y[1])])
        assert torch.equal(out2[0], out[0]) and torch.equal(out2[1], out[1] * 2.0 ** k)
    # rtk_absmax: the largest |element| as an unsigned maximum on the bits
    a = torch.zeros(1, device=DEV)
    _lib.call("rtk_absmax", x.data_ptr(), x.numel(), a.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert float(a) == float(x.abs().max())
    with pytest.raises(_lib.RtkError):
        _lib.call("rtk_tn_gemm256_split", 1, job, m, ws.data_ptr(), 65535, torch.cuda.current_stream().cuda_stream)


@pytest.mark.gpu
def test_tn_gemm256_split_outlier_rows():
    """The range contract of rtk_tn_gemm256_split (include/rtk_train.h; round-5 advice): ONE power-of-two scale per tensor, so a row
    (position) 1e8 above the rest pushes the others' low pieces into fp16 subnormals.  Pinned here: (i) the product stays within one fp32
    rounding of the float64 result RELATIVE TO THE LARGEST OUTPUT ELEMENT -- the error of any fp32 GEMM on such data --; (ii) restricted to
    the small rows alone (the outlier removed) the kernel is exact to fp32 again -- the floor is a property of the shared scale, not a
    defect of the small rows' path; (iii) a non-finite element makes the affected outputs non-finite instead of silently finite-and-wrong."""
    from ratrack_amd import train_ops as T
    g = torch.Generator(DEV).manual_seed(11)
    m = 4096
    x = torch.randn(m, 256, device=DEV, generator=g)
    y = torch.relu(torch.randn(m, 256, device=DEV, generator=g)) + 0.25
    xo = x.clone()
    xo[77] *= 1e8                                                        # one position's gradient 1e8 above the rest
    ref = xo.double().t() @ y.double()
    out = T.tn_gemm256([(xo, y)])[0]
    lib = xo.t() @ y
    scale = float(ref.abs().max())
    err, err_lib = float((out - ref).abs().max()) / scale, float((lib - ref).abs().max()) / scale
    assert err <= max(2.0 * err_lib, 3e-7), (err, err_lib)
    # what the other 4095 rows contributed is below that rounding -- by construction of the data, for any fp32 evaluation: documented, not hidden
    rest = (x.double().t() @ y.double() - torch.outer(x[77].double(), y[77].double())).abs().max()
    assert float(rest) / scale < 1e-5
    # (ii) without the outlier the same rows come out at fp32 accuracy
    xs = x.clone()
    xs[77] = 0
    ref_s = xs.double().t() @ y.double()
    out_s = T.tn_gemm256([(xs, y)])[0]
    assert float((out_s - ref_s).abs().max()) / float(ref_s.abs().max()) <= 3e-7
    # (iii) non-finite in, non-finite out
    xi = x.clone()
    xi[5, 9] = float("inf")
    out_i = T.tn_gemm256([(xi, y)])[0]
    assert not bool(torch.isfinite(out_i[9]).all())


@pytest.mark.gpu
def test_position_contraction_in_row_chunks(monkeypatch):
    """More rows than one launch addresses (m >= 2^22: B * N1 * 16 positions at B = 64, N = 4096) go in row chunks whose products are
    added (round-3 advisor: the backward raised there).  Exercised with a small chunk limit; chunk boundaries off the 16-row step."""
    from ratrack_amd import train_ops as T
    g = torch.Generator(DEV).manual_seed(5)
    m = 40000
    x, y = torch.randn(m, 256, device=DEV, generator=g), torch.rand(m, 256, device=DEV, generator=g)
    ref = x.double().t() @ y.double()
    whole = T.tn_gemm256([(x, y)])[0]
    monkeypatch.setattr(T, "TN_MAX_ROWS", 16 * 617 + 4)
    parts = T.tn_gemm256([(x, y)])[0]
    scale = float(ref.abs().max())
    err_lib = float((x.t() @ y - ref).abs().max()) / scale          # torch's fp32 GEMM as the yardstick
    assert float((whole - ref).abs().max()) / scale <= max(2.0 * err_lib, 3e-7)
    assert float((parts - ref).abs().max()) / scale <= max(2.0 * err_lib, 3e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,n,m", [(3, 13, 77, 50), (4, 128, 256, 256), (2, 64, 1024, 512)])
def test_three_interpolate_backward_gather_form(B, C, n, m):
    """rtk_three_interpolate_grad_gather over the inverse table of the interpolation indices against the reference-style scatter
    (rtk_three_interpolate_grad_set) and a float64 scatter_add: known points nobody references get exact zeros."""
    from ratrack_amd import _lib, train_ops as T
    g = torch.Generator(DEV).manual_seed(12)
    go = torch.randn(B, C, n, device=DEV, generator=g)
    idx = torch.randint(0, max(m - 5, 1), (B, n, 3), device=DEV, generator=g, dtype=torch.int32)      # the last known points stay unused
    w = torch.rand(B, n, 3, device=DEV, generator=g)
    st = torch.cuda.current_stream().cuda_stream
    off = torch.empty(B, m + 1, dtype=torch.int32, device=DEV)
    inv = torch.empty(B, 3 * n, dtype=torch.int16, device=DEV)
    _lib.call("rtk_group_inverse_index", B, m, 3 * n, idx.data_ptr(), off.data_ptr(), inv.data_ptr(), st)
    a = torch.full((B, C, m), float("nan"), device=DEV)
    b = torch.full((B, C, m), float("nan"), device=DEV)
    _lib.call("rtk_three_interpolate_grad_gather", B, C, n, m, go.data_ptr(), w.data_ptr(), off.data_ptr(), inv.data_ptr(), a.data_ptr(), None, st)
    _lib.call("rtk_three_interpolate_grad_set", B, C, n, m, go.data_ptr(), idx.data_ptr(), w.data_ptr(), b.data_ptr(), st)
    ref = torch.zeros(B, C, m, device=DEV, dtype=torch.float64)
    for k in range(3):
        ref.scatter_add_(2, idx[:, :, k].long().unsqueeze(1).expand(-1, C, -1), (go * w[:, :, k].unsqueeze(1)).double())
    assert torch.isfinite(a).all()
    assert float((a.double() - ref).abs().max()) <= 1e-5 * max(float(ref.abs().max()), 1.0)
    assert float((a - b).abs().max()) <= 1e-5 * max(float(ref.abs().max()), 1.0)
    assert torch.equal(a[:, :, m - 5:], torch.zeros_like(a[:, :, m - 5:])) if m > 5 else True
    # through the autograd operator
    feats = torch.randn(B, C, m, device=DEV, generator=g, requires_grad=True)
    out = T.three_interpolate(feats, idx, w, (off, inv))
    out.backward(go)
    assert float((feats.grad.double() - ref).abs().max()) <= 1e-5 * max(float(ref.abs().max()), 1.0)
    # padded clouds: the unknown points from n_valid[b] on are copies of point 0 (same indices, same weights).  The table is built
    # without their positions and the kernel folds their gradient into point 0's: the same sums as the full scatter.
    nv = torch.tensor([max(1, n - 7 * (s + 1)) for s in range(B)], dtype=torch.int32, device=DEV)
    pad = torch.arange(n, device=DEV)[None, :] >= nv[:, None].long()
    idx_p = torch.where(pad[:, :, None], idx[:, :1], idx).contiguous()
    w_p = torch.where(pad[:, :, None], w[:, :1], w).contiguous()
    job = [(m, 3 * n, idx_p, off, inv, nv, 3)]
    T.group_inverse_index_multi(B, job)
    assert torch.equal(off[:, -1], 3 * nv)
    c = torch.full((B, C, m), float("nan"), device=DEV)
    _lib.call("rtk_three_interpolate_grad_gather", B, C, n, m, go.data_ptr(), w_p.data_ptr(), off.data_ptr(), inv.data_ptr(), c.data_ptr(),
              nv.data_ptr(), st)
    ref = torch.zeros(B, C, m, device=DEV, dtype=torch.float64)
    for k in range(3):
        ref.scatter_add_(2, idx_p[:, :, k].long().unsqueeze(1).expand(-1, C, -1), (go * w_p[:, :, k].unsqueeze(1)).double())
    assert float((c.double() - ref).abs().max()) <= 2e-5 * max(float(ref.abs().max()), 1.0)


def test_cost_volume_split_train_and_backward_agree_with_fp32_mfma_kernels(monkeypatch):
    """The split-bf16 kernels (csrc/fused_split.hip) and the fp32-input MFMA kernels (csrc/fused_group.hip) are two implementations
    of the same operator pair with the same tensor formats: outputs, saved activations, sign masks and every gradient agree to
    fp32 rounding (masks: except where an activation is a rounding error away from zero)."""
    from ratrack_amd import train_ops as T
    torch.manual_seed(5)
    B, n = 3, 100                                     # 2-D grid (B % 8 != 0), last workgroup iteration partly empty
    r = lambda *sh: torch.randn(*sh, device=DEV)
    xyz1, xyz2 = r(B, n, 3), r(B, n, 3)
    knn = torch.randint(0, n, (B, n, 16), device=DEV)
    par = [r(256, 3) * 0.3, r(256, 256) * 0.06, r(256) * 0.1, r(256, 256) * 0.06, r(256) * 0.1, r(8, 3), r(8) * 0.1, r(8, 8) * 0.4, r(8) * 0.1,
           r(256, 8) * 0.4, r(256) * 0.1]
    p1, p2, dout = r(B * n, 256), r(B * n, 256), r(B * n, 256)
    res = []
    for split in (False, True):
        monkeypatch.setattr(T, "CV_SPLIT", split)
        leaves = [t.clone().requires_grad_(True) for t in [p1, p2] + par]
        out = T.cost_volume(*leaves, xyz1, xyz2, knn)
        acts, masks = out.grad_fn.saved_tensors[:2]
        out.backward(dout)
        res.append((out.detach(), acts.clone(), masks.clone(), [t.grad for t in leaves]))
    (o0, a0, m0, g0), (o1, a1, m1, g1) = res
    assert rel_err(o1.cpu(), o0.cpu()) < 5e-6
    assert rel_err(a1.cpu(), a0.cpu()) < 5e-6
    differing = (m0 ^ m1) != 0
    assert differing.float().mean() < 1e-3            # a word differs only where some activation sits within rounding of zero
    names = ["p1", "p2", "wd", "w2", "b2", "w3", "b3", "wa", "ba", "wb", "bb", "wc", "bc"]
    for name, x, y in zip(names, g0, g1):
        assert rel_err(y.cpu(), x.cpu()) < 2e-5, name


def test_split_packer_transposed():
    from ratrack_amd import _lib, fused as F
    torch.manual_seed(2)
    w = torch.randn(256, 256, device=DEV)
    img, inv = F.pack_split_device(w, transposed=True)
    himg, hinv = F.pack_layer_split(w.t().contiguous())
    assert torch.equal(img, himg) and float(inv) == hinv


@pytest.mark.gpu
def test_single_launch_adam_matches_torch_adam():
    """ratrack_amd.optim.FusedAdam against torch.optim.Adam (the reference's optimizer, main.py:61) over several steps: ragged tensor
    sizes (partial last workgroups), a parameter that has no gradient at first and gets one later, a decaying learning rate held in
    a device tensor (StepLR), weight decay."""
    from ratrack_amd.optim import FusedAdam
    torch.manual_seed(0)
    shapes = [(5,), (64, 64), (4097,), (3, 7, 11), (256, 515), (1,)]
    ref = [torch.randn(*s, device=DEV).requires_grad_(True) for s in shapes]
    got = [p.detach().clone().requires_grad_(True) for p in ref]
    o_ref = torch.optim.Adam(ref, lr=1e-2, weight_decay=1e-3)
    lr = torch.tensor(1e-2, device=DEV)
    o_got = FusedAdam(got, lr=lr, weight_decay=1e-3)
    s_ref = torch.optim.lr_scheduler.StepLR(o_ref, step_size=1, gamma=0.9)
    s_got = torch.optim.lr_scheduler.StepLR(o_got, step_size=1, gamma=0.9)
    for it in range(6):
        for k, (a, b) in enumerate(zip(ref, got)):
            if k == 3 and it < 2:
                a.grad = b.grad = None                                        # skipped by both, moments untouched
                continue
            gr = torch.randn_like(a)
            a.grad, b.grad = gr.clone(), gr.clone()
        o_ref.step(); o_got.step()
        s_ref.step(); s_got.step()
    assert o_got._steps.tolist() == [6.0, 6.0, 6.0, 4.0, 6.0, 6.0] and int(o_got._ticket) == 0      # per-parameter counts, as torch's
    for a, b in zip(ref, got):
        np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().cpu().numpy(), rtol=2e-5, atol=1e-7)
    # state layout of torch.optim.Adam
    st = o_got.state_dict()["state"]
    assert set(st[0].keys()) == {"step", "exp_avg", "exp_avg_sq"}
