"""GPU: size-independent properties at BASELINE.json's full sizes: batch consistency (a sample's result does not depend on
its batch neighbours), determinism, the module path as a second implementation, the pipelined graph path, and the N=1024 stress
configuration.  (The comparison with the CPU oracle at these sizes is tests/test_fullsize_oracle_gpu.py.)"""
import numpy as np
import pytest
import torch

from ratrack_amd import fused as F
from ratrack_amd import synth
from ratrack_amd.track4d import Args, Track4D

from _util import RTOL, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
NAMES = ["flow", "h", "cls", "cor", "pc1_features", "pc2_features", "prop"]


def make(b, n, case):
    net = Track4D(Args()).to(DEV).eval()
    synth.fill_state_dict(net.state_dict())
    net.invalidate_fused()
    d = synth.make_frame_pairs(b, n, case)
    t = [torch.from_numpy(d[k]).to(DEV) for k in ("pc1", "pc2", "feature1", "feature2")]
    return net, t


def test_b64_n256_batch_consistency_and_determinism():
    net, t = make(64, 256, 1000)
    h = torch.randn(5, 64, 128, device=DEV) * 0.1
    with torch.no_grad():
        full = net.backbone(*t, h)
        again = net.backbone(*t, h)
        for a, b in zip(full, again):
            assert torch.equal(a, b)                       # no atomics / races on the fused path: bit-reproducible
        for i in (0, 17, 63):
            one = net.backbone(*[x[i:i + 1].contiguous() for x in t], h[:, i:i + 1].contiguous())
            for name, a, b in zip(NAMES, full, one):
                a_i = a[:, i:i + 1] if name == "h" else a[i:i + 1]
                assert torch.equal(a_i, b), "sample %d: %s depends on the batch" % (i, name)
        assert all(torch.isfinite(x).all() for x in full)


def test_b32_n256_config2_forward():
    """BASELINE config 2 at its exact size (B=32, N=256, full backbone + scene-flow forward): fused path vs module path within
    the north-star tolerance, batch consistency bit for bit, EPE of the two paths against the synthetic ground truth equal."""
    from ratrack_amd.metrics import eval_scene_flow
    net, t = make(32, 256, 1010)
    d = synth.make_frame_pairs(32, 256, 1010)
    with torch.no_grad():
        full = net.backbone(*t, None)
        for i in (0, 9, 31):
            one = net.backbone(*[x[i:i + 1].contiguous() for x in t], None)
            for name, a, b in zip(NAMES, full, one):
                a_i = a[:, i:i + 1] if name == "h" else a[i:i + 1]
                assert torch.equal(a_i, b), "sample %d: %s depends on the batch" % (i, name)
        net._use_fused = False
        ref = net.backbone(*t, None)
    for name, a, b in zip(NAMES, full, ref):
        assert rel_err(a.cpu(), b.cpu()) <= RTOL, name
    gt_warp, static = torch.from_numpy(d["gt_warp"]), torch.from_numpy(~d["gt_cls"][0]).int()
    epe = [eval_scene_flow(t[0].cpu(), (t[0] + o[0]).cpu(), gt_warp, static)["epe"] for o in (full, ref)]
    assert abs(epe[0] - epe[1]) <= 1e-4 * max(epe[1], 1.0), epe


def test_b64_graphed_train_step_config3():
    """BASELINE config 3 at its full size: the captured train step (forward + loss + backward + Adam, one hipGraph) at
    B=64, N=256 reproduces the eager step's losses batch by batch (lr = 0: parameters frozen, BatchNorm statistics live)."""
    from ratrack_amd.train import Trainer
    from _util import reference_state_dict
    batches = []
    for i in range(5):
        d = synth.make_frame_pairs(64, 256, 1020 + i)
        batches.append({k: torch.from_numpy(v).to(DEV) for k, v in d.items()})
    res = []
    for graph in (False, True):
        net = Track4D(Args()).to(DEV)
        net.load_state_dict(reference_state_dict(DEV), strict=True)
        tr = Trainer(net, graph=graph, lr=0.0)
        h = torch.zeros(5, 64, 128, device=DEV)
        losses = []
        for t in batches:
            items, _ = tr.step(t["pc1"], t["pc2"], t["feature1"], t["feature2"], t["gt_warp"], t["gt_cls"], h)
            losses.append([float(items[k]) for k in ("Loss", "SceneFlowLoss", "SegLoss")])
        res.append(np.array(losses))
    assert np.isfinite(res[1]).all()
    np.testing.assert_allclose(res[1], res[0], rtol=1e-5)


def test_b64_fused_matches_module_path():
    net, t = make(64, 256, 1001)
    with torch.no_grad():
        fused = net.backbone(*t, None)
        net._use_fused = False
        ref = net.backbone(*t, None)
    for name, a, b in zip(NAMES, fused, ref):
        assert rel_err(a.cpu(), b.cpu()) <= RTOL, name


@pytest.mark.parametrize("depth", [2, 4])      # 2: forked geometry streams; 4 (the bench default): one stream per batch in flight
def test_graph_pipeline_matches_eager(depth):
    net, t = make(8, 256, 1002)
    h = torch.zeros(5, 8, 128, device=DEV)
    with torch.no_grad():
        ref = [x.clone() for x in net.backbone(*t, h)]
        pipe = F.GraphPipeline(net._fused, (*t, h), depth=depth)
        assert all(e.use_side_stream == (depth <= 2) for e in pipe.engines)
        outs = [pipe.submit(*t, h) for _ in range(2 * depth + 1)]      # slots are reused round-robin
        pipe.drain()
        torch.cuda.synchronize()
        for a, b in zip(outs[-1], ref):
            assert torch.equal(a, b)
        # new inputs go through the static buffers
        net2, t2 = make(8, 256, 1003)
        ref2 = [x.clone() for x in net.backbone(*t2, h)]
        out2 = pipe.submit(*t2, h)
        pipe.drain()
        torch.cuda.synchronize()
        for a, b in zip(out2, ref2):
            assert torch.equal(a, b)


def test_n1024_stress_config():
    """BASELINE config 5: N=1024 (radar_5frames clouds): FPS genuinely down-samples, FP1 genuinely interpolates."""
    net, t = make(4, 1024, 1004)
    with torch.no_grad():
        fused = net.backbone(*t, None)
        net._use_fused = False
        ref = net.backbone(*t, None)
    for name, a, b in zip(NAMES, fused, ref):
        assert rel_err(a.cpu(), b.cpu()) <= RTOL, name
    net._use_fused = True
    net32, t32 = make(32, 1024, 1005)
    with torch.no_grad():
        out = net32.backbone(*t32, None)
    assert all(torch.isfinite(x).all() for x in out) and out[0].shape == (32, 3, 1024)


@pytest.mark.parametrize("b,n", [(3, 100), (1, 2500), (2, 6000)])
def test_odd_shapes_fused_matches_module_path(b, n):
    """Shapes off the tuned path: B not a power of two, N < 256, N > 2048 (generic FPS kernel + gather, ball query without the
    LDS-staged pair kernel beyond 5461 points)."""
    net, t = make(b, n, 1100 + n)
    with torch.no_grad():
        fused = net.backbone(*t, None)
        net._use_fused = False
        ref = net.backbone(*t, None)
    for name, a, r in zip(NAMES, fused, ref):
        assert rel_err(a.cpu(), r.cpu()) <= RTOL, name
