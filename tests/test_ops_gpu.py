"""GPU: every native op through the C ABI against the CPU oracle on seeded inputs -- bit-exact for
indices and for the float outputs of three_nn / three_interpolate / gather / group."""
import numpy as np
import pytest
import torch

from oracle import pointnet2_ref as P
from ratrack_amd import pointnet2_utils as PU
from ratrack_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def cloud(b, n, case_id):
    d = synth.make_frame_pairs(b, n, case_id)
    return torch.from_numpy(d["pc1"]).permute(0, 2, 1).contiguous(), torch.from_numpy(d["pc2"]).permute(0, 2, 1).contiguous()


@pytest.mark.parametrize("b,n,m", [(3, 256, 512), (2, 242, 512), (2, 512, 512), (2, 1024, 512), (1, 100, 37),
                                   (1, 2000, 300), (1, 2500, 64), (2, 1, 4), (1, 3, 8)])
def test_fps(b, n, m):
    xyz, _ = cloud(b, n, 10 + n)
    ref = P.fps(xyz, m)
    out = PU.furthest_point_sample(xyz.to(DEV), m).cpu()
    assert torch.equal(out, ref)


def test_fps_duplicates_and_ties():
    xyz, _ = cloud(1, 256, 5)
    xyz[0, 40:56] = xyz[0, 8]                 # exact duplicates -> exhausted-set behaviour
    xyz[0, 100] = xyz[0, 101]
    # symmetric points at exactly equal distance from point 0: tie rule decides
    xyz[0, 200] = xyz[0, 0] + torch.tensor([300., 0, 0])
    xyz[0, 201] = xyz[0, 0] - torch.tensor([300., 0, 0])
    for n in (256, 250):
        x = xyz[:, :n].contiguous()
        assert torch.equal(PU.furthest_point_sample(x.to(DEV), 512).cpu(), P.fps(x, 512))


@pytest.mark.parametrize("n,m", [(64, 64), (125, 125), (242, 300), (256, 512), (343, 343), (512, 512), (1000, 512), (1024, 512),
                                 (2048, 100), (2500, 40), (4096, 33)])
def test_fps_lattice_ties(n, m):
    """Integer lattices: almost every round has distinct points at exactly the same distance, so each pick is decided by
    the reference's block reduction (sampling_gpu.cu:86-91,143-203: smallest bit-reversed thread id wins).  The oracle is
    pinned to a literal emulation of that reduction in tests/test_emulator_cpu.py."""
    side = 2
    while side ** 3 < n:
        side += 1
    g = torch.stack(torch.meshgrid(*[torch.arange(float(side))] * 3, indexing="ij"), -1).reshape(-1, 3)
    perm = torch.randperm(side ** 3, generator=torch.Generator().manual_seed(n))
    x = torch.stack([g[:n], g[perm][:n]]).contiguous()          # lexicographic and shuffled lattice points
    ref = P.fps(x, m)
    out = PU.furthest_point_sample(x.to(DEV), m).cpu()
    assert torch.equal(out, ref)
    if n == 64:       # the probe quoted in VERDICT round 1: tids 1,2 tied -> tid 2
        assert ref[0, :8].tolist() == [0, 63, 56, 14, 35, 28, 49, 7]


@pytest.mark.parametrize("n,m,r,ns", [(256, 512, 2.0, 4), (256, 512, 4.0, 8), (512, 512, 8.0, 16), (512, 512, 16.0, 32),
                                      (1024, 512, 2.0, 4), (242, 512, 4.0, 8), (6000, 70, 8.0, 16), (5, 3, 100.0, 7)])
def test_ball_query(n, m, r, ns):
    xyz, q = cloud(2, max(n, m), 20 + n)
    xyz, q = xyz[:, :n].contiguous(), q[:, :m].contiguous()
    q[:, 0] = torch.tensor([500., 500., 500.])       # empty ball -> zero row
    ref = P.ball_query(r, ns, xyz, q)
    out = PU.ball_query(r, ns, xyz.to(DEV), q.to(DEV)).cpu()
    assert torch.equal(out, ref)
    assert out[:, 0].abs().sum() == 0


@pytest.mark.parametrize("n,m", [(512, 512), (256, 512), (1024, 512), (300, 2), (7, 6000)])
def test_three_nn(n, m):
    a, b = cloud(2, max(n, m), 30 + n)
    unknown, known = a[:, :n].contiguous(), b[:, :m].contiguous()
    known[:, : min(m, 4)] = unknown[:, : min(m, 4)] if n >= min(m, 4) else known[:, : min(m, 4)]   # zero distances / ties
    d2r, ir = P.three_nn(unknown, known)
    from ratrack_amd import pointnet2_hip
    d2 = torch.empty(2, n, 3, device=DEV)
    idx = torch.empty(2, n, 3, dtype=torch.int32, device=DEV)
    pointnet2_hip.three_nn_wrapper(2, n, m, unknown.to(DEV), known.to(DEV), d2, idx)
    assert torch.equal(idx.cpu(), ir)
    assert torch.equal(d2.cpu(), d2r)            # squared distances: bit-exact
    dist, idx2 = PU.three_nn(unknown.to(DEV), known.to(DEV))   # autograd front-end returns sqrt (device sqrt: <= 1 ulp)
    assert torch.equal(idx2.cpu(), ir)
    assert torch.allclose(dist.cpu(), torch.sqrt(d2r), rtol=2e-7, atol=0)


def test_gather_group_interpolate_and_grads():
    torch.manual_seed(0)
    B, C, N, S, ns = 2, 37, 300, 512, 8
    feats = torch.randn(B, C, N)
    idx1 = torch.randint(0, N, (B, S), dtype=torch.int32)
    idx3 = torch.randint(0, N, (B, S, ns), dtype=torch.int32)
    assert torch.equal(PU.gather_operation(feats.to(DEV), idx1.to(DEV)).cpu(), P.gather(feats, idx1))
    assert torch.equal(PU.grouping_operation(feats.to(DEV), idx3.to(DEV)).cpu(), P.group(feats, idx3))
    idn = torch.randint(0, N, (B, 200, 3), dtype=torch.int32)
    w = torch.rand(B, 200, 3)
    assert torch.equal(PU.three_interpolate(feats.to(DEV), idn.to(DEV), w.to(DEV)).cpu(), P.three_interpolate(feats, idn, w))

    # gradients: atomics make the summation order free -> tolerance, not bit-equality
    f = feats.to(DEV).requires_grad_(True)
    go = torch.randn(B, C, S, ns)
    PU.grouping_operation(f, idx3.to(DEV)).backward(go.to(DEV))
    ref = torch.zeros(B, C, N)
    P.group_points_grad_wrapper(B, C, N, S, ns, go, idx3, ref)
    assert torch.allclose(f.grad.cpu(), ref, rtol=1e-5, atol=1e-5)

    f = feats.to(DEV).requires_grad_(True)
    go = torch.randn(B, C, 200)
    PU.three_interpolate(f, idn.to(DEV), w.to(DEV)).backward(go.to(DEV))
    ref = torch.zeros(B, C, N)
    P.three_interpolate_grad_wrapper(B, C, 200, N, go, idn, w, ref)
    assert torch.allclose(f.grad.cpu(), ref, rtol=1e-5, atol=1e-5)

    f = feats.to(DEV).requires_grad_(True)
    go = torch.randn(B, C, S)
    PU.gather_operation(f, idx1.to(DEV)).backward(go.to(DEV))
    ref = torch.zeros(B, C, N)
    P.gather_points_grad_wrapper(B, C, N, S, go, idx1, ref)
    assert torch.allclose(f.grad.cpu(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("n,s,k", [(256, 256, 16), (1024, 1024, 16), (242, 242, 16), (50, 20, 3), (300, 10, 32), (20000, 5, 16),
                                   (256, 256, 64), (300, 7, 200), (77, 9, 77), (20000, 3, 40)])      # k > 32: the any-k selection kernel
def test_knn_point(n, s, k):
    a, b = cloud(2, max(n, s), 40 + n)
    xyz, q = a[:, :n].contiguous(), b[:, :s].contiguous()
    ref = P.knn_point(k, xyz, q)
    out = PU.knn_point(k, xyz.to(DEV), q.to(DEV)).cpu()
    assert out.dtype == torch.int64
    assert torch.equal(out, ref)       # same (distance, index) order as the oracle, hence the same set


def test_knn_export():
    a, b = cloud(2, 300, 50)
    unknown, known = a[:, :100].contiguous(), b[:, :300].contiguous()
    k = 5
    d2 = torch.empty(2, 100, k)
    idx = torch.empty(2, 100, k, dtype=torch.int32)
    P.knn_wrapper(2, 100, 300, k, unknown, known, d2, idx)
    dist, i2 = PU.knn(k, unknown.to(DEV), known.to(DEV))
    assert torch.equal(i2.cpu(), idx) and torch.allclose(dist.cpu(), torch.sqrt(d2), rtol=2e-7, atol=0)


def test_errors_are_reported_not_fatal():
    from ratrack_amd import _lib, pointnet2_hip
    x = torch.zeros(1, 4, 3, device=DEV)
    with pytest.raises(_lib.RtkError):
        pointnet2_hip.knn_point_wrapper(1, 4, 4, 64, x, x, torch.zeros(1, 4, 64, dtype=torch.int64, device=DEV))
