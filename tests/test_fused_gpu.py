"""GPU: the fused stages (include/rtk_fused.h) against fp64 references of the same op on seeded data."""
import numpy as np
import pytest
import torch

from ratrack_amd import fused as F
from ratrack_amd import pointnet2_utils as PU

from _util import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def ref_act(x, act):
    if act == F.ACT_RELU:
        return torch.relu(x)
    if act == F.ACT_LEAKY:
        return torch.nn.functional.leaky_relu(x, 0.1)
    if act == F.ACT_SIGMOID:
        return torch.sigmoid(x)
    return x


@pytest.mark.parametrize("rows,n,cin,widths,acts", [
    (2 * 256, 256, 64, [96], [F.ACT_NONE]),
    (3 * 242, 242, 96, [192], [F.ACT_NONE]),          # partial last tile, tiles straddling samples
    (2 * 512, 512, 128, [64], [F.ACT_RELU]),
    (5 * 256, 256, 128, [256], [F.ACT_NONE]),
    (2 * 256, 256, 256, [128, 64, 32, 1], [F.ACT_RELU, F.ACT_RELU, F.ACT_RELU, F.ACT_SIGMOID]),
    (70 * 256, 256, 128, [128, 64, 32, 3], [F.ACT_RELU, F.ACT_RELU, F.ACT_RELU, F.ACT_NONE]),   # > one pass of the grid
])
@pytest.mark.parametrize("split", [True, False])      # split images (the default) / fp32-input MFMA images
def test_pointwise_chain(rows, n, cin, widths, acts, split, monkeypatch):
    monkeypatch.setattr(F, "PW_SPLIT", split)
    torch.manual_seed(rows + cin)
    x = torch.randn(rows, cin, dtype=torch.float64)
    layers, cur = [], cin
    for w_, a in zip(widths, acts):
        layers.append((torch.randn(w_, cur, dtype=torch.float64) / cur ** 0.5, torch.randn(w_, dtype=torch.float64) * 0.1, a))
        cur = w_
    sb = torch.randn(rows // n, F.ceil16(widths[0]), dtype=torch.float64) * 0.3
    ref = x
    for i, (w, b, a) in enumerate(layers):
        ref = ref @ w.T + b
        if i == 0:
            ref = ref + sb[:, :widths[0]].repeat_interleave(n, 0)
        ref = ref_act(ref, a)
    chain = F.Chain(layers, DEV)
    out = torch.full((rows, F.ceil16(widths[-1]) + 4), -7.0, device=DEV)
    F.pointwise_mlp(rows, n, [(x.float().to(DEV).contiguous(), cin, False)], chain, out, sample_bias=sb.float().to(DEV).contiguous())
    got = out[:, :widths[-1]].double().cpu()
    assert rel_err(got, ref) < 2e-6
    assert (out[:, widths[-1]:] == -7.0).all()          # nothing written beyond out_channels
    # channel-major output variant
    out_cm = torch.empty(rows // n, widths[-1], n, device=DEV)
    F.pointwise_mlp(rows, n, [(x.float().to(DEV).contiguous(), cin, False)], chain, out_cm, sample_bias=sb.float().to(DEV).contiguous(),
                    channel_major=True)
    assert rel_err(out_cm.permute(0, 2, 1).reshape(rows, -1).double().cpu(), ref) < 2e-6


@pytest.mark.parametrize("split", [True, False])
def test_pointwise_multi_source_and_broadcast(split, monkeypatch):
    monkeypatch.setattr(F, "PW_SPLIT", split)
    """[2 raw channels (padded slot) || 128 local || per-sample 256] -> 32, as the decoder's embedding projection."""
    torch.manual_seed(1)
    B, n = 3, 256
    raw = torch.randn(B * n, 4, dtype=torch.float64)
    raw[:, 2:] = 0
    loc = torch.randn(B * n, 128, dtype=torch.float64)
    cor = torch.randn(B * n, 256, dtype=torch.float64)
    w = torch.randn(32, 16 + 128 + 256, dtype=torch.float64) * 0.05
    b = torch.randn(32, dtype=torch.float64)
    xin = torch.cat([torch.nn.functional.pad(raw, (0, 12)), loc, cor], 1)
    ref = xin @ w.T + b
    chain = F.Chain([(w, b, F.ACT_NONE)], DEV)
    out = torch.empty(B * n, 32, device=DEV)
    f = lambda t: t.float().to(DEV).contiguous()
    F.pointwise_mlp(B * n, n, [(f(raw), 2, False), (f(loc), 128, False), (f(cor), 256, False)], chain, out)
    assert rel_err(out.double().cpu(), ref) < 2e-6
    # per-sample broadcast source
    glob = torch.randn(B, 128, dtype=torch.float64)
    ws = [torch.randn(128, 256, dtype=torch.float64) * 0.05, torch.randn(64, 128, dtype=torch.float64) * 0.1,
          torch.randn(32, 64, dtype=torch.float64) * 0.1, torch.randn(3, 32, dtype=torch.float64) * 0.1]
    ref2 = torch.cat([loc, glob.repeat_interleave(n, 0)], 1)
    for i, w_ in enumerate(ws):
        ref2 = ref2 @ w_.T
        if i < 3:
            ref2 = torch.relu(ref2)
    chain2 = F.Chain([(w_, torch.zeros(w_.shape[0], dtype=torch.float64), F.ACT_RELU if i < 3 else F.ACT_NONE)
                      for i, w_ in enumerate(ws)], DEV)
    out2 = torch.empty(B * n, 4, device=DEV)
    F.pointwise_mlp(B * n, n, [(f(loc), 128, False), (f(glob), 128, True)], chain2, out2, out_channels=3)
    assert rel_err(out2[:, :3].double().cpu(), ref2) < 2e-6


@pytest.mark.parametrize("n_unknown,m,cint,cskip", [(512, 512, 64, 64), (512, 512, 128, 32), (256, 512, 128, 0), (242, 512, 128, 0)])
@pytest.mark.parametrize("split", [True, False])
def test_pointwise_interp_prologue(n_unknown, m, cint, cskip, split, monkeypatch):
    monkeypatch.setattr(F, "PW_SPLIT", split)
    """FP module: three_nn weights + three_interpolate + cat(skip) + conv/BN/ReLU in one kernel."""
    torch.manual_seed(n_unknown + cint)
    B = 2
    unknown = torch.randn(B, n_unknown, 3) * 5
    known = torch.randn(B, m, 3) * 5
    known[:, :50] = unknown[:, :50]                 # exact matches: w0 == 1 after normalisation
    kf = torch.randn(B, m, cint)
    skip = torch.randn(B, n_unknown, cskip) if cskip else None
    w = torch.randn(128, cint + cskip, dtype=torch.float64) * 0.1
    b = torch.randn(128, dtype=torch.float64) * 0.1
    from ratrack_amd import pointnet2_hip
    d2 = torch.empty(B, n_unknown, 3, device=DEV)
    idx = torch.empty(B, n_unknown, 3, dtype=torch.int32, device=DEV)
    pointnet2_hip.three_nn_wrapper(B, n_unknown, m, unknown.to(DEV), known.to(DEV), d2, idx)
    # reference in fp64 with the module path's formula
    dist = torch.sqrt(d2.cpu().double())
    r = 1.0 / (dist + 1e-8)
    wgt = r / r.sum(2, keepdim=True)
    gathered = torch.stack([kf[bb][idx[bb].cpu().long()] for bb in range(B)]).double()     # (B,n,3,C)
    interp = (gathered * wgt[..., None]).sum(2)
    xin = torch.cat([interp, skip.double()], 2) if cskip else interp
    ref = torch.relu(xin.reshape(B * n_unknown, -1) @ w.T + b)
    chain = F.Chain([(w, b, F.ACT_RELU)], DEV)
    out = torch.empty(B * n_unknown, 128, device=DEV)
    srcs = [(skip.reshape(B * n_unknown, cskip).to(DEV).contiguous(), cskip, False)] if cskip else []
    F.pointwise_mlp(B * n_unknown, n_unknown, srcs, chain, out,
                    interp=(kf.reshape(B * m, cint).to(DEV).contiguous(), cint, m, idx.reshape(-1, 3), d2.reshape(-1, 3)))
    assert rel_err(out.double().cpu(), ref) < 5e-6


# ---- stage kernels against the module path (HIP ops + PyTorch dense layers) on the golden inputs --------------
def _net():
    from ratrack_amd.track4d import Args, Track4D
    from _util import reference_state_dict
    net = Track4D(Args()).to(DEV).eval()
    net.load_state_dict(reference_state_dict(DEV), strict=True)
    return net


@pytest.mark.parametrize("name", ["eval_b2_n256", "eval_b1_n242", "eval_b1_n1024", "eval_b1_n256_dups"])
def test_fused_pnhead_matches_modules(name):
    from _util import inputs_of, load_case
    case = load_case(name)
    net = _net()
    pc1, pc2, f1, f2 = inputs_of(case, DEV)
    B, _, N = pc1.shape
    with torch.no_grad():
        _, ref = net.pn_head(pc1.permute(0, 2, 1).contiguous(), f1)          # (B,128,N)
        eng = F.FusedBackbone(net)
        xyz = pc1.permute(0, 2, 1).contiguous()
        raw = torch.zeros(B, N, 4, device=DEV)
        raw[:, :, :2] = f1.permute(0, 2, 1)
        raw = raw.reshape(B * N, 4)
        geo = F.Geometry(xyz, 512)
        torch.cuda.synchronize()
        q1 = F.pointwise(B * N, N, [(raw, 2, False)], eng.enc_q1, torch.empty(B * N, 32, device=DEV))
        loc, gmax = F.run_pnhead(eng.enc, geo, q1)
        got = loc.view(B, N, 128).permute(0, 2, 1)
    assert rel_err(got.cpu(), ref.cpu()) < 2e-5
    assert torch.equal(gmax, loc.view(B, N, 128).amax(1))          # fused global max-pool == max over points


@pytest.mark.parametrize("split", [True, False])      # split-bf16 matrix path (the default) / fp32-input MFMA kernel
@pytest.mark.parametrize("name", ["eval_b2_n256", "eval_b1_n242", "eval_b1_n1024"])
def test_fused_cost_volume_matches_modules(name, split):
    from _util import inputs_of, load_case
    case = load_case(name)
    net = _net()
    pc1, pc2, f1, f2 = inputs_of(case, DEV)
    B, _, N = pc1.shape
    torch.manual_seed(0)
    feat1 = torch.randn(B, 256, N, device=DEV)
    feat2 = torch.randn(B, 256, N, device=DEV)
    feat1[:, 128:] = feat1[:, 128:, :1]      # global halves are constant over points, as in the model
    feat2[:, 128:] = feat2[:, 128:, :1]
    with torch.no_grad():
        ref = net.fc_layer(pc1, pc2, feat1, feat2)                             # (B,256,N)
        eng = F.FusedBackbone(net)
        new = lambda r, c: torch.empty(r, c, device=DEV)
        pm = lambda t: t.permute(0, 2, 1).reshape(B * N, -1).contiguous()
        l1, l2 = pm(feat1[:, :128]), pm(feat2[:, :128])
        g1, g2 = feat1[:, 128:, 0].contiguous(), feat2[:, 128:, 0].contiguous()
        sb1, sb2 = new(B, 256), new(B, 256)
        F.global_terms(torch.cat([g1, g2]), [(eng.p1_glob_wt, eng.p1_glob_b, sb1, 0), (eng.p2_glob_wt, None, sb2, B)])
        p1 = F.pointwise(B * N, N, [(l1, 128, False)], eng.p1_loc, new(B * N, 256), sample_bias=sb1)
        p2 = F.pointwise(B * N, N, [(l2, 128, False)], eng.p2_loc, new(B * N, 256), sample_bias=sb2)
        x1 = pc1.permute(0, 2, 1).contiguous()
        x2 = pc2.permute(0, 2, 1).contiguous()
        k1 = PU.knn_point(16, x2, x1)
        k2 = PU.knn_point(16, x1, x1)
        from ratrack_amd import _lib
        cor1 = torch.full((B * N, 256), float("nan"), device=DEV)
        eng.cv_split = split
        eng._cost_volume(B, N, x1, x2, k1, p1, p2, cor1)
        cor = new(B * N, 256)
        _lib.call("rtk_patch_cost", B, N, x1.data_ptr(), k2.data_ptr(), cor1.data_ptr(), 256, eng.wn2.arr, cor.data_ptr(), 256, 0, F._stream())
        got = cor.view(B, N, 256).permute(0, 2, 1)
    assert rel_err(got.cpu(), ref.cpu()) < 2e-5


def _pack2(ws):
    """host restatement of the device packer for layers back to back: (images, inverse scales)"""
    packed = [F.pack_layer_split(w) for w in ws]
    return torch.cat([im for im, _ in packed]), torch.tensor([inv for _, inv in packed], dtype=torch.float32, device=ws[0].device)


def _mlp2(positions, x, img, scales, b, y):
    from ratrack_amd import _lib
    _lib.call("rtk_split_mlp2", positions, x.data_ptr(), img.data_ptr(), scales.data_ptr(), b[0].data_ptr(), b[1].data_ptr(), y.data_ptr(),
              F._stream())


def test_split_images_are_two_fp16_pieces_and_packed_as_documented():
    """csrc/split_mfma.h: with the packer's power-of-two scale the two fp16 pieces carry every fp32 weight within 2^-16 of the matrix's
    largest to 2^-23 of itself (smaller ones to an absolute 2^-39 of the largest), and the device packer writes the image and the inverse scale the
    host restatement (fused.pack_layer_split) describes -- bit for bit."""
    from ratrack_amd import _lib
    torch.manual_seed(3)
    w = torch.randn(256, 256, device=DEV) * torch.logspace(-4, 0, 256, device=DEV)[:, None]      # 4 decades of magnitudes
    w[0, :3] = torch.tensor([0.0, 1.0, -1.5], device=DEV)
    scale, inv = F.pow2_scale(w.abs().max())
    assert 2.0 ** 14 <= float(w.abs().max()) * scale < 2.0 ** 15 and scale * inv == 1.0
    pieces = F.split2_f16(w, scale).view(torch.float16)                                         # (2, 256, 256)
    back = (pieces[0].double() + pieces[1].double()) * inv
    err = (back - w.double()).abs() * scale
    ws = w.double().abs() * scale
    assert bool((err <= 2.0 ** -25 + 2.0 ** -23 * ws).all())
    assert bool((err[ws >= 0.25] <= 2.0 ** -23 * ws[ws >= 0.25]).all())
    img, dinv = F.pack_split_device(w)
    himg, hinv = F.pack_layer_split(w)
    assert torch.equal(img, himg) and float(dinv) == hinv
    imgt, dinvt = F.pack_split_device(w, transposed=True)                                       # the backward's W^T image from the same weights
    himgt, hinvt = F.pack_layer_split(w.T.contiguous())
    assert torch.equal(imgt, himgt) and float(dinvt) == hinvt
    with pytest.raises(_lib.RtkError):
        _lib.call("rtk_pack_split_layer", 250, 256, w.data_ptr(), 0, img.data_ptr(), dinv.data_ptr(), F._stream())


@pytest.mark.parametrize("positions", [1, 77, 128, 5000])
def test_split_layers_carry_fp32_accuracy(positions):
    """Two 256x256 layers on the fp16 matrix pipe (three products of two pieces, a power-of-two scale per matrix and per position)
    against float64: the error is that of an fp32 GEMM -- not of an fp16 one (1e-3) nor of a three-product bf16 split (3e-5)."""
    torch.manual_seed(positions)
    W = [torch.randn(256, 256, device=DEV) / 16 for _ in range(2)]
    b = [torch.randn(256, device=DEV) * 0.1 for _ in range(2)]
    img, scales = _pack2(W)
    x = torch.randn(positions, 256, device=DEV) * 3
    y = torch.full((positions, 256), float("nan"), device=DEV)
    _mlp2(positions, x, img, scales, b, y)
    lk = lambda t: torch.nn.functional.leaky_relu(t, 0.1)
    r64 = lk(lk(x.double() @ W[0].double().T + b[0].double()) @ W[1].double().T + b[1].double())
    r32 = lk(lk(x @ W[0].T + b[0]) @ W[1].T + b[1])
    err = lambda t: float((t.double() - r64).abs().max() / r64.abs().max())
    assert err(y) < 2e-6 and err(y) < 3 * err(r32) + 1e-7, (err(y), err(r32))


@pytest.mark.parametrize("log2_scale", [-100, -60, -20, 0, 20, 60, 100])
def test_split_layers_are_scale_invariant(log2_scale):
    """The path has no operand range: activations scaled by 2^k (and the first layer's weights by 2^-k', the biases accordingly) give
    results that are the unscaled ones times the same power of two BIT FOR BIT -- every scale is a power of two chosen from the data,
    so the fp16 pieces are the same numbers.  k = +-100 puts the activations at 1e+-30, far outside fp16's (and bf16-exact fp32's) range."""
    torch.manual_seed(5)
    P = 384
    W = [torch.randn(256, 256, device=DEV) / 16 for _ in range(2)]
    b = [torch.randn(256, device=DEV) * 0.1 for _ in range(2)]
    x = torch.randn(P, 256, device=DEV) * 3
    x[7] = 0.0                                                                                  # an all-zero position
    img, scales = _pack2(W)
    y0 = torch.empty(P, 256, device=DEV)
    _mlp2(P, x, img, scales, b, y0)
    k = float(2.0 ** log2_scale)
    # x k, W1 unchanged, b1 k: layer 1's output is k times the old one; W2 / k: layer 2's input term is back at the old magnitude
    W2 = [W[0], W[1] / k]
    img2, scales2 = _pack2(W2)
    y1 = torch.empty(P, 256, device=DEV)
    _mlp2(P, x * k, img2, scales2, [b[0] * k, b[1]], y1)
    assert torch.isfinite(y1).all()
    assert torch.equal(y0, y1)
    assert torch.equal(y0[7], torch.nn.functional.leaky_relu(torch.nn.functional.leaky_relu(b[0], 0.1) @ W[1].T + b[1], 0.1)) or \
        rel_err(y0[7].cpu(), torch.nn.functional.leaky_relu(torch.nn.functional.leaky_relu(b[0], 0.1) @ W[1].T + b[1], 0.1).cpu()) < 1e-6


@pytest.mark.parametrize("case", ["same_sign", "scale_1e6", "scale_1e-6", "mixed_range", "tiny_1e-30", "post_relu"])
def test_split_layers_adversarial_operands(case):
    """Operands chosen against the split: all-positive activations x all-positive weights, as after a ReLU (a dropped l.l term or a
    biased rounding would add up coherently), dynamic ranges of 1e+-6 between activations and weights, activations spanning six
    decades inside one row (the small ones lose their low piece to the position's scale: an ABSOLUTE error 2^-39 of the row's
    largest), and activations at 1e-30 -- fifteen orders below fp16's smallest subnormal; float64 is the truth, the framework's fp32
    GEMM the yardstick."""
    torch.manual_seed(7)
    P = 640
    W = [torch.randn(256, 256, device=DEV) / 16 for _ in range(2)]
    b = [torch.randn(256, device=DEV) * 0.1 for _ in range(2)]
    x = torch.randn(P, 256, device=DEV) * 3
    if case == "same_sign":
        x, W = x.abs(), [w.abs() for w in W]
        b = [v.abs() for v in b]
    elif case == "scale_1e6":
        x, W[0] = x * 1e6, W[0] * 1e-6
    elif case == "scale_1e-6":
        x, W[0] = x * 1e-6, W[0] * 1e6
    elif case == "mixed_range":
        x = x * torch.pow(10.0, torch.randint(-3, 4, (P, 256), device=DEV).float())
    elif case == "tiny_1e-30":
        x, W[0], b[0] = x * 1e-30, W[0], b[0] * 0      # low pieces ~1e-33: fp32-normal only just, far below anything fp16 holds unscaled
        W[1], b[1] = W[1] * 1e10, b[1] * 0
    elif case == "post_relu":
        x = torch.relu(x)                                 # half zeros, half positive
        W = [w.abs() for w in W]
    img, scales = _pack2(W)
    y = torch.full((P, 256), float("nan"), device=DEV)
    _mlp2(P, x, img, scales, b, y)
    lk = lambda t: torch.nn.functional.leaky_relu(t, 0.1)
    r64 = lk(lk(x.double() @ W[0].double().T + b[0].double()) @ W[1].double().T + b[1].double())
    r32 = lk(lk(x @ W[0].T + b[0]) @ W[1].T + b[1])
    # per-element relative error where the result is not a cancellation (|r| >= 1e-3 of the row's largest), and error to scale
    big = r64.abs() >= 1e-3 * r64.abs().amax(1, keepdim=True)
    rel = lambda t: float(((t.double() - r64).abs() / r64.abs())[big].max())
    sc = lambda t: float((t.double() - r64).abs().max() / r64.abs().max())
    bias = lambda t: float(((t.double() - r64) / r64.abs())[big].mean())           # signed: a coherent truncation bias shows here
    print("\n%s: split  rel %.2e scale %.2e mean signed %.2e | fp32 GEMM rel %.2e scale %.2e mean signed %.2e" % (
        case, rel(y), sc(y), bias(y), rel(r32), sc(r32), bias(r32)))
    assert torch.isfinite(y).all()
    assert sc(y) < 2e-6 and sc(y) < 3 * sc(r32) + 2e-7, (case, sc(y), sc(r32))
    assert abs(bias(y)) < 2.5e-7, (case, bias(y))                                   # round-to-nearest pieces: no coherent bias


@pytest.mark.parametrize("B,N", [(3, 243), (8, 250), (16, 64), (1, 1024), (2, 17)])
def test_cost_volume_split_agrees_with_fp32_mfma_kernel(B, N):
    """Both kernels on the same operands: samples not a multiple of 8 (plain 2-D grid) and a multiple (XCD-aware grid), point counts
    that leave the last workgroup iteration partly empty, one tile per workgroup (16 x 64: no next tile to request rows for) and
    sixteen (1 x 1024), barely more points than neighbours (17), repeated neighbours (frame 2 holds every point twice), and rows
    beyond the last point untouched.  (The split kernel stages the gathered rows through LDS and broadcasts the p1 row across a
    point's neighbours: every lane / slot / swizzle mistake shows here.)"""
    net = _net()
    eng = F.FusedBackbone(net)
    torch.manual_seed(11 + B + N)
    x1, x2 = torch.randn(B, N, 3, device=DEV), torch.randn(B, N, 3, device=DEV)
    x2[:, N // 2:] = x2[:, :N - N // 2]
    p1, p2 = torch.randn(B * N, 256, device=DEV), torch.randn(B * N, 256, device=DEV)
    k1 = PU.knn_point(16, x2, x1)
    out = []
    for split in (False, True):
        eng.cv_split = split
        o = torch.full((B * N + 4, 256), 7.0, device=DEV)
        eng._cost_volume(B, N, x1, x2, k1, p1, p2, o)
        assert torch.all(o[B * N:] == 7.0)
        out.append(o[:B * N])
    assert rel_err(out[1].cpu(), out[0].cpu()) < 5e-6


@pytest.mark.parametrize("name", ["eval_b2_n256", "eval_b1_n242", "eval_b1_n1024", "eval_b1_n256_dups"])
def test_fused_backbone_matches_golden(name):
    """The product's default eval path (fused kernels) against the reference-graph golden vectors."""
    from _util import RTOL, assert_close, inputs_of, load_case
    case = load_case(name)
    net = _net()
    assert net._use_fused
    pc1, pc2, f1, f2 = inputs_of(case, DEV)
    with torch.no_grad():
        flow, h, cls, cor, pf1, pf2, prop = net.backbone(pc1, pc2, f1, f2, None)
        assert net._fused, "fused engine was not used"
        flow2, h2, *_ = net.backbone(pc1, pc2, f1, f2, h)
    cpu = lambda t: t.float().cpu().numpy()
    assert flow.shape == case["flow"].shape and cls.shape == case["cls"].shape
    assert_close(cpu(flow), case["flow"], RTOL, "flow")
    assert_close(cpu(cls), case["cls"], RTOL, "cls")
    assert_close(cpu(h), case["h_out"], RTOL, "h")
    assert_close(cpu(cor[:, :, ::8]), case["cor_s8"], RTOL, "cor")
    assert_close(cpu(prop[:, :, ::8]), case["prop_s8"], RTOL, "prop")
    assert_close(cpu(pf1[:, :, ::8]), case["pc1_features_s8"], RTOL, "pc1_features")
    assert_close(cpu(pf2[:, :, ::8]), case["pc2_features_s8"], RTOL, "pc2_features")
    assert_close(cpu(flow2), case["flow_step2"], RTOL, "flow step 2")
    assert_close(cpu(h2), case["h_out_step2"], RTOL, "h step 2")
    gt = torch.from_numpy(case["in_gt_warp"]).to(DEV)
    epe = float(torch.sqrt(((pc1[:1] + flow[:1] - gt[:1]) ** 2).sum(1) + 1e-20).mean())
    ref = float(case["metric_sf_vals"][list(case["metric_sf_keys"]).index("epe")])
    assert abs(epe - ref) <= 1e-4 * max(ref, 1.0), (epe, ref)


def test_misc_kernels():
    """prepare_inputs, fps_centroids (+ exhausted-cloud counter), gru_step, to_channel_major."""
    from oracle import pointnet2_ref as P
    from ratrack_amd import _lib, synth
    torch.manual_seed(3)
    B, N = 3, 242
    d = synth.make_frame_pairs(B, N, 11)
    t = {k: torch.from_numpy(v).to(DEV) for k, v in d.items() if k != "gt_cls"}
    xyz = torch.empty(2 * B, N, 3, device=DEV)
    raw = torch.empty(2 * B * N, 4, device=DEV)
    _lib.call("rtk_prepare_inputs", B, N, t["pc1"].data_ptr(), t["pc2"].data_ptr(), t["feature1"].data_ptr(), t["feature2"].data_ptr(),
              xyz.data_ptr(), raw.data_ptr(), F._stream())
    assert torch.equal(xyz, torch.cat([t["pc1"], t["pc2"]]).permute(0, 2, 1))
    assert torch.equal(raw.view(2 * B, N, 4)[:, :, :2], torch.cat([t["feature1"], t["feature2"]]).permute(0, 2, 1))
    assert (raw.view(2 * B, N, 4)[:, :, 2:] == 0).all()
    # fps + gather + counter
    idx = torch.empty(2 * B, 512, dtype=torch.int32, device=DEV)
    nxyz = torch.empty(2 * B, 512, 3, device=DEV)
    cnt = torch.empty(2 * B, dtype=torch.int32, device=DEV)
    tie = torch.empty(2 * B, dtype=torch.int32, device=DEV)
    _lib.call("rtk_fps_centroids", 2 * B, N, 512, xyz.data_ptr(), idx.data_ptr(), nxyz.data_ptr(), cnt.data_ptr(), tie.data_ptr(), None, None, None,
              F._stream())
    assert (tie.cpu() == 0).all()              # generic float cloud: no round had two points at the maximum
    ref = P.fps(xyz.cpu(), 512)
    assert torch.equal(idx.cpu(), ref)
    assert torch.equal(nxyz.cpu(), torch.gather(xyz.cpu(), 1, ref.long().unsqueeze(-1).expand(-1, -1, 3)))
    assert (cnt.cpu() == N).all()              # N distinct points, then exhausted
    # GRU step vs torch.nn.GRU (CPU)
    gru = torch.nn.GRU(128, 128, 5)
    x, h = torch.randn(4, 128), torch.randn(5, 4, 128)
    with torch.no_grad():
        y_ref, h_ref = gru(x.unsqueeze(0), h)
    st = lambda k: torch.stack([getattr(gru, "%s_l%d" % (k, l)).detach() for l in range(5)])
    wih, whh = st("weight_ih").transpose(1, 2).contiguous().to(DEV), st("weight_hh").transpose(1, 2).contiguous().to(DEV)
    bih, bhh = st("bias_ih").contiguous().to(DEV), st("bias_hh").contiguous().to(DEV)
    h_out, y = torch.empty(5, 4, 128, device=DEV), torch.empty(4, 128, device=DEV)
    xd, hd = x.to(DEV), h.to(DEV)        # keep the device copies alive across the asynchronous launch
    _lib.call("rtk_gru_step", 4, 5, 128, xd.data_ptr(), hd.data_ptr(), wih.data_ptr(), whh.data_ptr(), bih.data_ptr(),
              bhh.data_ptr(), h_out.data_ptr(), y.data_ptr(), F._stream())
    e1, e2 = rel_err(h_out.cpu(), h_ref), rel_err(y.cpu(), y_ref[0])
    assert e1 < 1e-5 and e2 < 1e-5, (e1, e2)
    # layout
    src = torch.randn(B * N, 132, device=DEV)
    dst = torch.full((B, 200, N), -1.0, device=DEV)
    _lib.call("rtk_to_channel_major", B, N, 100, src.data_ptr(), 132, 0, dst.data_ptr(), 200, 50, F._stream())
    assert torch.equal(dst[:, 50:150], src[:, :100].view(B, N, 100).permute(0, 2, 1))
    assert (dst[:, :50] == -1).all() and (dst[:, 150:] == -1).all()
    g = torch.randn(B, 128, device=DEV)
    _lib.call("rtk_to_channel_major", B, N, 128, g.data_ptr(), 128, 1, dst.data_ptr(), 200, 0, F._stream())
    assert torch.equal(dst[:, :128], g.unsqueeze(2).expand(-1, -1, N))


def test_fps_relevel():
    """Levels 2/3 (FPS of 512 out of the previous level's 512 centroids): rtk_fps_relevel -- per cloud a copy when the
    previous level had no tie, the full selection otherwise -- against (a) the full selection kernel level after level and
    (b) the CPU oracle, on every fixture cloud, on duplicate-heavy clouds and on lattices with exact distance ties
    between distinct points (where the reference's bit-reversed tie rule makes level 2 differ from level 1)."""
    from _util import EVAL_CASES, inputs_of, load_case
    from oracle import pointnet2_ref as P
    from ratrack_amd import _lib
    clouds = []
    for name in EVAL_CASES:
        pc1, pc2, _, _ = inputs_of(load_case(name), DEV)
        clouds += [pc1.permute(0, 2, 1).contiguous(), pc2.permute(0, 2, 1).contiguous()]
    g = torch.Generator().manual_seed(5)
    lattice = torch.randint(0, 6, (3, 700, 3), generator=g).float().to(DEV)      # integer lattice: many exact ties + duplicates
    grid = torch.stack(torch.meshgrid(*[torch.arange(8.)] * 3, indexing="ij"), -1).reshape(1, 512, 3).to(DEV)
    clouds += [lattice, lattice[:, :300].contiguous(), torch.zeros(2, 64, 3, device=DEV), grid, grid[:, :256].contiguous()]
    n_tied = n_moved = 0
    for xyz in clouds:
        S_, n, _ = xyz.shape
        l1 = torch.empty(S_, 512, 3, device=DEV)
        idx = torch.empty(S_, 512, dtype=torch.int32, device=DEV)
        c1 = torch.empty(S_, dtype=torch.int32, device=DEV)
        tie = torch.empty(S_, dtype=torch.int32, device=DEV)
        snap = torch.full((S_, n), float("nan"), device=DEV)
        first = torch.zeros(S_, dtype=torch.int32, device=DEV)
        _lib.call("rtk_fps_centroids", S_, n, 512, xyz.data_ptr(), idx.data_ptr(), l1.data_ptr(), c1.data_ptr(), tie.data_ptr(), None,
                  snap.data_ptr(), first.data_ptr(), F._stream())
        assert ((first > 0) == (tie > 0)).all() and (first <= tie).all()
        for resume in (True, False):          # tied clouds resume at level 1's first tied round / start over at round 1
            idx23 = torch.empty(2, S_, 512, dtype=torch.int32, device=DEV)
            xyz23 = torch.empty(2, S_, 512, 3, device=DEV)
            c23 = torch.empty(2, S_, dtype=torch.int32, device=DEV)
            tie23 = torch.empty(2, S_, dtype=torch.int32, device=DEV)
            extra = (idx.data_ptr(), snap.data_ptr(), n, first.data_ptr()) if resume else (None, None, 0, None)
            _lib.call("rtk_fps_relevel", S_, 512, 2, l1.data_ptr(), c1.data_ptr(), tie.data_ptr(), idx23.data_ptr(), xyz23.data_ptr(),
                      c23.data_ptr(), tie23.data_ptr(), *extra, F._stream())
            F.check_fps_relevel(l1, idx23, xyz23, c23)
            assert (tie23[0][tie == 0] == 0).all()      # an untied level stays untied
        # the CPU oracle, level after level
        src = l1.cpu()
        assert torch.equal(idx.cpu(), P.fps(xyz.cpu(), 512))
        for l in range(2):
            ref = P.fps(src, 512)
            assert torch.equal(idx23[l].cpu(), ref), l
            src = torch.gather(src, 1, ref.long().unsqueeze(-1).expand(-1, -1, 3))
            assert torch.equal(xyz23[l].cpu(), src), l
        n_tied += int(tie.sum())
        n_moved += int((xyz23[0] != l1).any(-1).any(-1).sum())
    assert n_tied > 0 and n_moved > 0, "the lattice clouds must exercise the tied (non-identity) path"


def test_ball_query_pair_and_masked_three_nn():
    from oracle import pointnet2_ref as P
    from ratrack_amd import _lib, synth
    d = synth.make_frame_pairs(3, 300, 21)
    xyz = torch.from_numpy(d["pc1"]).permute(0, 2, 1).contiguous()
    q = torch.from_numpy(d["pc2"]).permute(0, 2, 1).contiguous()[:, :200].contiguous()
    q[:, 0] = 900.0                                           # empty balls
    nuniq = torch.tensor([200, 150, 7], dtype=torch.int32)
    for (r1, n1, r2, n2) in [(2.0, 4, 4.0, 8), (4.0, 8, 8.0, 16), (8.0, 16, 16.0, 32)]:
        i1 = torch.zeros(3, 200, n1, dtype=torch.int32, device=DEV)
        i2 = torch.zeros(3, 200, n2, dtype=torch.int32, device=DEV)
        xd, qd, nd = xyz.to(DEV), q.to(DEV), nuniq.to(DEV)
        _lib.call("rtk_ball_query_pair", 3, 300, 200, r1, n1, r2, n2, qd.data_ptr(), xd.data_ptr(), i1.data_ptr(), i2.data_ptr(),
                  nd.data_ptr(), F._stream())
        ra, rb = P.ball_query(r1, n1, xyz, q), P.ball_query(r2, n2, xyz, q)
        for b in range(3):
            e = int(nuniq[b])
            assert torch.equal(i1[b, :e].cpu(), ra[b, :e]) and torch.equal(i2[b, :e].cpu(), rb[b, :e])
            assert (i1[b, e:] == 0).all() and (i2[b, e:] == 0).all()         # skipped rows keep the caller's zeros
    d2 = torch.full((3, 200, 3), -1.0, device=DEV)
    idx = torch.full((3, 200, 3), -1, dtype=torch.int32, device=DEV)
    _lib.call("rtk_three_nn_masked", 3, 200, 300, qd.data_ptr(), xd.data_ptr(), d2.data_ptr(), idx.data_ptr(), nd.data_ptr(), None,
              F._stream())
    rd, ri = P.three_nn(q, xyz)
    for b in range(3):
        e = int(nuniq[b])
        assert torch.equal(idx[b, :e].cpu(), ri[b, :e]) and torch.equal(d2[b, :e].cpu(), rd[b, :e])
        lim = (e + 63) // 64 * 64            # a workgroup takes 64 queries (4 lanes each; 16 x 16 lanes until round 6)
        assert (idx[b, lim:] == -1).all()


def test_three_nn_duplicate_aware_scan_is_exact():
    """Known clouds whose tail rows are copies of row 0 (over-sampled FPS levels): scanning only the unique prefix plus the two
    lowest duplicate indices reproduces the full scan bit for bit, including the tie order (0, E, E+1)."""
    from oracle import pointnet2_ref as P
    from ratrack_amd import _lib
    torch.manual_seed(9)
    B, m, n = 3, 512, 300
    known = torch.randn(B, m, 3) * 4
    ke = torch.tensor([256, 511, 40], dtype=torch.int32)
    for b in range(B):
        known[b, int(ke[b]):] = known[b, 0]
    unknown = torch.randn(B, n, 3) * 4
    unknown[:, :64] = known[:, :64]                # zero distances incl. to row 0 and its copies
    unknown[:, 64] = known[:, 0] + 1e-3
    rd, ri = P.three_nn(unknown, known)
    d2 = torch.empty(B, n, 3, device=DEV)
    idx = torch.empty(B, n, 3, dtype=torch.int32, device=DEV)
    ud, kd, ked = unknown.to(DEV), known.to(DEV), ke.to(DEV)
    _lib.call("rtk_three_nn_masked", B, n, m, ud.data_ptr(), kd.data_ptr(), d2.data_ptr(), idx.data_ptr(), None, ked.data_ptr(), F._stream())
    assert torch.equal(idx.cpu(), ri) and torch.equal(d2.cpu(), rd)


@pytest.mark.parametrize("m,n", [(1, 1), (3, 5), (20, 17), (1, 30), (64, 64), (100, 120)])
def test_log_sinkhorn_kernel_matches_framework_iterations(m, n):
    """rtk_log_sinkhorn (500 iterations in one launch) against log_optimal_transport (track4d_utils.py:405-434 restated with
    framework ops), and the mutual-best assignment built on either."""
    from ratrack_amd import association as A
    g = torch.Generator(DEV).manual_seed(m * 131 + n)
    aff = torch.rand(1, m, n, device=DEV, generator=g)
    ref = A.log_optimal_transport(aff, torch.tensor(0.9, device=DEV), 500)
    out = A._log_optimal_transport_hip(aff, 0.9, 500)
    assert out.shape == ref.shape
    assert float((out - ref).abs().max()) <= 2e-4 * max(1.0, float(ref.abs().max()))
    scores = ref
    max0, max1 = scores[:, :-1, :-1].max(2), scores[:, :-1, :-1].max(1)
    idx = A.sinkhorn_assignment(aff)
    assert idx.shape == (1, n) and int(idx.max()) < m
    # where the framework result has a clear margin, the kernel path picks the same partner
    top2 = scores[:, :-1, :-1].topk(min(2, m), dim=1).values
    clear = (top2[:, 0] - top2[:, -1] > 1e-3) if m > 1 else torch.ones(1, n, dtype=torch.bool, device=DEV)
    pick = torch.where(idx >= 0, idx, max1.indices)
    assert torch.equal(pick[clear], max1.indices[clear])


def test_padded_variable_n_batch():
    """the three real View-of-Delft example frames (N = 322 / 352 / 242) as ONE padded batch of frame-pairs through
    the fused path (clouds padded with copies of their point 0, true counts in n_valid) against every pair run on its own,
    unpadded, at B = 1 through the module path (what the reference does): valid columns agree within the north-star tolerance;
    and B = 1 pairs whose two frames differ in size go through the fused path directly (internal padding)."""
    import os
    from _util import GOLDEN, RTOL, reference_state_dict, rel_err
    from ratrack_amd import vod_gt, vod_io
    from ratrack_amd.track4d import Args, Track4D
    ex = os.path.join(GOLDEN, "vod_example")
    scans = [vod_io.load_radar_bin(os.path.join(ex, "radar_%s.bin" % f)) for f in ("00549", "01047", "01201")]
    pairs = [vod_io.frame_pair_tensors(scans[i], scans[(i + 1) % 3], device=DEV) for i in range(3)]
    pc1, pc2, f1, f2, nv = vod_gt.pad_frame_pairs(pairs, device=DEV)
    net = Track4D(Args()).to(DEV).eval()
    net.load_state_dict(reference_state_dict(DEV), strict=True)
    names = ["flow", "h", "cls", "cor", "pc1_features", "pc2_features", "prop"]
    with torch.no_grad():
        h0 = torch.randn(5, 3, 128, device=DEV, generator=torch.Generator(DEV).manual_seed(17)) * 0.1      # (seeded: the comparison is data dependent)
        out = net.backbone(pc1, pc2, f1, f2, h0, n_valid=nv)
        net._use_fused = False
        ref = net.backbone(pc1, pc2, f1, f2, h0, n_valid=nv)                  # per-sample, unpadded, module path
        singles = [net.backbone(*p, h0[:, b:b + 1].contiguous()) for b, p in enumerate(pairs)]
        net._use_fused = True
        fused_singles = [net.backbone(*p, h0[:, b:b + 1].contiguous()) for b, p in enumerate(pairs)]
    for b in range(3):
        n1, n2 = int(nv[0, b]), int(nv[1, b])
        for name, a, r, s_, fs in zip(names, out, ref, singles[b], fused_singles[b]):
            if name == "h":
                a, r = a[:, b], r[:, b]
                s_, fs = s_[:, 0], fs[:, 0]
            else:
                n = n2 if name == "pc2_features" else n1
                a, r = a[b, ..., :n], r[b, ..., :n]
                s_, fs = s_[0], fs[0]
            assert s_.shape == a.shape == fs.shape, (name, s_.shape, a.shape, fs.shape)
            assert torch.equal(r, s_), name                                    # the per-sample fallback IS the B = 1 run
            assert rel_err(a.cpu(), s_.cpu()) <= RTOL, (b, name, rel_err(a.cpu(), s_.cpu()))
            assert rel_err(fs.cpu(), s_.cpu()) <= RTOL, (b, name, "B=1 fused with internal padding")


def test_in_place_weight_edit_invalidates_the_folded_engine():
    """Round-2 review: `load_state_dict` / `train()` / `.to()` rebuilt the engine, an in-place edit of a weight in eval mode did not."""
    from _util import inputs_of, load_case
    net = _net()
    pc1, pc2, f1, f2 = inputs_of(load_case("eval_b2_n256"), DEV)
    with torch.no_grad():
        a = net.backbone(pc1, pc2, f1, f2, None)[0].clone()
        eng = net._fused
        assert torch.equal(a, net.backbone(pc1, pc2, f1, f2, None)[0]) and net._fused is eng      # untouched weights: same engine
        net.fd_layer.fp.conv2.weight.mul_(2.0)                                                       # in place, eval mode
        b = net.backbone(pc1, pc2, f1, f2, None)[0]
        assert net._fused is not eng
        assert torch.allclose(b, 2.0 * a, rtol=1e-5, atol=1e-6)                                      # the flow head's last layer is linear
        net.fd_layer.fp.sf_mlp[0][1].running_var.add_(0.5)                                           # a buffer
        eng2 = net._fused
        net.backbone(pc1, pc2, f1, f2, None)
        assert net._fused is not eng2


def test_fused_real_frames_match_reference_fixture():
    """The reference graph run on the radar frames it ships (tools/make_golden.py real_*: B = 1, N1 != N2, eval mode) against the
    fused engine -- (a) every pair on its own (internal padding of the smaller cloud), (b) the three pairs as ONE padded batch with
    n_valid: flow, cls, h, the recurrent second step, strided samples of cor / point features / prop, FPS / ball / three-NN
    indices through the engine's geometry, EPE."""
    from _util import REAL_CASES, RTOL, assert_close, inputs_of, load_case, reference_state_dict
    from ratrack_amd import vod_gt
    from ratrack_amd.track4d import Args, Track4D
    net = Track4D(Args()).to(DEV).eval()
    net.load_state_dict(reference_state_dict(DEV), strict=True)
    cases = [load_case(n) for n in REAL_CASES]
    cpu = lambda t: t.float().cpu().numpy()

    def check(case, out, out2, b, n1, n2, what):
        flow, h, cls, cor, pf1, pf2, prop = out
        assert_close(cpu(flow[b, :, :n1])[None], case["flow"], RTOL, what + " flow")
        assert_close(cpu(cls[b, :n1])[None], case["cls"], RTOL, what + " cls")
        assert_close(cpu(h[:, b:b + 1]), case["h_out"], RTOL, what + " h")
        assert_close(cpu(cor[b, :, :n1:8])[None], case["cor_s8"], RTOL, what + " cor")
        assert_close(cpu(prop[b, :, :n1:8])[None], case["prop_s8"], RTOL, what + " prop")
        assert_close(cpu(pf1[b, :, :n1:8])[None], case["pc1_features_s8"], RTOL, what + " pc1_features")
        assert_close(cpu(pf2[b, :, :n2:8])[None], case["pc2_features_s8"], RTOL, what + " pc2_features")
        assert_close(cpu(out2[0][b, :, :n1])[None], case["flow_step2"], RTOL, what + " flow (recurrent step)")
        assert_close(cpu(out2[1][:, b:b + 1]), case["h_out_step2"], RTOL, what + " h (recurrent step)")
        gt = torch.from_numpy(case["in_gt_warp"]).to(DEV)
        pc1 = torch.from_numpy(case["in_pc1"]).to(DEV)
        epe = float(torch.sqrt(((pc1 + flow[b:b + 1, :, :n1] - gt) ** 2).sum(1) + 1e-20).mean())
        ref = float(case["metric_sf_vals"][list(case["metric_sf_keys"]).index("epe")])
        assert abs(epe - ref) <= 1e-4 * max(ref, 1.0), (what, epe, ref)

    with torch.no_grad():
        for name, case in zip(REAL_CASES, cases):                           # (a) B = 1, frames of different sizes
            pc1, pc2, f1, f2 = inputs_of(case, DEV)
            out = net.backbone(pc1, pc2, f1, f2, None)
            out2 = net.backbone(pc1, pc2, f1, f2, out[1])
            check(case, out, out2, 0, pc1.shape[2], pc2.shape[2], name)
        pairs = [inputs_of(c, DEV) for c in cases]                          # (b) one padded batch
        pc1, pc2, f1, f2, nv = vod_gt.pad_frame_pairs(pairs, device=DEV)
        out = net.backbone(pc1, pc2, f1, f2, None, n_valid=nv)
        out2 = net.backbone(pc1, pc2, f1, f2, out[1], n_valid=nv)
        for b, (name, case) in enumerate(zip(REAL_CASES, cases)):
            check(case, out, out2, b, int(nv[0, b]), int(nv[1, b]), name + " (padded batch)")
        # index tensors of the engine's geometry for the pc1 clouds of the padded batch: bit-exact with the reference's
        from ratrack_amd import fused
        xyz = torch.cat([pc1, pc2], 0).permute(0, 2, 1).contiguous()
        geo = fused.Geometry(xyz, 512, knn_frames=3, n_valid=nv.reshape(-1).contiguous())
        torch.cuda.synchronize()
        for b, case in enumerate(cases):
            n1 = int(nv[0, b])
            nu = int(geo.nuniq[0][b])
            assert np.array_equal(geo.fps_idx[0][b, :nu].cpu().numpy(), case["fps_idx_c0_l1"][0, :nu]), ("fps", b)
            assert (case["fps_idx_c0_l1"][0, nu:] == 0).all()
            k = np.sort(geo.knn[0][b, :n1].cpu().numpy(), axis=-1)
            ok = case["knn_kth_gap_0"][0] > 0
            assert np.array_equal(k[ok], case["knn_set_0"][0][ok]), ("knn", b)


@pytest.mark.parametrize("n,eps,ms,seed", [(256, 1.5, 2, 0), (300, 1.5, 2, 1), (1000, 1.2, 4, 2), (2048, 0.9, 3, 3), (64, 0.5, 2, 4),
                                           (500, 1.5, 6, 5), (10, 1.5, 2, 6), (3000, 1.1, 2, 8), (4096, 1.0, 3, 7)])      # > ~2900: global workspace
def test_dbscan_kernel_matches_host_restatement(n, eps, ms, seed):
    """rtk_dbscan (mover selection + DBSCAN in one launch) against association.dbscan -- the host restatement that
    tests/test_association_cpu.py pins to scikit-learn -- on clustered clouds, incl. min_samples > 2 (border points, which
    join the lowest-numbered cluster among their core neighbours) and chains of points (many propagation rounds)."""
    from ratrack_amd import association as A
    g = torch.Generator().manual_seed(seed)
    centres = torch.randn(12, 8, generator=g) * 6
    feats8 = centres[torch.randint(0, 12, (n,), generator=g)] + torch.randn(n, 8, generator=g) * 0.45
    feats8[: n // 8] = torch.randn(n // 8, 8, generator=g) * 12                    # scattered noise points
    k = min(40, n // 2)
    feats8[n - k:] = torch.arange(k).float().unsqueeze(1) * torch.tensor([0.6 * eps, 0, 0, 0, 0, 0, 0, 0]) + 50.0      # a chain: one long component
    pf = torch.randn(1, 139, n, generator=g)
    pf[0, [3, 4, 5, 6, 7, 8, 10, 11]] = feats8.t()
    cls = torch.rand(1, n, generator=g)
    objs = A.cluster_objects_device(pf.to(DEV), cls.to(DEV), eps=eps, min_samples=ms)
    mov = (cls > 0.5).squeeze(0)
    ref = A.cluster_objects(pf[:, :, mov], eps=eps, min_samples=ms)
    assert len(objs) == len(ref) and (len(ref) > 1 or n < 64)
    for a, b in zip(objs, ref):
        assert torch.equal(a.cpu(), b)
    # all-noise and no-mover inputs
    assert A.cluster_objects_device(pf.to(DEV), torch.zeros(1, n, device=DEV), eps=eps, min_samples=ms) == []
    far = pf.clone()
    far[0, 3] = torch.arange(n).float() * 100
    assert A.cluster_objects_device(far.to(DEV), torch.ones(1, n, device=DEV), eps=eps, min_samples=2) == []


def test_eval_mode_uses_fused_engine_with_autograd_enabled():
    """The reference's evaluation loop calls net.eval() but never torch.no_grad() (main_utils.py:44-127): the fused engine must
    serve that call too (outputs without a graph), and must be rebuilt after .to() / weight moves."""
    from _util import reference_state_dict
    from ratrack_amd import synth
    from ratrack_amd.track4d import Args, Track4D
    net = Track4D(Args()).to(DEV).eval()
    net.load_state_dict(reference_state_dict(DEV), strict=True)
    d = synth.make_frame_pairs(2, 256, 3)
    t = [torch.from_numpy(d[k]).to(DEV) for k in ("pc1", "pc2", "feature1", "feature2")]
    with torch.no_grad():
        ref = net.backbone(*t, None)
    eng = net._fused
    assert eng
    out = net.backbone(*t, None)                       # grad mode ON
    assert net._fused is eng and all(o.grad_fn is None and not o.requires_grad for o in out)
    for a, b in zip(out, ref):
        assert torch.equal(a, b)
    net.float()                                        # Module._apply
    assert net._fused is None
    x = t[0].clone().requires_grad_(True)              # an input that wants a gradient: the differentiable module path
    flow = net.backbone(x, *t[1:], None)[0]
    assert flow.grad_fn is not None


@pytest.mark.parametrize("nsample,c1,samples,n,npoint", [(32, 64, 3, 200, 77), (16, 64, 8, 256, 256), (16, 32, 5, 128, 130)])
def test_sa_scale_split_agrees_with_fp32_mfma_kernel(nsample, c1, samples, n, npoint):
    """rtk_sa_scale_split against rtk_sa_scale on the same operands: both grids (samples % 8), centroid counts that leave the last
    tile partly empty, duplicate-centroid and duplicate-source counters (rows beyond them untouched / aliased to row 0)."""
    from ratrack_amd import _lib
    torch.manual_seed(nsample + c1)
    xyz, new_xyz = torch.randn(samples, n, 3, device=DEV), torch.randn(samples, npoint, 3, device=DEV)
    idx = torch.randint(0, n, (samples, npoint, nsample), device=DEV, dtype=torch.int32)
    q = torch.randn(samples * n, c1, device=DEV)
    w1 = torch.randn(c1, 4, device=DEV) * 0.3                                  # [Wx | b1]
    w2, b2 = torch.randn(64, c1, device=DEV) / 8, torch.randn(64, device=DEV) * 0.1
    w1img = F.offset_image(w1.double(), DEV)
    chain = F.Chain([(w2.double(), b2.double(), F.ACT_RELU)], DEV)
    img, inv = F.pack_split_device(w2)
    src_nu = torch.randint(n // 2, n + 1, (samples,), device=DEV, dtype=torch.int32)
    dst_nu = torch.randint(npoint // 2, npoint + 1, (samples,), device=DEV, dtype=torch.int32)
    a = torch.full((samples * npoint, 96), -5.0, device=DEV)
    b = a.clone()
    common = (samples, n, npoint, nsample, xyz.data_ptr(), new_xyz.data_ptr(), idx.data_ptr(), q.data_ptr(), c1)
    _lib.call("rtk_sa_scale", *common, c1 // 16, w1img.data_ptr(), 1, chain.arr, a.data_ptr(), 96, 16, src_nu.data_ptr(), dst_nu.data_ptr(),
              F._stream())
    _lib.call("rtk_sa_scale_split", *common, c1, w1img.data_ptr(), img.data_ptr(), inv.data_ptr(), b2.data_ptr(), b.data_ptr(), 96, 16,
              src_nu.data_ptr(), dst_nu.data_ptr(), F._stream())
    assert torch.equal(a == -5.0, b == -5.0)                                   # the same rows / columns are written
    assert rel_err(b.cpu(), a.cpu()) < 5e-6
    with pytest.raises(_lib.RtkError):
        _lib.call("rtk_sa_scale_split", samples, n, npoint, 8, xyz.data_ptr(), new_xyz.data_ptr(), idx.data_ptr(), q.data_ptr(), c1, c1,
                  w1img.data_ptr(), img.data_ptr(), inv.data_ptr(), b2.data_ptr(), b.data_ptr(), 96, 16, None, None, F._stream())


def test_split16_image_is_packed_as_documented():
    """fused.pack_layer_split16 against its index formula, element by element, on a ragged layer (cout, cin not multiples of 16/32)."""
    torch.manual_seed(9)
    cout, cin = 40, 50
    w = torch.randn(cout, cin)
    img, inv = F.pack_layer_split16(w)                                     # (up, v, p, lane, t) int16
    scale, inv2 = F.pow2_scale(w.abs().max())
    assert inv == inv2
    pieces = F.split2_f16(torch.nn.functional.pad(w, (0, 64 - cin, 0, 48 - cout)), scale)      # (2, 48, 64)
    V, U2 = 3, 2
    assert img.numel() == U2 * V * 2 * 64 * 8
    for up in range(U2):
        for v in range(V):
            for p in range(2):
                for lane in (0, 5, 17, 38, 63):
                    g, i = lane // 16, lane % 16
                    for t in range(8):
                        got = img[(((up * V + v) * 2 + p) * 64 + lane) * 8 + t]
                        assert got == pieces[p, 16 * v + i, 16 * (2 * up + t // 4) + 4 * g + t % 4]


@pytest.mark.gpu
@pytest.mark.parametrize("B,N", [(8, 250), (16, 64), (32, 256)])
def test_cost_volume_on_a_share_of_the_cus_is_the_same_result(B, N):
    """rtk_cost_volume_split_shared (what GraphPipeline launches from depth 3: 3/4 of the CUs or fewer, tiles indexed flat per XCD)
    against the full-chip launch: bit for bit."""
    net = _net()
    eng = F.FusedBackbone(net)
    torch.manual_seed(B + N)
    x1, x2 = torch.randn(B, N, 3, device=DEV), torch.randn(B, N, 3, device=DEV)
    p1, p2 = torch.randn(B * N, 256, device=DEV), torch.randn(B * N, 256, device=DEV)
    k1 = PU.knn_point(16, x2, x1)
    out = []
    for shared in (False, True):
        eng.cv_shared = shared
        o = torch.full((B * N + 4, 256), 7.0, device=DEV)
        eng._cost_volume(B, N, x1, x2, k1, p1, p2, o)
        assert torch.all(o[B * N:] == 7.0)
        out.append(o[:B * N])
    wgs = F.cv_shared_workgroups(B, N, DEV)
    cus = torch.cuda.get_device_properties(DEV).multi_processor_count
    assert 0 < wgs <= cus * 3 // 4 and wgs % 8 == 0
    assert torch.equal(out[0], out[1])


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,limit", [(8, 250, 3 * 250), (16, 64, 5 * 64 + 7), (3, 243, 243)])
def test_cost_volume_in_slices_of_the_batch_is_the_same_result(B, N, limit, monkeypatch):
    """The split cost volume requests its gathered rows with 32-bit byte offsets (2^22 rows of p2 per launch, csrc/fused_split.hip);
    FusedBackbone._cost_volume cuts larger batches into slices of whole samples.  With the limit lowered so that a small batch is cut
    (into 3 + 3 + 2, 5 + 5 + 5 + 1 and 1 + 1 + 1 samples): bit for bit the one-launch result, nothing written beyond the batch."""
    net = _net()
    eng = F.FusedBackbone(net)
    torch.manual_seed(B * N)
    x1, x2 = torch.randn(B, N, 3, device=DEV), torch.randn(B, N, 3, device=DEV)
    p1, p2 = torch.randn(B * N, 256, device=DEV), torch.randn(B * N, 256, device=DEV)
    k1 = PU.knn_point(16, x2, x1)
    out = []
    for lim in (F.CV_SPLIT_MAX_ROWS, limit):
        monkeypatch.setattr(F, "CV_SPLIT_MAX_ROWS", lim)
        o = torch.full((B * N + 4, 256), 7.0, device=DEV)
        eng._cost_volume(B, N, x1, x2, k1, p1, p2, o)
        assert torch.all(o[B * N:] == 7.0)
        out.append(o[:B * N])
    assert torch.equal(out[0], out[1])


def _geometry_tables(geo):
    """Every table of a fused.Geometry as a flat dict of tensors (for bit-for-bit comparison)."""
    out = {"tie": geo.tie}
    for l in range(3):
        out["fps_idx%d" % l], out["xyz%d" % (l + 1)], out["nuniq%d" % l] = geo.fps_idx[l], geo.xyz[l + 1], geo.nuniq[l]
        for s in range(2):
            out["ball%d%d" % (l, s)] = geo.ball[l][s]
    for name, (d2, idx, m) in geo.nn.items():
        out["nn_idx_" + name], out["nn_d2_" + name] = idx, d2
    if geo.knn is not None:
        out["knn12"], out["knn11"] = geo.knn
    return out


@pytest.mark.parametrize("case", ["synthetic_b4_n256", "lattice_ties", "n1024", "n2048", "n600", "padded", "odd_n", "tiny", "point_major_one_frame"])
def test_two_launch_geometry_equals_the_separate_launches(case, monkeypatch):
    """rtk_geometry_front + rtk_geometry_tables (two launches, the product path since round 6) against the eleven launches of the ops' own
    entry points (rtk_prepare_inputs, rtk_fps_centroids, rtk_fps_relevel x 2, rtk_ball_query_pair x 3, rtk_three_nn_masked x 3,
    rtk_knn_point_masked x 2): every table bit for bit -- FPS indices / centroids / exhausted-cloud and tie counters of the three levels,
    the six ball tables, the three three-NN tables (indices and squared distances), both kNN tables, xyz and raw.  Clouds with exact
    distance ties between distinct points (the tied levels resume and settle inside the one launch), duplicates, padded batches, odd
    sizes; with the layout conversion from the API's channel-major tensors and without (the training path hands over point-major clouds)."""
    from ratrack_amd import synth
    g = torch.Generator().manual_seed(11)
    nv = None
    if case == "synthetic_b4_n256":
        d = synth.make_frame_pairs(4, 256, case_id=3)
        pc1, pc2 = torch.from_numpy(d["pc1"]).to(DEV), torch.from_numpy(d["pc2"]).to(DEV)
    elif case == "lattice_ties":
        pc1 = torch.randint(0, 6, (3, 3, 300), generator=g).float().to(DEV)
        pc2 = torch.randint(0, 5, (3, 3, 300), generator=g).float().to(DEV)
    elif case in ("n1024", "n2048", "n600"):      # 16, 32 and 16 points per lane in the selection wave; 2048 = the largest fused cloud
        d = synth.make_frame_pairs(2, int(case[1:]), case_id=5)
        pc1, pc2 = torch.from_numpy(d["pc1"]).to(DEV), torch.from_numpy(d["pc2"]).to(DEV)
    elif case == "padded":
        d = synth.make_frame_pairs(3, 352, case_id=9)
        pc1, pc2 = torch.from_numpy(d["pc1"]).to(DEV), torch.from_numpy(d["pc2"]).to(DEV)
        nv = torch.tensor([[322, 352, 242], [300, 17, 352]], dtype=torch.int32, device=DEV)
        for f, pc in enumerate((pc1, pc2)):       # padding = copies of the cloud's own point 0
            for b in range(3):
                pc[b, :, int(nv[f, b]):] = pc[b, :, :1]
        nv = nv.reshape(-1).contiguous()
    elif case == "odd_n":
        pc1, pc2 = torch.randn(5, 3, 243, generator=g).to(DEV) * 6, torch.randn(5, 3, 243, generator=g).to(DEV) * 6
    elif case == "tiny":
        pc1, pc2 = torch.randn(2, 3, 16, generator=g).to(DEV), torch.randn(2, 3, 16, generator=g).to(DEV)
    else:
        pc1 = torch.randint(0, 7, (5, 3, 200), generator=g).float().to(DEV)
        pc2 = None
    if pc2 is not None:
        B, _, N = pc1.shape
        f1, f2 = torch.randn(B, 2, N, generator=g).to(DEV), torch.randn(B, 2, N, generator=g).to(DEV)

        def build():
            xyz = torch.empty(2 * B, N, 3, device=DEV)
            raw = torch.full((2 * B * N, 4), float("nan"), device=DEV)
            geo = F.Geometry(xyz, 512, knn_frames=B, n_valid=nv, finite=True, prepare=(pc1, pc2, f1, f2, raw))
            torch.cuda.synchronize()
            t = _geometry_tables(geo)
            t["xyz0"], t["raw"] = xyz, raw
            return geo, t
    else:
        xyz = pc1.permute(0, 2, 1).contiguous()       # five clouds of one frame, point-major, no kNN (the training path's call)

        def build():
            geo = F.Geometry(xyz, 512, finite=True)
            torch.cuda.synchronize()
            return geo, _geometry_tables(geo)
    geo2, two = build()
    assert geo2.fused_geometry
    monkeypatch.setattr(F, "FUSED_GEOMETRY", False)
    geo11, eleven = build()
    assert not geo11.fused_geometry
    assert two.keys() == eleven.keys()
    for k in two:
        assert two[k].shape == eleven[k].shape, k
        if two[k].dtype.is_floating_point:      # same bits (NaN-safe)
            assert torch.equal(two[k].view(torch.int32), eleven[k].view(torch.int32)), k
        else:
            assert torch.equal(two[k], eleven[k]), k
    if case == "lattice_ties":
        assert int(two["tie"].sum()) > 0 and bool((two["xyz2"] != two["xyz1"]).any()), "the lattice must exercise the tied levels"


@pytest.mark.parametrize("B,N", [(4, 256), (3, 242), (2, 1000), (5, 37)])
def test_unfilled_geometry_workspace_is_never_read(B, N, monkeypatch):
    """The eval path no longer zero-fills the geometry's index workspace (9 MB per batch): the two launches write every entry a consumer
    reads.  Two runs with the workspace pre-filled with two different (valid) indices must agree in every output bit -- a stray read
    of an unwritten entry would gather a different row in the two runs."""
    from ratrack_amd import synth
    net = _net()
    d = synth.make_frame_pairs(B, N, case_id=77)      # (N = 242, 37: the last tile of live centroids is partial)
    t = [torch.from_numpy(d[k]).to(DEV) for k in ("pc1", "pc2", "feature1", "feature2")]
    outs = []
    for poison in (1, 200, None):
        monkeypatch.setattr(F, "GEOMETRY_POISON", poison)
        with torch.no_grad():
            o = net.backbone(*t, None)
        torch.cuda.synchronize()
        outs.append([x.detach().clone() for x in o])
    for a, b, c in zip(*outs):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)) and torch.equal(a.view(torch.int32), c.view(torch.int32))


def test_global_terms_copy_multi_and_gru_head():
    """The three small entry points of round 6 against the framework: rtk_global_terms (per-sample linear maps of the global feature over
    sample ranges + the broadcast over a sample's rows), rtk_copy_multi (several copies, one launch; odd sizes and unaligned tails),
    rtk_gru_step_head (rtk_gru_step + a linear map of its output)."""
    from ratrack_amd import _lib
    g = torch.Generator().manual_seed(3)
    S, cin, n = 6, 128, 37
    glob = torch.randn(S, cin, generator=g).to(DEV)
    w1, b1, w2, w3 = (torch.randn(256, cin, generator=g).to(DEV), torch.randn(256, generator=g).to(DEV), torch.randn(40, cin, generator=g).to(DEV),
                      torch.randn(32, cin, generator=g).to(DEV))
    o1, o2, o3 = torch.empty(3, 256, device=DEV), torch.empty(3, 40, device=DEV), torch.empty(6, 32, device=DEV)
    dst = torch.full((S * n, 200), -7.0, device=DEV)
    F.global_terms(glob, [(w1.t().contiguous(), b1, o1, 0), (w2.t().contiguous(), None, o2, 3), (w3.t().contiguous(), None, o3, 0)],
                   bcast=dst[:, 64:], n=n)
    ref = lambda w, b, x: (x.double() @ w.double().t() + (0 if b is None else b.double())).float()
    assert rel_err(o1.cpu(), ref(w1, b1, glob[:3]).cpu()) < 1e-6 and rel_err(o2.cpu(), ref(w2, None, glob[3:]).cpu()) < 1e-6
    assert rel_err(o3.cpu(), ref(w3, None, glob).cpu()) < 1e-6
    assert torch.equal(dst[:, 64:64 + cin], glob.repeat_interleave(n, 0)) and (dst[:, :64] == -7).all() and (dst[:, 64 + cin:] == -7).all()
    # copies
    srcs = [torch.randn(k, generator=g).to(DEV) for k in (1, 5, 4096, 70001, 16384 * 3 + 2)] + [torch.arange(999, dtype=torch.int32, device=DEV)]
    big = torch.randn(100003, generator=g).to(DEV)
    srcs.append(big[1:])                                      # 4-byte aligned only
    dsts = [torch.zeros_like(s_) for s_ in srcs[:-1]] + [torch.zeros(100002, device=DEV)]
    F.copy_multi(list(zip(dsts, [s_.contiguous() for s_ in srcs])))
    for d_, s_ in zip(dsts, srcs):
        assert torch.equal(d_, s_)
    # GRU step with the head epilogue
    B, L, H = 5, 5, 128
    gru = torch.nn.GRU(H, H, L).to(DEV)
    x, h0 = torch.randn(B, H, generator=g).to(DEV), torch.randn(L, B, H, generator=g).to(DEV)
    wh, bh = torch.randn(96, H, generator=g).to(DEV), torch.randn(96, generator=g).to(DEV)
    st = lambda k: torch.stack([getattr(gru, "%s_l%d" % (k, l)).detach() for l in range(L)])
    wih, whh = st("weight_ih").transpose(1, 2).contiguous(), st("weight_hh").transpose(1, 2).contiguous()
    bih, bhh = st("bias_ih").contiguous(), st("bias_hh").contiguous()
    h_out, y, head = torch.empty_like(h0), torch.empty(B, H, device=DEV), torch.empty(B, 96, device=DEV)
    _lib.call("rtk_gru_step_head", B, L, H, x.data_ptr(), h0.data_ptr(), wih.data_ptr(), whh.data_ptr(), bih.data_ptr(), bhh.data_ptr(),
              h_out.data_ptr(), y.data_ptr(), wh.t().contiguous().data_ptr(), bh.data_ptr(), head.data_ptr(), 96, F._stream())
    with torch.no_grad():
        yr, hr = gru(x.unsqueeze(0), h0)
    assert rel_err(y.cpu(), yr[0].cpu()) < 1e-5 and rel_err(h_out.cpu(), hr.cpu()) < 1e-5
    assert rel_err(head.cpu(), ref(wh, bh, y).cpu()) < 1e-6
