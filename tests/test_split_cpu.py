"""CPU: the arithmetic the split matrix paths rest on (csrc/split_mfma.h, csrc/fused_common.h), restated in numpy / torch-CPU -- no GPU
involved.
  * round 5 -- TWO fp16 pieces of a power-of-two-scaled fp32 number carry it to 2^-23; a fp16 x fp16 product is exact in fp32; THREE
    partial products (l.h + h.l + h.h) accumulated in fp32 carry the error of an fp32 product chain, at every operand magnitude once
    the scale is taken from the data;
  * rounds 2-4 (still used by the 16-position per-point chains and the training kernels' position contractions) -- three bf16
    pieces, split by truncation, sum to the fp32 number exactly; six partial products carry fp32 accuracy, three do not;
  * the host packers write the images the headers describe."""
import numpy as np
import torch

from ratrack_amd import fused as F


def _split3(x):
    x = x.astype(np.float32)
    trunc = lambda v: (v.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
    p0 = trunc(x); r1 = x - p0; p1 = trunc(r1); r2 = r1 - p1; p2 = trunc(r2)
    return p0, p1, p2, r2


def test_three_truncated_pieces_are_exact():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200000) * np.exp(rng.uniform(-30, 30, 200000))).astype(np.float32)
    x[:4] = [0.0, 1.0, -1.5, 3.0e38]
    p0, p1, p2, r2 = _split3(x)
    assert np.array_equal(p2, r2)                                             # nothing is left after the third piece
    assert np.array_equal((p0.astype(np.float64) + p1 + p2).astype(np.float32), x)
    for p in (p0, p1, p2):                                                    # every piece is a bf16 number
        assert not (p.view(np.uint32) & np.uint32(0xFFFF)).any()
    # the torch restatement the device packer is compared with (tests/test_fused_gpu.py) produces the same pieces
    t = F.split3_bf16(torch.from_numpy(x))
    for k, p in enumerate((p0, p1, p2)):
        assert np.array_equal((t[k].numpy().astype(np.int32) << 16).view(np.float32), p)


def test_bf16_products_are_exact_in_fp32():
    rng = np.random.default_rng(1)
    a = _split3(rng.standard_normal(100000).astype(np.float32))[0]
    b = _split3(rng.standard_normal(100000).astype(np.float32))[1]
    assert np.array_equal((a * b).astype(np.float64), a.astype(np.float64) * b.astype(np.float64))


def test_six_products_carry_fp32_accuracy_three_do_not():
    rng = np.random.default_rng(2)
    K = 256
    W = (rng.standard_normal((64, K)) / 16).astype(np.float32)
    X = rng.standard_normal((K, 96)).astype(np.float32)
    ref = W.astype(np.float64) @ X.astype(np.float64)
    w, x = _split3(W)[:3], _split3(X)[:3]

    def accumulate(terms):                                                    # fp32 accumulation, 16 exact products at a time
        out = np.zeros(ref.shape, np.float32)
        for k0 in range(0, K, 16):
            for a, b in terms:
                out = out + (a[:, k0:k0 + 16].astype(np.float64) @ b[k0:k0 + 16].astype(np.float64)).astype(np.float32)
        return out
    six = [(w[2], x[0]), (w[0], x[2]), (w[1], x[1]), (w[1], x[0]), (w[0], x[1]), (w[0], x[0])]      # the kernels' order: small terms first
    err = lambda o: np.abs(o - ref).max() / np.abs(ref).max()
    e32, e6, e3 = err(W @ X), err(accumulate(six)), err(accumulate(six[3:]))
    assert e6 < 1.5 * e32 + 1e-7, (e6, e32)
    assert e3 > 10 * e32, (e3, e32)


def _split2(x, scale):
    xs = (x.astype(np.float32) * np.float32(scale)).astype(np.float32)
    h = xs.astype(np.float16)
    l = (xs - h.astype(np.float32)).astype(np.float16)
    return h.astype(np.float32), l.astype(np.float32)


def test_two_fp16_pieces_carry_23_bits_under_a_data_scale():
    rng = np.random.default_rng(3)
    for mag in (1e-30, 1e-6, 1.0, 300.0, 1e20):
        x = (rng.standard_normal(100000) * mag).astype(np.float32)
        scale, inv = F.pow2_scale(float(np.abs(x).max()))
        assert 2.0 ** 14 <= float(np.abs(x).max()) * scale < 2.0 ** 15 and scale * inv == 1.0
        h, l = _split2(x, scale)
        assert np.isfinite(h).all()
        xs = np.abs(x).astype(np.float64) * scale
        err = np.abs((h.astype(np.float64) + l) - x.astype(np.float64) * scale)      # in scaled units
        big = xs >= 2.0 ** -2                                                  # the low piece is a normal fp16 number there: 2^-16 of the largest
        assert (err[big] <= 2.0 ** -23 * xs[big]).all()
        assert (err <= 2.0 ** -25 + 2.0 ** -23 * xs).all()                      # below: an ABSOLUTE 2^-25 (fp16 subnormal spacing / 2) = 2^-39 of the largest
    # the torch restatement the device packer is compared with produces the same pieces
    x = rng.standard_normal(4096).astype(np.float32)
    scale, _ = F.pow2_scale(float(np.abs(x).max()))
    t = F.split2_f16(torch.from_numpy(x), scale).view(torch.float16).numpy().astype(np.float32)
    h, l = _split2(x, scale)
    assert np.array_equal(t[0], h) and np.array_equal(t[1], l)


def test_fp16_products_are_exact_in_fp32():
    rng = np.random.default_rng(4)
    a = rng.standard_normal(100000).astype(np.float16).astype(np.float32) * np.float32(1024)
    b = rng.standard_normal(100000).astype(np.float16).astype(np.float32) * np.float32(2.0 ** -10)
    assert np.array_equal((a * b).astype(np.float64), a.astype(np.float64) * b.astype(np.float64))


def test_three_fp16_products_carry_fp32_accuracy_at_every_magnitude():
    rng = np.random.default_rng(5)
    K = 256
    W = (rng.standard_normal((64, K)) / 16).astype(np.float32)
    for mag in (1e-25, 1e-4, 1.0, 300.0, 1e15):
        X = (rng.standard_normal((K, 96)) * mag).astype(np.float32)
        ref = W.astype(np.float64) @ X.astype(np.float64)
        sw, iw = F.pow2_scale(float(np.abs(W).max()))
        wh, wl = _split2(W, sw)
        out = np.zeros(ref.shape, np.float32)
        xh, xl, inv = np.zeros_like(X), np.zeros_like(X), np.zeros(96, np.float64)
        for j in range(96):                                                    # one scale per position (column)
            sx, ix = F.pow2_scale(float(np.abs(X[:, j]).max()))
            xh[:, j], xl[:, j] = _split2(X[:, j], sx)
            inv[j] = ix * iw
        for k0 in range(0, K, 16):                                             # one MFMA k-step: 16 exact products, one fp32 accumulate
            for a, b in ((wl, xh), (wh, xl), (wh, xh)):                        # the kernels' order: small terms first
                out = out + (a[:, k0:k0 + 16].astype(np.float64) @ b[k0:k0 + 16].astype(np.float64)).astype(np.float32)
        got = out.astype(np.float64) * inv[None, :]
        err = lambda o: np.abs(o - ref).max() / np.abs(ref).max()
        e32, e3 = err((W @ X).astype(np.float64)), err(got)
        assert e3 < 1.5 * e32 + 1e-7, (mag, e3, e32)


def test_split_images_follow_their_index_formulas():
    torch.manual_seed(0)
    w = torch.randn(64, 96)
    scale, inv = F.pow2_scale(w.abs().max())
    p = F.split2_f16(w, scale)
    img, inv2 = F.pack_layer_split(w)
    assert inv2 == inv
    img = img.view(6, 2, 2, 64, 8)                                             # (s, v, piece, lane, t)
    for s in range(6):
        for v in range(2):
            for lane in (0, 7, 31, 32, 50, 63):
                hh, i = lane // 32, lane % 32
                for t in range(8):
                    c = 32 * (s // 2) + 16 * (s % 2) + 8 * (t // 4) + 4 * hh + t % 4
                    assert torch.equal(img[s, v, :, lane, t], p[:, 32 * v + i, c])
    w = torch.randn(40, 50)                                                    # ragged: zero padded to (48, 64)
    scale, inv = F.pow2_scale(w.abs().max())
    p = F.split2_f16(torch.nn.functional.pad(w, (0, 14, 0, 8)), scale)
    img, inv2 = F.pack_layer_split16(w)
    assert inv2 == inv
    img = img.view(2, 3, 2, 64, 8)                                             # (up, v, piece, lane, t)
    for up in range(2):
        for v in range(3):
            for lane in (0, 5, 17, 38, 63):
                g, i = lane // 16, lane % 16
                for t in range(8):
                    assert torch.equal(img[up, v, :, lane, t], p[:, 16 * v + i, 16 * (2 * up + t // 4) + 4 * g + t % 4])


def test_shared_cost_volume_workgroup_policy(monkeypatch):
    """fused.cv_shared_workgroups (host policy of rtk_cost_volume_split_shared): a multiple of 8, between half and three quarters of
    the CUs, never more than one workgroup per tile, balanced (slowest workgroup <= 5 % above the mean) where such a count exists,
    0 = all CUs when the batch does not take the XCD-aware grid."""
    class Props:
        multi_processor_count = 256
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda dev: Props)
    assert F.cv_shared_workgroups(64, 256, "cuda") == 192
    assert F.cv_shared_workgroups(3, 243, "cuda") == 0                        # samples % 8 != 0: plain 2-D grid, all CUs
    for samples, n in [(64, 256), (32, 256), (32, 1024), (8, 256), (16, 64), (8, 17), (128, 256), (64, 242), (8, 8)]:
        w = F.cv_shared_workgroups(samples, n, "cuda")
        tiles = (samples // 8) * ((n + 7) // 8)
        assert w % 8 == 0 and 0 < w <= 192 and w // 8 <= tiles, (samples, n, w)
        per = w // 8
        if per < tiles and per >= 16:
            best = max(tiles / (c * -(-tiles // c)) for c in range(16, 25))
            assert tiles / (per * -(-tiles // per)) >= min(0.95, best) - 1e-9, (samples, n, w)
