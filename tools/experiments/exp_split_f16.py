#!/usr/bin/env python3
"""Round 5: error of a TWO-piece fp16 split with THREE partial products (h.h + h.l + l.h, fp32 accumulate) against the three-piece
bf16 split with six products and against an fp32 matmul, all measured against float64 -- numpy only, no GPU.

  x = h + l + e,  h = fp16(x) (round to nearest even),  l = fp16(x - h);  |e| <= 2^-23 |x| while l is a normal fp16 number.
  A fp16 x fp16 product is exact in fp32 (11 + 11 significant bits), the dropped l.l term is below 2^-22 |a||b|.

Range: fp16 overflows above 65504 and l loses relative precision once it is subnormal (|l| < 2^-14, i.e. |x| < ~2^-3) -- but its
ABSOLUTE error stays below 2^-25 (subnormals kept) which is what matters against max|y|.  The experiment sweeps operand scales
(exact powers of two applied before the split and undone after the product) to show where the scheme holds.

python tools/experiments/exp_split_f16.py"""
import numpy as np

rng = np.random.default_rng(0)
K = 256


def split3_bf16(x):
    x = x.astype(np.float32)
    trunc = lambda v: (v.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
    p0 = trunc(x); r1 = x - p0; p1 = trunc(r1); r2 = r1 - p1; p2 = trunc(r2)
    return p0, p1, p2


def split2_f16(x, flush=False, rtz=False):
    x = x.astype(np.float32)
    if rtz:
        def cvt(v):                                     # round toward zero to fp16 (normal range only; good enough for the experiment)
            h = v.astype(np.float16)
            over = np.abs(h.astype(np.float32)) > np.abs(v)
            hb = h.view(np.uint16).copy()
            hb[over] -= 1
            return hb.view(np.float16)
    else:
        cvt = lambda v: v.astype(np.float16)
    h = cvt(x)
    r = x - h.astype(np.float32)                        # exact
    l = cvt(r)
    if flush:                                           # matrix core flushing subnormal inputs (MI200 behaviour)
        l = np.where(np.abs(l.astype(np.float32)) < 2.0 ** -14, np.float16(0), l)
        h = np.where(np.abs(h.astype(np.float32)) < 2.0 ** -14, np.float16(0), h)
    return h.astype(np.float32), l.astype(np.float32)


def acc32(terms, shape):
    out = np.zeros(shape, np.float32)
    for k0 in range(0, K, 16):                          # one MFMA k-step: 16 exact products summed, then one fp32 accumulate
        for a, b in terms:
            out = (out + (a[:, k0:k0 + 16].astype(np.float64) @ b[k0:k0 + 16].astype(np.float64)).astype(np.float32)).astype(np.float32)
    return out


def report(name, o, ref, sc):
    e = np.abs(o.astype(np.float64) - ref)
    print("  %-34s max err / max|y| %.3e   max err / (|W||X|) %.3e   rms / max|y| %.3e" %
          (name, e.max() / np.abs(ref).max(), (e / sc).max(), np.sqrt((e ** 2).mean()) / np.abs(ref).max()))
    return e.max() / np.abs(ref).max()


def case(title, W, X):
    print(title)
    ref = W.astype(np.float64) @ X.astype(np.float64)
    sc = np.abs(W).astype(np.float64) @ np.abs(X) + 1e-300
    shape = ref.shape
    report("fp32 matmul", W @ X, ref, sc)
    w, x = split3_bf16(W), split3_bf16(X)
    six = [(w[2], x[0]), (w[0], x[2]), (w[1], x[1]), (w[1], x[0]), (w[0], x[1]), (w[0], x[0])]
    report("bf16 x3, six products", acc32(six, shape), ref, sc)
    report("bf16 x3, three products", acc32(six[3:], shape), ref, sc)
    for sw, sx in ((0, 0), (8, 0), (8, 6), (12, 10)):
        for flush in (False, True):
            wh, wl = split2_f16(W * np.float32(2.0 ** sw), flush)
            xh, xl = split2_f16(X * np.float32(2.0 ** sx), flush)
            if not (np.isfinite(wh).all() and np.isfinite(xh).all()):
                print("  fp16 x2 (2^%d, 2^%d): OVERFLOW" % (sw, sx))
                continue
            o = acc32([(wl, xh), (wh, xl), (wh, xh)], shape) * np.float32(2.0 ** -(sw + sx))
            report("fp16 x2 RNE, W*2^%d X*2^%d%s" % (sw, sx, " flush" if flush else ""), o, ref, sc)
    wh, wl = split2_f16(W * np.float32(2.0 ** 8), rtz=True)
    xh, xl = split2_f16(X * np.float32(2.0 ** 6), rtz=True)
    report("fp16 x2 RTZ, W*2^8 X*2^6", acc32([(wl, xh), (wh, xl), (wh, xh)], shape) * np.float32(2.0 ** -14), ref, sc)


W = (rng.standard_normal((256, K)) / 16).astype(np.float32)
X = rng.standard_normal((K, 512)).astype(np.float32)
case("gaussian W / 16, gaussian X (the cost volume's scale)", W, X)
case("post-ReLU X (half zeros), W as trained-like (heavy tails)", (rng.standard_t(3, (256, K)) / 20).astype(np.float32),
     np.maximum(rng.standard_normal((K, 512)), 0).astype(np.float32))
case("wide dynamic range inside X (x * exp(U(-8, 8)))", W, (rng.standard_normal((K, 512)) * np.exp(rng.uniform(-8, 8, (K, 512)))).astype(np.float32))
case("tiny activations (X * 1e-4)", W, (X * 1e-4).astype(np.float32))
case("large activations (X * 300)", W, (X * 300).astype(np.float32))
