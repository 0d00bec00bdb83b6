#!/usr/bin/env python3
"""Error of the split-bf16 product scheme (csrc/split_mfma.h) emulated in numpy: 3, 6 and 9 partial products of the exact three-way
bf16 split against float64, next to an fp32 matmul.  python tools/exp_split_error.py"""
import numpy as np
rng = np.random.default_rng(0)
def split3(x):
    x = x.astype(np.float32)
    def trunc(v): return (v.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
    p0 = trunc(x); r1 = x - p0; p1 = trunc(r1); r2 = r1 - p1; p2 = trunc(r2)
    assert np.array_equal(p2, r2)
    assert np.array_equal((p0.astype(np.float64)+p1+p2).astype(np.float32), x)
    return p0, p1, p2
K=256
W = (rng.standard_normal((256,K))/16).astype(np.float32); X = rng.standard_normal((K,512)).astype(np.float32)
ref = W.astype(np.float64) @ X.astype(np.float64)
f32 = W @ X
def acc32(terms):
    out = np.zeros((256,512), np.float32)
    for a,b in terms:
        # products exact in f32 (bf16 x bf16), accumulation in f32 in k order, blocks of 16
        for k0 in range(0,K,16):
            out = (out + (a[:,k0:k0+16].astype(np.float64) @ b[k0:k0+16].astype(np.float64)).astype(np.float32)).astype(np.float32)
    return out
w = split3(W); x = split3(X)
six = [(w[0],x[0]),(w[0],x[1]),(w[1],x[0]),(w[0],x[2]),(w[2],x[0]),(w[1],x[1])]
three = six[:3]
nine = six + [(w[1],x[2]),(w[2],x[1]),(w[2],x[2])]
sc = np.abs(W).astype(np.float64) @ np.abs(X)
for name, o in (("fp32 matmul", f32), ("3 terms", acc32(three)), ("6 terms", acc32(six)), ("6 small first", acc32(six[::-1])), ("9 terms", acc32(nine))):
    e = np.abs(o - ref)
    print("%-14s max err / max|ref| %.3e   max err/(|W||X|) %.3e  rms %.3e" % (name, e.max()/np.abs(ref).max(), (e/sc).max(), np.sqrt((e**2).mean())))
