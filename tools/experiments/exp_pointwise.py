"""Per-point chain kernel (rtk_pointwise_mlp) at the forward's shapes: microseconds next to the MFMA floor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ratrack_amd import fused as F
from ratrack_amd.benchutil import time_graph
dev = "cuda"
g = torch.Generator(dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g)
# (samples, rows per sample (allocated), live rows, cin, [cout...])
for S, rps, live, cin, couts in [(128, 512, 256, 128, [128]), (128, 512, 256, 64, [96]), (128, 512, 256, 96, [192]), (128, 512, 256, 128, [64]),
                                 (64, 256, 256, 128, [256]), (64, 256, 256, 128, [128, 64, 32, 3]), (64, 512, 256, 128, [128])]:
    rows = S * rps
    x = r(rows, cin)
    layers, ci = [], cin
    for co in couts:
        layers.append((r(co, ci).double() * 0.1, r(co).double(), F.ACT_RELU))
        ci = co
    chain = F.Chain(layers, dev)
    out = torch.empty(rows, ((couts[-1] + 15) // 16) * 16, device=dev)
    nu = torch.full((S,), live, dtype=torch.int32, device=dev)
    ms = time_graph(lambda: F.pointwise(rows, rps, [(x, cin, False)], chain, out, row_nuniq=nu), 10)
    macs, ci = 0, cin
    for co in couts:
        macs += ci * co; ci = co
    gf = 2.0 * S * live * macs / 1e9
    print("S=%3d live rows %d  %d -> %s : %.1f us  (%.1f TFLOP/s, MFMA floor %.1f us)" % (S, live, cin, couts, ms * 1e3, gf / ms, gf / 157.3e-3))
