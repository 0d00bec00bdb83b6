"""The reference's own training regime -- B = 1, every pair of a different size -- on the hand-written path: ms per step, eager vs ONE
captured graph that serves all sizes (clouds padded to 384 columns, sizes on the device).  python tools/experiments/time_train_b1.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from _util import load_case, inputs_of, reference_state_dict, REAL_CASES
from ratrack_amd.track4d import Args, Track4D
from ratrack_amd.train import Trainer
DEV = "cuda"
subs = []
for n in REAL_CASES:
    c = load_case(n)
    subs.append({k: v for k, v in c.items() if k.startswith("in_")})
pad = lambda t, n: torch.cat([t, t[..., :1].expand(*t.shape[:-1], n - t.shape[-1])], dim=-1).contiguous()
batches = []
for s in subs:
    pc1, pc2, f1, f2 = inputs_of(s, DEV)
    gt, gc = torch.from_numpy(s["in_gt_warp"]).to(DEV), torch.from_numpy(s["in_gt_cls"]).to(DEV)
    nv = torch.tensor([[pc1.shape[2]], [pc2.shape[2]]], dtype=torch.int32, device=DEV)
    batches.append(((pc1, pc2, f1, f2, gt, gc), (pad(pc1, 384), pad(pc2, 384), pad(f1, 384), pad(f2, 384), pad(gt, 384), pad(gc, 384)), nv))
for mode in ("eager, unpadded (N1 != N2 handled inside backbone)", "eager, padded + n_valid", "one captured graph, padded + n_valid"):
    net = Track4D(Args()).to(DEV)
    net.load_state_dict(reference_state_dict(DEV), strict=True)
    tr = Trainer(net, lr=1e-4, graph=mode.startswith("one"), graph_warmup=3)
    def step(i):
        raw, padded, nv = batches[i % 3]
        if mode.startswith("eager, unpadded"):
            return tr.step(*raw)
        return tr.step(*padded, n_valid=nv)
    for i in range(12): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(60): step(i)
    torch.cuda.synchronize()
    print("%-55s %.2f ms per step" % (mode, (time.perf_counter() - t0) / 60 * 1e3))

# synthetic clouds for comparison: N = 256 / 384, with and without n_valid
from ratrack_amd import synth
for N, use_nv in ((256, False), (384, False), (384, True)):
    d = synth.make_frame_pairs(1, N, 5)
    t = {k: torch.from_numpy(v).to(DEV) for k, v in d.items()}
    net = Track4D(Args()).to(DEV)
    net.load_state_dict(reference_state_dict(DEV), strict=True)
    tr = Trainer(net, lr=1e-4, graph=True, graph_warmup=3)
    nv = torch.tensor([[N - 40], [N - 7]], dtype=torch.int32, device=DEV) if use_nv else None
    f = lambda: tr.step(t["pc1"], t["pc2"], t["feature1"], t["feature2"], t["gt_warp"], t["gt_cls"], n_valid=nv)
    for i in range(12): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(100): f()
    torch.cuda.synchronize()
    print("synthetic B=1 N=%d n_valid=%s, one graph: %.2f ms per step" % (N, use_nv, (time.perf_counter() - t0) / 100 * 1e3))
