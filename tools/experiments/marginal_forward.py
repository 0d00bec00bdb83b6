"""Marginal cost of every kernel family in the PIPELINED forward (B=64, N=256, four graphs in flight), measured by issuing the family's
launches TWICE: every launch of the path is idempotent (same inputs, same outputs into the same buffers; the one atomic is a maximum),
so the doubled step computes the same results and its extra time is what one more copy of the family costs where it runs -- no
skipped work downstream (round 5's "without fps" line skipped the selection and with it half of the path: the centroids' counters
collapsed).  Also the classic ablation (family skipped, results wrong) for the families whose outputs are not indices.
python tools/experiments/marginal_forward.py [> profiles/r06_forward_marginal.txt]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ratrack_amd import _lib, fused, synth
from ratrack_amd.track4d import Args, Track4D

dev = torch.device("cuda")
net = Track4D(Args()).to(dev).eval()
synth.fill_state_dict(net.state_dict())
batches = []
for i in range(8):
    d = synth.make_frame_pairs(64, 256, 1000 + 100 * i)
    batches.append([torch.from_numpy(d[k]).to(dev) for k in ("pc1", "pc2", "feature1", "feature2")] + [torch.zeros(5, 64, 128, device=dev)])
FAM = {"cost_volume": ["rtk_cost_volume_split", "rtk_cost_volume_split_shared"], "sa_scale": ["rtk_sa_scale", "rtk_sa_scale_split"],
       "pointwise": ["rtk_pointwise_mlp"], "patch_cost": ["rtk_patch_cost"], "gru": ["rtk_gru_step_head"], "global_terms": ["rtk_global_terms"],
       "geometry_front": ["rtk_geometry_front"], "geometry_tables": ["rtk_geometry_tables"], "input copy": ["rtk_copy_multi"]}
orig = _lib.call


def run(twice=(), skip=(), steps=1500):
    t2 = set(sum((FAM[f] for f in twice), []))
    sk = set(sum((FAM[f] for f in skip), []))

    def call(name, *a):
        if name in sk:
            return 0
        r = orig(name, *a)
        if name in t2:
            orig(name, *a)
        return r
    _lib.call = call
    fused._lib.call = call
    fused.copy_multi.__globals__["_lib"].call = call
    try:
        with torch.no_grad():
            net.invalidate_fused()
            net.backbone(*batches[0])
            pipe = fused.GraphPipeline(net._fused, tuple(batches[0]), depth=4)
            for i in range(400):
                pipe.submit(*batches[i % 8])
            pipe.drain(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                pipe.submit(*batches[i % 8])
            pipe.drain(); torch.cuda.synchronize()
            el = time.perf_counter() - t0
    finally:
        _lib.call = orig
        fused._lib.call = orig
    return el / steps * 1e3


# Every configuration in a process of its own: which hardware queue a stream lands on depends on the streams the process created before,
# and two pipelines built one after the other in ONE process can differ by 20 % for that reason alone (HISTORY.md, round 5).
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    tw = [f for f in sys.argv[2].split(",") if f]
    sk = [f for f in sys.argv[3].split(",") if f]
    print("ONE %.4f" % run(twice=tw, skip=sk), flush=True)
    sys.exit(0)

import subprocess


def one(twice=(), skip=()):
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", ",".join(twice), ",".join(skip)], capture_output=True, text=True).stdout
    return float([l for l in out.split("\n") if l.startswith("ONE ")][-1].split()[1])


base = [one() for _ in range(3)]
b = sorted(base)[1]
print("baseline %.4f ms/batch (three processes: %s)" % (b, ", ".join("%.4f" % x for x in base)))
print("family issued TWICE (same results):")
tot = 0.0
for f in FAM:
    ms = one(twice=[f])
    tot += ms - b
    print("  %-16s %.4f ms/batch  (+%.4f)" % (f, ms, ms - b))
print("  sum of the marginal costs %.4f ms (the step: %.4f)" % (tot, b))
ms = one(twice=["geometry_front", "geometry_tables"])
print("  geometry (both launches) twice: %.4f (+%.4f)" % (ms, ms - b))
print("family SKIPPED (results wrong; only families that produce no index table):")
for f in ("cost_volume", "sa_scale", "pointwise", "patch_cost", "gru"):
    ms = one(skip=[f])
    print("  without %-14s %.4f ms/batch  (-%.4f)" % (f, ms, b - ms))
print("  without cost_volume+sa_scale %.4f" % one(skip=["cost_volume", "sa_scale"]))
print("  without cost_volume+sa_scale+pointwise %.4f" % one(skip=["cost_volume", "sa_scale", "pointwise"]))
