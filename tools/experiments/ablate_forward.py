"""Which kernel family bounds the PIPELINED forward?  Throughput at B=64 with four graphs in flight when one family's launches are
skipped (results are wrong; only the timing matters; index-producing geometry kernels are never skipped -- garbage indices fault).  python tools/experiments/ablate_forward.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ratrack_amd import _lib, fused, synth
from ratrack_amd.track4d import Args, Track4D

dev = torch.device("cuda")
net = Track4D(Args()).to(dev).eval()
synth.fill_state_dict(net.state_dict())
d = synth.make_frame_pairs(64, 256, 1000)
t = [torch.from_numpy(d[k]).to(dev) for k in ("pc1", "pc2", "feature1", "feature2")] + [torch.zeros(5, 64, 128, device=dev)]
FAM = {"cost_volume": ["rtk_cost_volume_split", "rtk_cost_volume_split_shared"], "sa_scale": ["rtk_sa_scale", "rtk_sa_scale_split"], "pointwise": ["rtk_pointwise_mlp"],
       "patch+gru": ["rtk_patch_cost", "rtk_gru_step"], "fps": ["rtk_fps_centroids", "rtk_fps_relevel"]}
orig = _lib.call
def run(skip):
    names = set(sum((FAM[f] for f in skip), []))
    _lib.call = lambda name, *a: 0 if name in names else orig(name, *a)
    fused._lib.call = _lib.call
    with torch.no_grad():
        net.invalidate_fused()
        net.backbone(*t)
        pipe = fused.GraphPipeline(net._fused, tuple(t), depth=4)
        for _ in range(300):
            pipe.submit(*t)
        pipe.drain(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(1000):
            pipe.submit(*t)
        pipe.drain(); torch.cuda.synchronize()
        el = time.perf_counter() - t0
    _lib.call = orig; fused._lib.call = orig
    return el / 1000 * 1e3
base = run([])
print("baseline %.3f ms/batch" % base)
for f in FAM:
    ms = run([f])
    print("without %-14s %.3f ms/batch  (-%.3f)" % (f, ms, base - ms))
print("without cost_volume+sa_scale %.3f" % run(["cost_volume", "sa_scale"]))
print("without cost_volume+sa_scale+pointwise %.3f" % run(["cost_volume", "sa_scale", "pointwise"]))
