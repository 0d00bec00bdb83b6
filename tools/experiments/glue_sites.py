"""Which Python lines launch the framework (at::native / memcpy) kernels of one eager train step at B = 64?"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile
from ratrack_amd import synth
from ratrack_amd.track4d import Args, Track4D
from ratrack_amd.train import Trainer
dev = "cuda"
net = Track4D(Args()).to(dev); synth.fill_state_dict(net.state_dict())
d = synth.make_frame_pairs(64, 256, 1000)
t = {k: torch.from_numpy(v).to(dev) for k, v in d.items()}
h = torch.zeros(5, 64, 128, device=dev)
tr = Trainer(net, graph=False)
step = lambda: tr.step(t["pc1"], t["pc2"], t["feature1"], t["feature2"], t["gt_warp"], t["gt_cls"], h)
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
sites = collections.Counter(); times = collections.Counter()
for e in prof.events():
    if e.device_type.name != "CPU" or not e.name.startswith("aten::"):
        continue
    kern = [k for k in e.kernels] if hasattr(e, "kernels") else []
    if not kern:
        continue
    if e.cpu_parent is not None and e.cpu_parent.name.startswith("aten::"):
        continue                      # count the outermost aten op only
    st = [s for s in (e.stack or []) if "ratrack_amd" in s or "bench" in s]
    site = st[0].split("/")[-1] if st else ("autograd engine" if not e.stack else e.stack[0][-60:])
    key = "%-28s %s" % (e.name, site)
    sites[key] += len(kern); times[key] += sum(k.duration for k in kern)
print("%-95s %6s %9s" % ("aten op @ first ratrack_amd frame", "kernels", "us"))
for k, n in sorted(sites.items(), key=lambda kv: -times[kv[0]])[:45]:
    print("%-95s %6d %9.1f" % (k[:95], n, times[k]))
print("total framework kernels %d, %.0f us" % (sum(sites.values()), sum(times.values())))
