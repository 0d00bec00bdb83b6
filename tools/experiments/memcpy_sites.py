"""Which calls of one eager train step become device-to-device memcpys (the __amd_rocclr_copyBuffer nodes of the captured step)?"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile
from ratrack_amd import synth
from ratrack_amd.track4d import Args, Track4D
from ratrack_amd.train import Trainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = "cuda"
net = Track4D(Args()).to(dev); synth.fill_state_dict(net.state_dict())
d = synth.make_frame_pairs(B, 256, 1000)
t = {k: torch.from_numpy(v).to(dev) for k, v in d.items()}
h = torch.zeros(5, B, 128, device=dev)
tr = Trainer(net, graph=False)
step = lambda: tr.step(t["pc1"], t["pc2"], t["feature1"], t["feature2"], t["gt_warp"], t["gt_cls"], h)
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
ev = prof.events()
mem = [e for e in ev if e.device_type.name != "CPU" and ("emcpy" in e.name or "copyBuffer" in e.name or "emset" in e.name)]
print("device memcpy/memset events:", len(mem))
byname = collections.Counter(e.name for e in mem)
print(dict(byname))
# CPU-side ops that launched them: match by correlation via e.kernels of cpu events is not populated for memcpy; use time containment
cpu = [e for e in ev if e.device_type.name == "CPU"]
sites = collections.Counter()
for e in cpu:
    if not e.name.startswith("aten::") and "Memcpy" not in e.name and "hipMemcpy" not in e.name:
        continue
    if "hipMemcpy" in e.name or "Memcpy" in e.name:
        # find the innermost enclosing aten op
        par = e.cpu_parent
        chain = []
        while par is not None:
            chain.append(par.name)
            par = par.cpu_parent
        st = [s for s in (e.stack or []) if "ratrack_amd" in s]
        sites[(e.name, " < ".join(chain[:4]), st[0].split("/")[-1] if st else "")] += 1
for k, v in sites.most_common(40):
    print(v, k)
