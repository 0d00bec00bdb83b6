#!/usr/bin/env python3
"""Which framework ops still launch kernels in the train step: per aten / autograd op, calls per step and device time
(torch.profiler, eager steps after warm-up).  Ops that are ours (rtk_* launches inside autograd Functions) show up under the
Function's name."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from ratrack_amd import synth
from ratrack_amd.track4d import Track4D, Args
from ratrack_amd.train import Trainer
dev = "cuda"; B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
net = Track4D(Args()).to(dev); synth.fill_state_dict(net.state_dict())
d = synth.make_frame_pairs(B, 256, 0); t = {k: torch.from_numpy(v).to(dev) for k, v in d.items()}
h = torch.zeros(5, B, 128, device=dev); tr = Trainer(net)
step = lambda: tr.step(t["pc1"], t["pc2"], t["feature1"], t["feature2"], t["gt_warp"], t["gt_cls"], h)
for _ in range(4): step()
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(N): step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages():
    dt = getattr(e, "self_device_time_total", None)
    if dt is None:
        dt = e.self_cuda_time_total
    if dt > 0:
        rows.append((dt / N, e.count / N, e.key))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows if not r[2].startswith(("void ", "Cijk", "(anonymous", "__amd", "Memcpy", "Memset")) and "kernel" not in r[2].lower())
print("device us/step by op (self time), B=%d" % B)
for dt, cnt, key in rows:
    print("%9.1f us %7.1f calls  %s" % (dt, cnt, key[:110]))
