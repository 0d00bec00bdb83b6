#!/usr/bin/env python3
"""Where the ~6 ms of detect_and_associate go (B = 1): selection, clustering, descriptors + affinity, Sinkhorn, bookkeeping."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ratrack_amd import synth, association as A
from ratrack_amd.track4d import Track4D, Args
dev = "cuda"
net = Track4D(Args()).to(dev).eval()
synth.fill_state_dict(net.state_dict())
with torch.no_grad():
    net.fd_layer.cp.linear.bias.add_(0.09)
def T():
    torch.cuda.synchronize(); return time.perf_counter()
prev, h = None, None
acc = {}
with torch.no_grad():
    for i in range(12):
        d = synth.make_frame_pairs(1, 256, 300 + i)
        t = {k: torch.from_numpy(v).to(dev) for k, v in d.items() if k != "gt_cls"}
        out = net.backbone(t["pc1"], t["pc2"], t["feature1"], t["feature2"], h)
        flow, cls, prop = out[0], out[2], out[6]
        t0 = T()
        pf = torch.cat((t["pc1"] + flow, t["pc1"], flow, t["feature1"], prop), dim=1)
        mask = (cls > 0.5).squeeze(0)
        sel = pf[:, :, mask]
        t1 = T()
        objs = A.cluster_objects(sel, eps=1.5, min_samples=2)
        t2 = T()
        res = net.associator(objs, prev or dict())
        t3 = T()
        prev, h = res[4], out[1]
        if i >= 3:
            for k, v in (("select", t1 - t0), ("cluster", t2 - t1), ("associate", t3 - t2)):
                acc[k] = acc.get(k, 0) + v / 9
print({k: "%.2f ms" % (v * 1e3) for k, v in acc.items()})
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
with torch.no_grad():
    objs = A.cluster_objects(sel, eps=1.5, min_samples=2); res = net.associator(objs, prev)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
