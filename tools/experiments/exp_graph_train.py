#!/usr/bin/env python3
"""Debug helper: graphed Trainer with/without a preceding eager trainer in the same process."""
import sys, os, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("FH"): faulthandler.enable()
import torch
from ratrack_amd import synth
from ratrack_amd.track4d import Track4D, Args
from ratrack_amd.train import Trainer
DEV = "cuda"
B, N = 2, 256
mode = sys.argv[1]
batches = []
for i in range(6):
    d = synth.make_frame_pairs(B, N, 20 + i)
    batches.append({k: torch.from_numpy(v).to(DEV) for k, v in d.items()})
seq = {"graph_only": (True,), "both": (False, True), "same_batch": (True,)}[mode]
for graph in seq:
    net = Track4D(Args()).to(DEV)
    synth.fill_state_dict(net.state_dict())
    tr = Trainer(net, graph=graph, lr=float(os.environ.get('LR', '1e-3')))
    h = torch.zeros(5, B, 128, device=DEV)
    for i, t in enumerate(batches):
        if mode == "same_batch":
            t = batches[0]
        cs = float(sum(p.detach().double().abs().sum() for p in net.parameters()))
        st = [float(v["step"]) for v in list(tr.opt.state.values())[:1]]
        print("  params checksum before step", i, "%.10f" % cs, st, flush=True)
        items, _ = tr.step(t["pc1"], t["pc2"], t["feature1"], t["feature2"], t["gt_warp"], t["gt_cls"], h)
        if not os.environ.get("NOSYNC"):
            print(mode, graph, i, float(items["Loss"].detach()), flush=True)
        else:
            print(mode, graph, i, flush=True)
print("done", mode)
