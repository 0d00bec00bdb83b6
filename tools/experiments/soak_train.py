"""Soak: 400 captured train steps over 8 synthetic batches (B=16): the loss falls monotonically-ish and every parameter stays finite."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ratrack_amd import synth
from ratrack_amd.track4d import Track4D, Args
from ratrack_amd.train import Trainer
dev = "cuda"
torch.manual_seed(0)
net = Track4D(Args()).to(dev); synth.fill_state_dict(net.state_dict())
tr = Trainer(net, graph=True, lr=1e-3)
losses = []
batches = []
for i in range(8):
    d = synth.make_frame_pairs(16, 256, 3000 + i)
    batches.append({k: torch.from_numpy(v).to(dev) for k, v in d.items()})
h = torch.zeros(5, 16, 128, device=dev)
for it in range(400):
    t = batches[it % 8]
    items, _ = tr.step(t["pc1"], t["pc2"], t["feature1"], t["feature2"], t["gt_warp"], t["gt_cls"], h)
    if it % 40 == 0 or it == 399:
        losses.append(float(items["Loss"]))
print("losses", [round(x, 4) for x in losses])
assert all(x == x and x < 1e6 for x in losses), "non-finite loss"
assert losses[-1] < losses[0], "loss did not decrease"
bad = [k for k, p in net.named_parameters() if not torch.isfinite(p).all()]
assert not bad, bad
print("soak ok")
