import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ratrack_amd import synth
from ratrack_amd.track4d import Args, Track4D
from ratrack_amd.train import Trainer
for B, N in ((8, 1024), (32, 1024)):
    net = Track4D(Args()).to("cuda"); synth.fill_state_dict(net.state_dict())
    d = synth.make_frame_pairs(B, N, 77)
    t = {k: torch.from_numpy(v).cuda() for k, v in d.items()}
    h = torch.zeros(5, B, 128, device="cuda")
    tr = Trainer(net, graph=True, graph_warmup=2)
    for i in range(5):
        items, h2 = tr.step(t["pc1"], t["pc2"], t["feature1"], t["feature2"], t["gt_warp"], t["gt_cls"], h)
    torch.cuda.synchronize()
    g = [p.grad for p in net.parameters() if p.grad is not None]
    print(B, N, {k: float(v) for k, v in items.items()}, all(torch.isfinite(x).all().item() for x in g), len(g))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(10):
        tr.step(t["pc1"], t["pc2"], t["feature1"], t["feature2"], t["gt_warp"], t["gt_cls"], h)
    e1.record(); torch.cuda.synchronize()
    print("  ms/step", e0.elapsed_time(e1) / 10)
