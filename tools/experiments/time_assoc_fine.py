import sys, os, time
sys.path.insert(0, os.getcwd())
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
objs_hist = []
with torch.no_grad():
    for i in range(6):
        d = synth.make_frame_pairs(1, 256, 300 + i)
        t = {k: torch.from_numpy(v).to(dev) for k, v in d.items() if k != "gt_cls"}
        out = net.backbone(t["pc1"], t["pc2"], t["feature1"], t["feature2"], None)
        pf = torch.cat((t["pc1"] + out[0], t["pc1"], out[0], t["feature1"], out[6]), dim=1)
        sel = pf[:, :, (out[2] > 0.5).squeeze(0)]
        objs_hist.append(A.cluster_objects(sel, eps=1.5, min_samples=2))
    prev = {i: o for i, o in enumerate(objs_hist[-2])}
    curr = objs_hist[-1]
    for rep in range(3):
        cache = {}
        t0 = T(); aff_list, aff_mat, m, n = A.affinity_matrix(net.affinity, curr, prev, cache); t1 = T()
        sc = A._log_optimal_transport_hip(aff_mat, 0.9, 500); t2 = T()
        idx = A.sinkhorn_assignment(aff_mat); t3 = T()
        host = idx[0].tolist(); ah = aff_mat[0].cpu(); t4 = T()
        ds = [A.object_descriptor(o, 128) for o in curr]; t5 = T()
        y = net.affinity.affinity(torch.randn(m * n, 141, device=dev)); t6 = T()
        print("m=%d n=%d  affinity_matrix %.2f  sinkhorn kernel %.2f  assignment(total) %.2f  d2h %.2f  descriptors(n) %.2f  mlp %.2f ms" % (m, n, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3, (t5-t4)*1e3, (t6-t5)*1e3))
