import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_varn_train_gpu as T
from _util import load_case, inputs_of
from ratrack_amd import loss as L, train_ops

case = load_case("train_b8_n256")
DEV = "cuda"

def run(dedup):
    net = T.make_net()
    net._dedup_train = dedup
    rec = {}
    fp = net.fd_layer.fp
    orig_fwd = fp.forward
    o_bn, o_lin = train_ops.pw_bn_relu, train_ops.pw_linear
    state = {"on": False, "i": 0}
    def hook(name):
        def h(g): rec["d" + name] = g.detach().clone()
        return h
    def w_bn(srcs, *a, **k):
        out = o_bn(srcs, *a, **k)
        if state["on"]:
            n = "y%d" % state["i"]; state["i"] += 1
            rec[n] = out.detach().clone(); out.register_hook(hook(n))
        return out
    def w_lin(srcs, *a, **k):
        out = o_lin(srcs, *a, **k)
        if state["on"]:
            n = "lin%d" % state["i"]; state["i"] += 1
            rec[n] = out.detach().clone(); out.register_hook(hook(n))
        return out
    train_ops.pw_bn_relu, train_ops.pw_linear = w_bn, w_lin
    def fwd(x, *a, **k):
        x.retain_grad(); rec["x"] = x
        state["on"] = True
        out = orig_fwd(x, *a, **k)
        state["on"] = False
        return out
    fp.forward = fwd
    pc1, pc2, f1, f2 = inputs_of(case, DEV)
    gt, gcls = torch.from_numpy(case["in_gt_warp"]).to(DEV), torch.from_numpy(case["in_gt_cls"]).to(DEV)
    flow, h, cls, cor, pf1, pf2, prop = net.backbone(pc1, pc2, f1, f2, None)
    flow.retain_grad()
    total, items = L.backbone_loss(pc1 + flow, cls, gt, gcls, pretrain=False)
    total.backward()
    train_ops.pw_bn_relu, train_ops.pw_linear = o_bn, o_lin
    rec["dflow"] = flow.grad; rec["dx"] = rec["x"].grad; rec["x"] = rec["x"].detach()
    return rec

a = run(True); b = run(False); a2 = run(True)
def err(u, v): return float((u.double() - v.double()).abs().max() / v.double().abs().max())
for k in sorted(a):
    print("%-8s train-vs-module %.2e   train-vs-train(rerun) %.2e" % (k, err(a[k], b[k]), err(a[k], a2[k])), tuple(a[k].shape))
m_a, m_b = a["y0"] > 0, b["y0"] > 0
diff = (m_a != m_b)
print("ReLU mask of flow-head layer 0: %d of %d elements differ between the two runs" % (int(diff.sum()), diff.numel()))
idx = diff.nonzero()
for i in idx[:8].tolist():
    bb, c, n = i
    print("  element (b=%d, c=%d, n=%d): y0 train %.3e module %.3e ; dy0 there %.3e ; max|dy0| %.3e" % (bb, c, n, float(a["y0"][bb, c, n]), float(b["y0"][bb, c, n]),
          float(a["dy0"][bb, c, n]), float(a["dy0"].abs().max())))
e = (a["dx"] - b["dx"]).abs()
pos = e.amax(1)            # (B, N)
top = pos.flatten().topk(5)
print("dx error by position (top 5 of %d): " % pos.numel(), [(int(i) // 256, int(i) % 256, "%.2e" % float(v)) for v, i in zip(top.values, top.indices)], " median %.2e" % float(pos.median()))
