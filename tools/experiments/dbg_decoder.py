import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_varn_train_gpu as T
from _util import load_case, inputs_of
from ratrack_amd import loss as L

case = load_case("train_b8_n256")
DEV = "cuda"

def run(dedup, dtype=torch.float32):
    net = T.make_net()
    net._dedup_train = dedup
    if dtype == torch.float64:
        net = net.double()
    rec = {}
    fp = net.fd_layer.fp
    orig = fp.forward
    def fwd(x, *a, **k):
        x.retain_grad(); rec["x"] = x
        out = orig(x, *a, **k)
        return out
    fp.forward = fwd
    mse = net.fd_layer.mse
    pc1, pc2, f1, f2 = (t.to(dtype) for t in inputs_of(case, DEV))
    gt, gcls = torch.from_numpy(case["in_gt_warp"]).to(DEV).to(dtype), torch.from_numpy(case["in_gt_cls"]).to(DEV)
    flow, h, cls, cor, pf1, pf2, prop = net.backbone(pc1, pc2, f1, f2, None)
    prop.retain_grad(); cor.retain_grad()
    total, items = L.backbone_loss(pc1 + flow, cls, gt, gcls, pretrain=False)
    total.backward()
    g = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
    return dict(flow=flow.detach(), x=rec["x"].detach(), dx=rec["x"].grad, dprop=prop.grad, dcor=cor.grad, prop=prop.detach(), h=h.detach()), g

a, ga = run(True)
b, gb = run(False)
try:
    c, gc = run(False, torch.float64)
except Exception as e:
    print("f64 module path failed:", repr(e)[:200]); c, gc = None, None
def err(u, v): return float((u.double() - v.double()).abs().max() / v.double().abs().max())
for k in a:
    print("%-8s train-vs-module %.2e" % (k, err(a[k], b[k])), "" if c is None else " train-vs-f64 %.2e  module-vs-f64 %.2e" % (err(a[k], c[k]), err(b[k], c[k])))
xs = a["dx"]; 
print("dx[:, :128] (prop part) err vs module %.2e ; dx[:,128:] (gfeat part) %.2e" % (err(a["dx"][:, :128], b["dx"][:, :128]), err(a["dx"][:, 128:], b["dx"][:, 128:])))
for k in ["fd_layer.fp.sf_mlp.0.1.bias", "fd_layer.fp.sf_mlp.0.1.weight", "fd_layer.fp.sf_mlp.0.0.weight", "fd_layer.torchGRU.weight_ih_l2", "fd_layer.mse.fp1.mlp.layer0.conv.weight", "fd_layer.mse.sa3.mlps.1.layer0.bn.bn.weight"]:
    print("%-48s train-vs-module %.2e" % (k, err(ga[k], gb[k])), "" if gc is None else " train-vs-f64 %.2e module-vs-f64 %.2e" % (err(ga[k], gc[k]), err(gb[k], gc[k])))
