import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import copy
import torch
from ratrack_amd.model_utils import FlowPredictor

torch.manual_seed(0)
dev = "cuda"
B, N = 8, 256
m = FlowPredictor(256, [128, 64, 32]).to(dev).train()
ref = copy.deepcopy(m).double()
x = torch.randn(B, 256, N, device=dev)
ct = torch.randn(B, 3, N, device=dev)
xb = x.double().requires_grad_(True)
yb = ref._modules["conv2"](torch.nn.Sequential(*ref.sf_mlp)(xb.unsqueeze(3))).squeeze(3)
yb.backward(ct.double())
def pollute(val):
    bufs = [torch.full((1 << k,), val, device=dev) for k in range(8, 25)]
    del bufs
for trial, val in enumerate([float("nan"), 1e3, float("nan"), -7.0, 0.0]):
    pollute(val)
    for p in m.parameters():
        p.grad = None
    xa = x.clone().requires_grad_(True)
    ya = m(xa)
    pollute(val)
    ya.backward(ct)
    errs = {k: float((p.grad.double() - q.grad).abs().max() / q.grad.abs().max()) for (k, p), (_, q) in zip(m.named_parameters(), ref.named_parameters())}
    errs["dx"] = float((xa.grad.double() - xb.grad).abs().max() / xb.grad.abs().max())
    print("pollute", val, " ".join("%s %.1e" % (k.replace("sf_mlp.", "L"), v) for k, v in errs.items()))
