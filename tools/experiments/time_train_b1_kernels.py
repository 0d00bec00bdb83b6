"""Per entry point: one eager B = 1 train step on a real frame pair vs a synthetic pair of the same padded size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from _util import load_case, inputs_of, reference_state_dict
from ratrack_amd import _lib, synth
from ratrack_amd.track4d import Args, Track4D
from ratrack_amd.train import Trainer
DEV = "cuda"
pad = lambda t, n: torch.cat([t, t[..., :1].expand(*t.shape[:-1], n - t.shape[-1])], dim=-1).contiguous()
def batch_real():
    s = load_case(os.environ.get("RTK_CASE", "real_1201_549"))
    pc1, pc2, f1, f2 = inputs_of(s, DEV)
    gt, gc = torch.from_numpy(s["in_gt_warp"]).to(DEV), torch.from_numpy(s["in_gt_cls"]).to(DEV)
    nv = torch.tensor([[pc1.shape[2]], [pc2.shape[2]]], dtype=torch.int32, device=DEV)
    return (pad(pc1, 384), pad(pc2, 384), pad(f1, 384), pad(f2, 384), pad(gt, 384), pad(gc, 384)), nv
def batch_syn():
    d = synth.make_frame_pairs(1, 384, 5)
    t = {k: torch.from_numpy(v).to(DEV) for k, v in d.items()}
    return (t["pc1"], t["pc2"], t["feature1"], t["feature2"], t["gt_warp"], t["gt_cls"]), torch.tensor([[322], [352]], dtype=torch.int32, device=DEV)
res = {}
for name, mk in (("real", batch_real), ("synthetic", batch_syn)):
    net = Track4D(Args()).to(DEV)
    net.load_state_dict(reference_state_dict(DEV), strict=True)
    tr = Trainer(net, lr=1e-4, graph=False)
    b, nv = mk()
    for _ in range(3): tr.step(*b, n_valid=nv)
    torch.cuda.synchronize()
    _lib.TIMING = tm = []
    tr.step(*b, n_valid=nv)
    _lib.TIMING = None
    torch.cuda.synchronize()
    acc = {}
    for nm, e0, e1 in tm:
        a = acc.setdefault(nm, [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e3
    res[name] = acc
print("%-36s %6s %10s %10s" % ("entry point", "calls", "real us", "synth us"))
for nm in sorted(res["real"], key=lambda k: -res["real"][k][1]):
    r, s = res["real"][nm], res["synthetic"].get(nm, [0, 0.0])
    print("%-36s %6d %10.1f %10.1f" % (nm, r[0], r[1], s[1]))
print("total (our entry points only): real %.0f us, synthetic %.0f us" % (sum(v[1] for v in res["real"].values()), sum(v[1] for v in res["synthetic"].values())))
