"""Which Python lines of the training path call framework (aten) operators that launch kernels, in one eager train step?
TorchDispatchMode logs every aten call with the innermost ratrack_amd frame; pure view operators are skipped."""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from ratrack_amd import synth
from ratrack_amd.track4d import Args, Track4D
from ratrack_amd.train import Trainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
VIEWS = {"view", "_unsafe_view", "reshape", "slice", "select", "transpose", "unbind", "expand", "as_strided", "detach", "alias", "t", "permute",
         "unsqueeze", "squeeze", "split", "split_with_sizes", "narrow", "unfold", "view_as", "expand_as", "chunk", "unflatten", "flatten", "movedim",
         "empty", "empty_like", "empty_strided", "new_empty", "new_empty_strided", "sym_size", "sym_stride", "is_contiguous", "stride", "size", "numel",
         "_local_scalar_dense", "lift_fresh", "is_same_size", "result_type", "set_", "storage_offset", "dim", "record_stream"}
sites = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__
        if name not in VIEWS:
            on_gpu = any(isinstance(a, torch.Tensor) and a.is_cuda for a in torch.utils._pytree.tree_leaves((args, kwargs)))
            st = [f for f in traceback.extract_stack() if "/ratrack_amd/" in f.filename]
            where = "%s:%d %s" % (os.path.basename(st[-1].filename), st[-1].lineno, st[-1].name) if st else "(outside)"
            shape = next((tuple(a.shape) for a in torch.utils._pytree.tree_leaves((args, kwargs)) if isinstance(a, torch.Tensor)), ())
            if on_gpu or name in ("zeros", "ones", "full", "arange", "zeros_like", "ones_like", "full_like"):
                sites[(name, where, shape)] += 1
        return func(*args, **(kwargs or {}))


dev = "cuda"
net = Track4D(Args()).to(dev); synth.fill_state_dict(net.state_dict())
d = synth.make_frame_pairs(B, 256, 1000)
t = {k: torch.from_numpy(v).to(dev) for k, v in d.items()}
h = torch.zeros(5, B, 128, device=dev)
tr = Trainer(net, graph=False)
step = lambda: tr.step(t["pc1"], t["pc2"], t["feature1"], t["feature2"], t["gt_warp"], t["gt_cls"], h)
for _ in range(2): step()
torch.cuda.synchronize()
with Log():
    step()
torch.cuda.synchronize()
byop = collections.Counter()
for (name, where, shape), n in sites.items():
    byop[name] += n
print("aten calls by operator:", dict(byop.most_common()))
print("%-26s %-48s %-28s %s" % ("operator", "innermost ratrack_amd frame", "first tensor", "calls"))
for (name, where, shape), n in sorted(sites.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print("%-26s %-48s %-28s %d" % (name, where, str(shape), n))
print("total", sum(sites.values()))
