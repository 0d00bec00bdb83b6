#!/usr/bin/env python3
"""Ad-hoc timing of group_points_grad (LDS scatter) and sa_first_layer at the largest SA shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ratrack_amd import _lib, train_ops, synth, fused  # noqa
from ratrack_amd.train_path import TrainGeometry
dev = "cuda"
d = synth.make_frame_pairs(64, 256, 0)
xyz = torch.cat([torch.from_numpy(d["pc1"]), torch.from_numpy(d["pc2"])], 0).to(dev).permute(0, 2, 1).contiguous()
tg = TrainGeometry(xyz, 512)
st = torch.cuda.current_stream().cuda_stream
def t(name, fn, bytes_):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("%-44s %7.1f us  %5.2f TB/s" % (name, ms * 1e3, bytes_ / ms / 1e9))
for lvl, s, C in [(2, 1, 64), (2, 0, 64), (1, 1, 32), (0, 1, 16)]:
    idx = tg.ball[lvl][s]; S_, U, ns = idx.shape
    n_src = 256
    dz = torch.randn(S_, C, U, ns, device=dev); out = torch.empty(S_, C, n_src, device=dev)
    t("group_grad lvl%d s%d C=%d ns=%d" % (lvl, s, C, ns), lambda: _lib.call("rtk_group_points_grad_set", S_, C, n_src, U, ns, dz.data_ptr(), idx.data_ptr(), out.data_ptr(), st), dz.numel() * 4)
    ridx = torch.randint(0, n_src, idx.shape, device=dev, dtype=torch.int32)
    t("   same with random indices", lambda: _lib.call("rtk_group_points_grad_set", S_, C, n_src, U, ns, dz.data_ptr(), ridx.data_ptr(), out.data_ptr(), st), dz.numel() * 4)
