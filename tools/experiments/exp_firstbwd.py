"""Experiment: first-layer backward, gather form (inverse index + rtk_sa_first_layer_bwd) vs LDS-atomic scatter + bmm, train-step shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ratrack_amd import _lib, synth, train_ops as T
from ratrack_amd.train_path import TrainGeometry
from ratrack_amd.benchutil import _time
dev = "cuda"
B = 64
d = synth.make_frame_pairs(B, 256, 1000)
xyz = torch.cat([torch.from_numpy(d["pc1"]), torch.from_numpy(d["pc2"])]).permute(0, 2, 1).contiguous().to(dev)
tg = TrainGeometry(xyz, 512)
st = torch.cuda.current_stream().cuda_stream
C1s = [[16, 16], [32, 32], [64, 64]]
for lvl in range(3):
    for s in range(2):
        idx, dxyz = tg.ball[lvl][s], tg.dxyz[lvl][s]
        S_, rows, ns = idx.shape
        n_src = 256
        C = C1s[lvl][s]
        off, inv = tg.inv[lvl][s]
        t_build = _time(lambda: _lib.call("rtk_group_inverse_index", S_, n_src, rows * ns, idx.data_ptr(), off.data_ptr(), inv.data_ptr(), st), 10)
        dz = torch.randn(S_, C, rows, ns, device=dev)
        dproj = torch.empty(S_, C, n_src, device=dev)
        dwx = torch.zeros(C, 3, device=dev)
        t_new = _time(lambda: _lib.call("rtk_sa_first_layer_bwd", S_, C, rows, ns, n_src, dz.data_ptr(), dxyz.data_ptr(), off.data_ptr(),
                                        inv.data_ptr(), dproj.data_ptr(), dwx.data_ptr(), 3, torch.empty(S_ * C * 3, device=dz.device).data_ptr(), st), 10)
        t_old = _time(lambda: _lib.call("rtk_group_points_grad_set", S_, C, n_src, rows, ns, dz.data_ptr(), idx.data_ptr(), dproj.data_ptr(), st), 10)
        t_bmm = _time(lambda: torch.bmm(dz.view(S_, C, -1), dxyz.view(S_, 3, -1).transpose(1, 2)).sum(0), 10)
        mb = dz.numel() * 4 / 1e6
        lens = (off[:, 1:] - off[:, :-1]).max().item()
        print("lvl %d scale %d: C=%d ns=%d dz %.0f MB  longest list %d | build %.1f us | gather+dwx %.1f us (%.2f TB/s) | old scatter %.1f us + bmm/sum %.1f us" % (
            lvl, s, C, ns, mb, lens, t_build * 1e3, t_new * 1e3, mb / t_new / 1e3 / 1e3 * 1e3, t_old * 1e3, t_bmm * 1e3))
