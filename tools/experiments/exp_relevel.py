"""Experiment: level-1 tie flags on the bench clouds and the cost of rtk_fps_relevel (copy path vs full path)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ratrack_amd import _lib, fused, synth
from ratrack_amd.benchutil import _time
dev = "cuda"
for case, B, N in [(1000, 64, 256), (1001, 64, 256), (1005, 32, 1024)]:
    d = synth.make_frame_pairs(B, N, case)
    xyz = torch.cat([torch.from_numpy(d["pc1"]), torch.from_numpy(d["pc2"])]).permute(0, 2, 1).contiguous().to(dev)
    S_ = 2 * B
    i32 = lambda *s: torch.zeros(*s, dtype=torch.int32, device=dev)
    idx, nx, cnt, tie = i32(S_, 512), torch.empty(S_, 512, 3, device=dev), i32(S_), i32(S_)
    snap, first, scratch = torch.empty(S_, N, device=dev), i32(S_), torch.empty(S_, 1024, device=dev)
    st = fused._stream
    f = lambda: _lib.call("rtk_fps_centroids", S_, N, 512, xyz.data_ptr(), idx.data_ptr(), nx.data_ptr(), cnt.data_ptr(), tie.data_ptr(), None, snap.data_ptr(), first.data_ptr(), st())
    t_fps = _time(f, 20)
    idx23, nx23, cnt23 = i32(2, S_, 512), torch.empty(2, S_, 512, 3, device=dev), i32(2, S_)
    g = lambda: _lib.call("rtk_fps_relevel", S_, 512, 2, nx.data_ptr(), cnt.data_ptr(), tie.data_ptr(), idx23.data_ptr(), nx23.data_ptr(), cnt23.data_ptr(), idx.data_ptr(), snap.data_ptr(), N, first.data_ptr(), scratch.data_ptr(), st())
    t_rel = _time(g, 20)
    ties = int((tie > 0).sum())
    tie0 = tie.clone(); tie.zero_()
    t_copy = _time(g, 20)
    tie.fill_(1)
    t_full = _time(g, 20)
    print("case %d B=%d N=%d: ties %d/%d nuniq min %d  fps %.1f us  relevel %.1f us  (all-copy %.1f us, all-full %.1f us)" % (
        case, B, N, ties, S_, int(cnt.min()), t_fps * 1e3, t_rel * 1e3, t_copy * 1e3, t_full * 1e3))
    if ties:
        b = int(torch.nonzero(tie0)[0])
        x = xyz[b].cpu()
        u, c = torch.unique(x, dim=0, return_counts=True)
        print("   sample %d: %d distinct of %d points" % (b, u.shape[0], x.shape[0]))
