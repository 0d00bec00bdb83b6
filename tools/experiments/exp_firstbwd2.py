"""rtk_sa_first_layer_bwd at the train-step shapes (ball-query tables of the synthetic clouds): time and GB/s per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ratrack_amd import _lib, synth, train_ops as T, pointnet2_utils as PU
dev = "cuda"
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
d = synth.make_frame_pairs(B, 256, 1000)
xyz = torch.cat([torch.from_numpy(d["pc1"]), torch.from_numpy(d["pc2"])]).permute(0, 2, 1).contiguous().to(dev)      # (2B, 256, 3)
S_, n_src = xyz.shape[0], 256
tot = 0.0
for C, radius, ns in [(16, 2.0, 4), (16, 4.0, 8), (32, 4.0, 8), (32, 8.0, 16), (64, 8.0, 16), (64, 16.0, 32)]:
    rows = 256
    idx = PU.ball_query(radius, ns, xyz, xyz).int().contiguous()          # (S, rows, ns)
    P = rows * ns
    off = torch.empty(S_, n_src + 1, dtype=torch.int32, device=dev); inv = torch.empty(S_, P, dtype=torch.int16, device=dev)
    _lib.call("rtk_group_inverse_index", S_, n_src, P, idx.data_ptr(), off.data_ptr(), inv.data_ptr(), st)
    g = torch.Generator(dev).manual_seed(ns)
    dz = torch.randn(S_, C, rows, ns, device=dev, generator=g)
    dxyz = torch.randn(S_, 3, rows, ns, device=dev, generator=g)
    dproj = torch.empty(S_, C, n_src, device=dev)
    dwx = torch.zeros(C, 3, device=dev)
    dwx_ws = torch.empty(S_ * C * 3, device=dz.device)
    fn = lambda: _lib.call("rtk_sa_first_layer_bwd", S_, C, rows, ns, n_src, dz.data_ptr(), dxyz.data_ptr(), off.data_ptr(), inv.data_ptr(),
                           dproj.data_ptr(), dwx.data_ptr(), 3, dwx_ws.data_ptr(), st)
    dwx.zero_(); fn(); torch.cuda.synchronize()
    ref = torch.zeros(S_, C, n_src, device=dev, dtype=torch.float64)
    ref.scatter_add_(2, idx.long().view(S_, 1, P).expand(S_, C, P), dz.double().view(S_, C, P))
    refw = torch.einsum("scp,sdp->cd", dz.double().view(S_, C, P), dxyz.double().view(S_, 3, P))
    e1 = float((dproj - ref).abs().max() / ref.abs().max()); e2 = float((dwx - refw).abs().max() / refw.abs().max())
    us = timeit(fn)
    tot += us
    print("C=%2d ns=%2d dz %5.0f MB  longest list %4d | %7.1f us  %.2f TB/s | err dproj %.2g dwx %.2g" % (
        C, ns, dz.numel() * 4 / 1e6, int((off[:, 1:] - off[:, :-1]).max()), us, dz.numel() * 4 / us / 1e6, e1, e2))
print("sum %.1f us (encoder-sized launches only)" % tot)
