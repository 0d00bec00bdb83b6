"""rtk_scatter_add_rows (256 channels) at the train-step shape: the p2 gradient of the cost volume."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ratrack_amd import _lib, train_ops as T
from ratrack_amd.benchutil import time_graph
dev = "cuda"; B, n = 64, 256
from ratrack_amd import synth
from ratrack_amd.model_utils import knn_point
d = synth.make_frame_pairs(B, n, 1000)
x1 = torch.from_numpy(d["pc1"]).permute(0, 2, 1).contiguous().to(dev); x2 = torch.from_numpy(d["pc2"]).permute(0, 2, 1).contiguous().to(dev)
knn = knn_point(16, x2, x1).contiguous()
src = torch.randn(B * n * 16, 256, device=dev); dst = torch.empty(B * n, 256, device=dev)
ms = time_graph(lambda: _lib.call("rtk_scatter_add_rows", B, n * 16, n, 256, knn.data_ptr(), src.data_ptr(), dst.data_ptr(), T._stream()), 10)
ref = torch.zeros(B, n, 256, device=dev).index_add_(1, knn.view(B, -1)[0], src.view(B, -1, 256)[0:1].expand(1, -1, -1)[0].unsqueeze(0).expand(B, -1, -1)) if False else None
print("scatter_add_rows: %.1f us (%.2f TB/s of the %.0f MB source)" % (ms * 1e3, src.numel() * 4 / ms / 1e9, src.numel() * 4 / 1e6))
