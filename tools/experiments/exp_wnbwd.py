"""Experiment: rtk_weightnet_bwd at the train-step shape (B=64: 262144 (point, neighbour) positions x 256 channels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ratrack_amd import _lib, train_ops as T
from ratrack_amd.benchutil import time_graph
dev = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
M, C = B * 256 * 16, 256
r = lambda *s: torch.randn(*s, device=dev)
d4, dq3, dt2 = r(M, 4), r(M, C), r(M, 8)
wa, ba, wb, bb = r(8, 3), r(8), r(8, 8), r(8)
z = lambda *s: torch.zeros(*s, device=dev)
ws = torch.empty(1024 * ((9 * C + 107) & ~3), device=dev)
dwa, dba, dwb, dbb, dwc, dbc = z(8, 3), z(8), z(8, 8), z(8), z(C, 8), z(C)
ms = time_graph(lambda: _lib.call("rtk_weightnet_bwd", M, C, d4.data_ptr(), dq3.data_ptr(), dt2.data_ptr(), wa.data_ptr(), ba.data_ptr(),
                                  wb.data_ptr(), bb.data_ptr(), dwa.data_ptr(), dba.data_ptr(), dwb.data_ptr(), dbb.data_ptr(), dwc.data_ptr(),
                                  dbc.data_ptr(), ws.data_ptr(), ws.numel(), T._stream()), 10)
print("weightnet_bwd M=%d C=%d: %.1f us  (%.2f TB/s of the %.0f MB dq3 stream)" % (M, C, ms * 1e3, dq3.numel() * 4 / ms / 1e9, dq3.numel() * 4 / 1e6))
