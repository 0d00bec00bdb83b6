"""rtk_cost_volume_bwd at the bench shape: ms per launch and fraction of the fp32 MFMA peak."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ratrack_amd import train_ops as T
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ms, fl = T.time_cost_volume_bwd(B, 256, "cuda", 20)
print("cost_volume_bwd B=%d: %.4f ms, %.1f TFLOP/s, frac %.3f" % (B, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3))
