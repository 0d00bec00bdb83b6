#!/usr/bin/env python3
"""The forward cost-volume kernel alone at the bench shape, for each library given (A/B of builds inside one process: boxes differ by
+-3 %, so two builds are only comparable inside one call).   python tools/experiments/cv_alone.py [--batch 64] lib1.so lib2.so ..."""
import argparse, ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--one", default="")
ap.add_argument("libs", nargs="*")
a = ap.parse_args()
if not a.one:
    for rep in range(2):
        for so in a.libs:
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--batch", str(a.batch), "--one", so])
    sys.exit(0)
import ratrack_amd._lib as L
L.SO_PATH = os.path.abspath(a.one)
import torch
from ratrack_amd import synth
from ratrack_amd.track4d import Args, Track4D
net = Track4D(Args()).to("cuda").eval()
synth.fill_state_dict(net.state_dict()); net.invalidate_fused()
d = synth.make_frame_pairs(a.batch, 256, 1)
t = [torch.from_numpy(d[k]).to("cuda") for k in ("pc1", "pc2", "feature1", "feature2")]
with torch.no_grad():
    out = net.backbone(*t, None)
    eng = net._fused_engine()
    eng.time_dominant_kernel(10)
    ev = eng.time_dominant_kernel(60)
ms = sorted(s.elapsed_time(e) for s, e in ev)
chk = float(sum(o.double().abs().sum() for o in out if torch.is_tensor(o)))
print("%-40s cost volume forward B=%d alone: median %.1f us  min %.1f   (output checksum %.9e)" % (os.path.basename(a.one), a.batch, ms[len(ms) // 2] * 1e3, ms[0] * 1e3, chk), flush=True)
