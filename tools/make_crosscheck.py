#!/usr/bin/env python3
"""Generate tests/golden/crosscheck_pytorch_pointnet2.npz: outputs of the reference's INDEPENDENT pure-PyTorch
PointNet++ helpers (src/models/pointnet2_utils.py:66-111 farthest_point_sample / query_ball_point, :302-304 the
three-nearest-neighbour selection of PointNetFeaturePropagation) on seeded clouds.

SURVEY 8(c): those functions are dead code on the live path but are a second, independently written statement of what
the CUDA kernels compute, so they pin oracle/pointnet2_ref.c on inputs where rounding cannot change a decision.
Runs ONLY in the build container (imports /root/reference); the .npz holds inputs' seeds and expected outputs -- data.
tests/test_emulator_cpu.py::test_oracle_vs_reference_pytorch_helpers compares the C oracle with it.

The pure-PyTorch versions differ from the kernels in arithmetic (expansion-formula distances, `>` instead of `<`,
sort instead of scan), so each output row carries a `safe` flag: the decision margins of that row (distance to the
ball surface, gap between the 3rd and 4th neighbour, gap between the two largest FPS candidates), evaluated in
float64, are far above fp32 rounding.  Only safe rows are compared -- and the test asserts that they are the vast majority.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ratrack_amd import synth  # noqa: E402

REF_FILE = "/root/reference/src/models/pointnet2_utils.py"
MARGIN = 1e-4       # relative decision margin required for a row to count as rounding-proof
MARGIN_FPS = 5e-6   # FPS: a sequential chain -- one unsafe round taints the rest, so the bar is 40x fp32 epsilon instead


def load_reference_helpers():
    spec = importlib.util.spec_from_file_location("ref_pure_pointnet2", REF_FILE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def clouds(case_id, b, n):
    d = synth.make_frame_pairs(b, n, case_id)
    return (torch.from_numpy(d["pc1"]).permute(0, 2, 1).contiguous(), torch.from_numpy(d["pc2"]).permute(0, 2, 1).contiguous())


def d2_f64(a, b):
    a, b = a.double(), b.double()
    return ((a[:, :, None, :] - b[:, None, :, :]) ** 2).sum(-1)


def expansion_error(a, b):
    """Absolute error bound of the reference helpers' fp32 expansion-formula distance |a|^2 + |b|^2 - 2ab (:21-42):
    a few ulps of the LARGEST term, i.e. it scales with the squared norms (x up to 100 m), not with the distance."""
    a, b = a.double(), b.double()
    return 16 * 2.0 ** -23 * ((a ** 2).sum(-1)[:, :, None] + (b ** 2).sum(-1)[:, None, :])


def main(out_path):
    R = load_reference_helpers()
    out = {"margin": np.float64(MARGIN)}
    # ---- farthest_point_sample (:66-88).  Its start index is random (:79): pin it to 0 like the kernel (:113).
    real_randint = torch.randint
    torch.randint = lambda low, high, size, **kw: torch.zeros(size, dtype=kw.get("dtype", torch.long))
    try:
        for tag, (case_id, b, n, npoint) in {"fps_a": (101, 2, 256, 128), "fps_b": (130, 1, 1024, 512), "fps_c": (103, 2, 242, 200)}.items():
            xyz, _ = clouds(case_id, b, n)
            idx = R.farthest_point_sample(xyz, npoint)
            # margin of every round, float64: gap between the two largest running min-distances
            x = xyz.double()
            safe = np.ones((b, npoint), dtype=bool)
            for bi in range(b):
                dist = torch.full((n,), 1e10, dtype=torch.float64)
                for j in range(1, npoint):
                    c = x[bi, idx[bi, j - 1]]
                    dist = torch.minimum(dist, ((x[bi] - c) ** 2).sum(-1))
                    top = torch.topk(dist, 2).values
                    safe[bi, j] = bool((top[0] - top[1]) > MARGIN_FPS * top[0])
            safe = np.logical_and.accumulate(safe, axis=1)        # FPS is sequential: one unsafe round taints the rest
            out[tag + "_args"] = np.array([case_id, b, n, npoint])
            out[tag + "_idx"] = idx.numpy().astype(np.int32)
            out[tag + "_safe"] = safe
    finally:
        torch.randint = real_randint
    # ---- query_ball_point (:91-111): first nsample in index order inside the ball, padded with the first
    for tag, (case_id, b, n, s, radius, nsample) in {"ball_a": (111, 2, 256, 256, 2.0, 4), "ball_b": (112, 2, 256, 256, 4.0, 8),
                                                     "ball_c": (113, 1, 1024, 512, 8.0, 16), "ball_d": (114, 2, 242, 242, 16.0, 32)}.items():
        xyz, _ = clouds(case_id, b, n)
        new_xyz = xyz[:, :s].contiguous()           # centroids are source points: no ball is empty
        idx = R.query_ball_point(radius, nsample, xyz, new_xyz)
        d2 = d2_f64(new_xyz, xyz)
        safe = ((d2 - radius ** 2).abs() > MARGIN * radius ** 2 + expansion_error(new_xyz, xyz)).all(-1).numpy()
        out[tag + "_args"] = np.array([case_id, b, n, s, radius, nsample], dtype=np.float64)
        out[tag + "_idx"] = idx.numpy().astype(np.int32)
        out[tag + "_safe"] = safe
    # ---- three nearest neighbours as PointNetFeaturePropagation selects them (:302-304): sort, first three
    for tag, (case_id, b, n, m) in {"nn_a": (121, 2, 256, 512), "nn_b": (122, 1, 1024, 512), "nn_c": (123, 2, 242, 100)}.items():
        unknown, known = clouds(case_id, b, max(n, m))
        unknown, known = unknown[:, :n].contiguous(), known[:, :m].contiguous()
        dists = R.square_distance(unknown, known)
        dists, idx = dists.sort(dim=-1)
        idx = idx[:, :, :3]
        d2s, order = d2_f64(unknown, known).sort(dim=-1)
        d2s, err = d2s[:, :, :4], torch.gather(expansion_error(unknown, known), 2, order[:, :, :4])
        gaps = d2s[:, :, 1:] - d2s[:, :, :-1]
        safe = (gaps > MARGIN * d2s[:, :, 1:] + 2 * err[:, :, 1:]).all(-1).numpy()
        out[tag + "_args"] = np.array([case_id, b, n, m])
        out[tag + "_idx"] = idx.numpy().astype(np.int32)
        out[tag + "_safe"] = safe
    np.savez_compressed(out_path, **out)
    for k in sorted(out):
        if k.endswith("_safe"):
            print("%-12s safe rows: %d / %d" % (k, int(out[k].sum()), out[k].size))
    print("wrote", out_path)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "crosscheck_pytorch_pointnet2.npz"))
