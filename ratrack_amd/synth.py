"""Synthetic radar frame-pairs and deterministic weights (there is no dataset or checkpoint here).

The cloud generator follows SURVEY.md section 8(d): clustered, radar-like point clouds tuned against the
three real VoD frames the reference ships (N = 242/322/352 points; neighbourhood statistics at
r = 2/4/8/16 m match).  Uniform-in-a-box points would never exercise the ball-query early exit and
are deliberately not used.

Weights come from raw PCG64 output mapped to uniform ranges (no transcendental functions), keyed by
the state-dict entry name, so any process -- the fixture generator in the build container, the
tests on the GPU box, every DDP rank -- reproduces them bit-for-bit without shipping 7 MB of floats.
"""
import zlib

import numpy as np

__all__ = ["make_frame_pairs", "fill_state_dict", "tensor_for_key"]


def _one_pair(n, rng):
    K = 10
    cx = np.minimum(3.0 + rng.exponential(22.0, K), 98.0)
    cy = np.clip(rng.normal(0.0, 7.0, K), -40.0, 40.0)
    cz = rng.normal(0.5, 1.5, K)
    centres = np.stack([cx, cy, cz], 1)

    n_cl = int(round(0.8 * n))
    n_bg = n - n_cl
    cid = rng.integers(0, K, n_cl)
    sigma = np.where(rng.random(n_cl) < 0.5, 0.8, 4.0)
    cl = centres[cid] + rng.normal(0.0, 1.0, (n_cl, 3)) * sigma[:, None] * np.array([1.0, 1.0, 0.5])
    bg = np.stack([np.minimum(rng.exponential(35.0, n_bg), 100.0),
                   rng.normal(0.0, 12.0, n_bg),
                   rng.normal(0.5, 2.5, n_bg)], 1)
    pts = np.concatenate([cl, bg], 0)
    cluster_of = np.concatenate([cid, np.full(n_bg, -1)])
    perm = rng.permutation(n)
    pts, cluster_of = pts[perm], cluster_of[perm]

    feat = np.stack([rng.normal(-13.0, 13.0, n), rng.normal(-2.0, 1.7, n)], 0)  # (2,n): RCS, v_r

    # ego motion + a common displacement for 25 % of the clusters
    yaw = rng.normal(0.0, 0.01)
    t = rng.normal([-0.8, 0.0, 0.0], 0.1)
    c, s = np.cos(yaw), np.sin(yaw)
    R = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    moving = rng.random(K) < 0.25
    disp = rng.normal(0.0, 0.5, (K, 3)) * moving[:, None]
    gt_warp = pts @ R.T + t
    has_cl = cluster_of >= 0
    gt_warp[has_cl] += disp[cluster_of[has_cl]]
    gt_cls = np.zeros(n, dtype=bool)
    gt_cls[has_cl] = moving[cluster_of[has_cl]]

    pc2 = gt_warp + rng.normal(0.0, 0.05, (n, 3))
    feat2 = feat + rng.normal(0.0, 0.5, feat.shape)
    perm2 = rng.permutation(n)
    pc2, feat2 = pc2[perm2], feat2[:, perm2]
    return pts.T, pc2.T, feat, feat2, gt_warp.T, gt_cls


def make_frame_pairs(batch, n, case_id=0):
    """-> dict of float32 arrays: pc1, pc2 (B,3,n); feature1, feature2 (B,2,n); gt_warp (B,3,n);
    gt_cls bool (B,n).  Seed = 1234 + case_id."""
    rng = np.random.Generator(np.random.PCG64(1234 + case_id))
    cols = [[] for _ in range(6)]
    for _ in range(batch):
        for c, v in zip(cols, _one_pair(n, rng)):
            c.append(v)
    pc1, pc2, f1, f2, gt, cls = (np.stack(c, 0) for c in cols)
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return {"pc1": f32(pc1), "pc2": f32(pc2), "feature1": f32(f1), "feature2": f32(f2),
            "gt_warp": f32(gt), "gt_cls": np.ascontiguousarray(cls)}


# ------------------------------------------------------------------------------------------------
# deterministic weights
# ------------------------------------------------------------------------------------------------

def _uniform01(key, count, seed):
    bg = np.random.PCG64((zlib.crc32(key.encode()) << 16) ^ seed)
    raw = bg.random_raw(count)
    return (raw >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def tensor_for_key(key, shape, dtype_is_int=False, seed=20241008):
    """Deterministic value for one state-dict entry (numpy array of `shape`)."""
    count = int(np.prod(shape)) if len(shape) else 1
    if dtype_is_int or key.endswith("num_batches_tracked"):
        return np.zeros(shape, dtype=np.int64)
    u = _uniform01(key, count, seed).reshape(shape)
    if key.endswith("running_mean"):
        v = (u - 0.5) * 0.34
    elif key.endswith("running_var"):
        v = 0.5 + u
    elif len(shape) == 0:
        v = np.ones(shape)
    elif len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        a = 0.55 * np.sqrt(6.0 / fan_in)   # keeps eval-mode activations O(1) through ~30 layers
        v = (2.0 * u - 1.0) * a
    elif "bias" in key.rsplit(".", 1)[-1]:
        v = (2.0 * u - 1.0) * 0.1
    else:  # 1-D "weight": a normalisation gain in [0.3, 0.9]
        v = 0.3 + 0.6 * u
    return v.astype(np.float32)


def fill_state_dict(state_dict, seed=20241008):
    """In-place deterministic fill of a torch state_dict (any device)."""
    import torch
    for k, t in state_dict.items():
        a = tensor_for_key(k, tuple(t.shape), dtype_is_int=not t.is_floating_point(), seed=seed)
        t.copy_(torch.from_numpy(np.ascontiguousarray(a)).reshape(t.shape).to(t.dtype))
    return state_dict
