#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python graph on CPU.

Runs ONLY in the build container (needs /root/reference); the fixtures it writes are data (inputs,
expected outputs) and travel to the GPU box, the reference never does.  Recipe = SURVEY.md 8(c):

  1. `pointnet2_cuda` (CUDA-only in the reference, lib/pointnet2_utils.py:7) is replaced in
     sys.modules by oracle/pointnet2_ref.py -- our C restatement of the ten kernels;
  2. `.cuda()` / torch.cuda.{Float,Int}Tensor become CPU no-ops / allocators;
  3. absent third-party imports that the hot path never executes are mocked;
  4. `import models` first (import-cycle order), then Track4D(args) from configs.yaml.

What this pins: the reference's Python graph (layer wiring, channel orders, BN/activation placement,
topk-based kNN, GRU, loss arithmetic, metrics) exactly; the ten native kernels only as faithfully as
oracle/pointnet2_ref.c follows the .cu sources.

Usage:  python tools/make_golden.py [--out tests/golden]
"""
import argparse
import json
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import pointnet2_ref as pref  # noqa: E402
from ratrack_amd import synth  # noqa: E402

REF_SRC = "/root/reference/src"


# ------------------------------------------------------------------------------------------------
# import recipe
# ------------------------------------------------------------------------------------------------

class Recorder(types.ModuleType):
    """pointnet2_cuda stand-in that forwards to the C oracle and logs every call."""

    def __init__(self):
        super().__init__("pointnet2_cuda")
        self.calls = []
        for name in dir(pref):
            if name.endswith("_wrapper"):
                setattr(self, name, self._wrap(name, getattr(pref, name)))

    def _wrap(self, name, fn):
        def inner(*args):
            rc = fn(*args)
            ints = [a for a in args if not torch.is_tensor(a)]
            tens = [a.detach().clone() for a in args if torch.is_tensor(a)]
            self.calls.append((name, ints, tens))
            return rc
        return inner


def import_reference():
    rec = Recorder()
    sys.modules["pointnet2_cuda"] = rec
    for name in ["cv2", "glob2", "open3d", "k3d"]:
        sys.modules[name] = MagicMock()
    nb = types.ModuleType("numba")

    def _jit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    class NumbaWarning(Warning):
        pass

    nb.jit = nb.njit = _jit
    nb.NumbaWarning = NumbaWarning
    nb.errors = types.ModuleType("numba.errors")
    for w in ["NumbaWarning", "NumbaDeprecationWarning", "NumbaPendingDeprecationWarning", "NumbaPerformanceWarning"]:
        setattr(nb.errors, w, NumbaWarning)
    sys.modules["numba"] = nb
    sys.modules["numba.errors"] = nb.errors
    tv = MagicMock()
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.ops"] = tv.ops

    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.FloatTensor = lambda *s: torch.empty(*s, dtype=torch.float32)
    torch.cuda.IntTensor = lambda *s: torch.empty(*s, dtype=torch.int32)

    sys.path.insert(0, REF_SRC)
    import models  # noqa: F401  (must come first: import cycle)
    from models.track4d import Track4D
    from utils.parser_util import parse_args_from_yaml
    import utils.model_utils.model_utils as mu
    import losses.loss as ref_loss
    import main_utils as ref_main_utils
    args = parse_args_from_yaml(os.path.join(REF_SRC, "configs.yaml"))
    return rec, Track4D, args, mu, ref_loss, ref_main_utils


# ------------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------------

def npf(t):
    return t.detach().cpu().numpy().astype(np.float32)


def checksum(t):
    a = t.detach().cpu().numpy().astype(np.float64)
    return np.array([a.sum(), np.abs(a).sum()], dtype=np.float64)


def special_cloud(n=256):
    """B=1 cloud with 16 exact duplicates and one isolated point: tie / empty-ball behaviour."""
    d = synth.make_frame_pairs(1, n, case_id=4)
    for key, fkey in (("pc1", "feature1"), ("pc2", "feature2")):
        p = d[key]
        p[0, :, 40:56] = p[0, :, 8:9]            # 16 exact copies of point 8 ...
        d[fkey][0, :, 40:56] = d[fkey][0, :, 8:9]  # ... features too: torch.topk's pick among 17 ties is unspecified,
        #                                            identical points make every pick equivalent downstream
        p[0, :, 200] = np.array([250.0, -180.0, 40.0], dtype=np.float32)  # isolated: empty balls at r<=16
    return d


def run_backbone(net, rec, mu, d, h=None, capture=True):
    """One backbone() with hooks; returns dict of arrays."""
    out = {}
    B, _, N = d["pc1"].shape
    pc1, pc2 = torch.from_numpy(d["pc1"]), torch.from_numpy(d["pc2"])
    f1, f2 = torch.from_numpy(d["feature1"]), torch.from_numpy(d["feature2"])
    if h is None:
        h = torch.zeros(5, B, 128)

    acts = {}
    handles = []
    counters = {}

    def hook(name):
        def fn(mod, inp, outp):
            c = counters.get(name, 0)
            counters[name] = c + 1
            acts["%s#%d" % (name, c)] = outp
        return fn

    if capture:
        for name in ["sa1", "sa2", "sa3", "linear1", "linear2", "linear3", "fp3", "fp2", "fp1"]:
            handles.append(getattr(net.pn_head, name).register_forward_hook(hook("pn_head." + name)))
            handles.append(getattr(net.fd_layer.mse, name).register_forward_hook(hook("mse." + name)))

    knn_log = []
    orig_knn, orig_sqd = mu.knn_point, mu.square_distance

    def knn_rec(nsample, xyz, new_xyz):
        idx = orig_knn(nsample, xyz, new_xyz)
        dist = orig_sqd(new_xyz, xyz)
        knn_log.append((idx.clone(), dist.clone()))
        return idx

    mu.knn_point = knn_rec
    rec.calls.clear()
    try:
        flow, h_out, cls, cor, pc1_feat, pc2_feat, prop = net.backbone(pc1, pc2, f1, f2, h)
    finally:
        mu.knn_point = orig_knn
        for hd in handles:
            hd.remove()

    out["flow"], out["h_out"], out["cls"] = npf(flow), npf(h_out), npf(cls)
    for name, t in [("cor", cor), ("pc1_features", pc1_feat), ("pc2_features", pc2_feat), ("prop", prop)]:
        out[name + "_s8"] = npf(t[:, :, ::8])
        out[name + "_cs"] = checksum(t)
    out["flow_cs"] = checksum(flow)

    if capture:
        # ---- native call trace: 30 calls per PNHead, order of SURVEY.md A.2 -----------------------
        calls = list(rec.calls)
        assert len(calls) == 90, len(calls)
        for c in range(3):
            seg = calls[c * 30:(c + 1) * 30]
            names = [s[0] for s in seg]
            fps = [s for s in seg if s[0] == "furthest_point_sampling_wrapper"]
            balls = [s for s in seg if s[0] == "ball_query_wrapper"]
            tnn = [s for s in seg if s[0] == "three_nn_wrapper"]
            assert len(fps) == 3 and len(balls) == 6 and len(tnn) == 3, names
            for i, s in enumerate(fps):
                out["fps_idx_c%d_l%d" % (c, i + 1)] = s[2][2].numpy().astype(np.int16)
            for i, s in enumerate(balls):
                idx = s[2][2].numpy()
                w = (np.arange(idx.size, dtype=np.int64).reshape(idx.shape) % 1009) + 1
                out["ball_cs_c%d_%d" % (c, i)] = np.array([idx.astype(np.int64).sum(), (idx.astype(np.int64) * w).sum()])
                if c == 0:
                    out["ball_idx_%d" % i] = idx.astype(np.int16)
                    out["ball_radius_%d" % i] = np.float32(s[1][3])
            for i, s in enumerate(tnn):
                if c == 0:
                    out["three_nn_dist2_%d" % i] = s[2][2].numpy()
                    out["three_nn_idx_%d" % i] = s[2][3].numpy().astype(np.int16)
        # ---- kNN (torch.topk on the expansion-formula distance): sorted index sets -----------------
        assert len(knn_log) == 2
        for i, (idx, dist) in enumerate(knn_log):
            out["knn_set_%d" % i] = np.sort(idx.numpy(), axis=-1).astype(np.int16)
            out["knn_dist_row0_%d" % i] = npf(dist[0, 0])
            # distance of the k-th and (k+1)-th neighbour: lets a test detect boundary ties
            srt = torch.sort(dist, dim=-1)[0]
            out["knn_kth_gap_%d" % i] = npf(srt[:, :, 16] - srt[:, :, 15])
        # ---- activations (PNHead call 0 = pc1, call 1 = pc2; mse call 0) ---------------------------
        for key, v in acts.items():
            mod, call = key.split("#")
            if call != "0":
                continue
            if isinstance(v, tuple):          # SA modules return (new_xyz, features)
                out["act_%s_xyz" % mod] = npf(v[0][:, :8])
                out["act_%s" % mod] = npf(v[1][:, :, :8])
                out["act_%s_cs" % mod] = checksum(v[1])
            elif "linear" in mod:             # (B, S, C)
                out["act_%s" % mod] = npf(v[:, :8, :])
                out["act_%s_cs" % mod] = checksum(v)
            else:                             # FP modules (B, C, n)
                out["act_%s" % mod] = npf(v[:, :, :8])
                out["act_%s_cs" % mod] = checksum(v)
    return out, (flow, h_out, cls, cor, pc1_feat, pc2_feat, prop)


def build_net(Track4D, args, train=False):
    torch.manual_seed(0)
    net = Track4D(args)
    synth.fill_state_dict(net.state_dict())
    net.train(train)
    return net


def save(path, d, arrays):
    payload = {("in_" + k): v for k, v in d.items()}
    payload.update(arrays)
    np.savez_compressed(path, **payload)
    print("wrote %s  %.1f KB  (%d arrays)" % (path, os.path.getsize(path) / 1024, len(payload)))


# ------------------------------------------------------------------------------------------------
# cases
# ------------------------------------------------------------------------------------------------

def eval_case(name, d, Track4D, args, rec, mu, ref_main_utils, outdir):
    net = build_net(Track4D, args, train=False)
    with torch.no_grad():
        out, tensors = run_backbone(net, rec, mu, d)
        flow, h_out, cls = tensors[0], tensors[1], tensors[2]
        # temporal recurrence: second call fed with h_out
        out2, _ = run_backbone(net, rec, mu, d, h=h_out, capture=False)
        out["flow_step2"], out["h_out_step2"] = out2["flow"], out2["h_out"]
        # metrics (main_utils.py:342-389) on batch element 0, as the epoch loop does (B=1)
        pc1 = torch.from_numpy(d["pc1"][:1])
        pc1_warp = pc1 + flow[:1]
        gt = torch.from_numpy(d["gt_warp"][:1])
        mask = torch.from_numpy(~d["gt_cls"][:1]).float()   # epoch(): mask = 1 for static points
        sf = ref_main_utils.eval_scene_flow(pc1, pc1_warp, gt, mask)
        pre = (cls[:1] > 0.5).float()
        seg = ref_main_utils.eval_motion_seg(pre, torch.from_numpy(d["gt_cls"][:1]).float())
        out["metric_sf_keys"] = np.array(sorted(sf.keys()))
        out["metric_sf_vals"] = np.array([sf[k] for k in sorted(sf.keys())], dtype=np.float64)
        out["metric_seg_keys"] = np.array(sorted(seg.keys()))
        out["metric_seg_vals"] = np.array([seg[k] for k in sorted(seg.keys())], dtype=np.float64)
    save(os.path.join(outdir, name + ".npz"), d, out)


def train_case(name, d, Track4D, args, rec, mu, ref_loss, outdir):
    """B=1 train-mode forward + multi-task loss + backward (losses/loss.py:8-31, main_utils.py:148-156)."""
    net = build_net(Track4D, args, train=True)
    out, tensors = run_backbone(net, rec, mu, d, capture=False)
    flow, h_out, cls = tensors[0], tensors[1], tensors[2]
    pc1, pc2 = torch.from_numpy(d["pc1"]), torch.from_numpy(d["pc2"])
    gt = torch.from_numpy(d["gt_warp"])
    gt_cls = torch.from_numpy(d["gt_cls"][0])
    pc1_warp = pc1 + flow

    def loss(pretrain, gcls):
        return ref_loss.track_4d_loss(None, None, {}, {}, None, None, None, pc1, pc2, pc1_warp, cls, gt, [],
                                      None, gcls, None, None, None, pretrain=pretrain)

    total, items = loss(False, gt_cls)
    keys = ["Loss", "SceneFlowLoss", "TrackingLoss", "SegLoss"]
    out["loss_keys"] = np.array(keys)
    out["loss_vals"] = np.array([float(items[k]) for k in keys], dtype=np.float64)
    tp, ip = loss(True, gt_cls)
    out["loss_vals_pretrain"] = np.array([float(ip[k]) for k in keys], dtype=np.float64)
    tn, inn = loss(False, torch.zeros_like(gt_cls))   # no positive label -> BCE over an empty set -> NaN -> 0
    out["loss_vals_nopos"] = np.array([float(inn[k]) for k in keys], dtype=np.float64)

    net.zero_grad()
    total.backward()
    names, norms = [], []
    for k, p in net.named_parameters():
        names.append(k)
        norms.append(float(p.grad.norm()) if p.grad is not None else -1.0)
    out["grad_names"] = np.array(names)
    out["grad_norms"] = np.array(norms, dtype=np.float64)
    # BN buffers after the forward (momentum 0.1 update; pn_head runs twice per step)
    for k, v in net.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            if k.startswith("pn_head.sa") or k.startswith("fd_layer.fp") or k.startswith("fd_layer.mse.fp") \
                    or k.startswith("pn_head.fp"):
                out["bn/" + k] = npf(v)
        if k.endswith("num_batches_tracked") and (k.startswith("pn_head.sa1.mlps.0.layer0") or k.startswith("fd_layer.fp.sf_mlp.0")):
            out["bn/" + k] = v.numpy()
    save(os.path.join(outdir, name + ".npz"), d, out)


def forward_case(name, Track4D, args, rec, outdir):
    """Two consecutive B=1 frames through the reference's full forward() (models/track4d.py:49-65): backbone ->
    mover selection -> sklearn DBSCAN -> Affinity MLP + log-Sinkhorn association with the previous frame's objects.
    The cls bias is raised so that a realistic share of points is classified as moving."""
    net = build_net(Track4D, args, train=False)
    sd = net.state_dict()
    sd["fd_layer.cp.linear.bias"] += FORWARD_CLS_BIAS_SHIFT
    frames = [synth.make_frame_pairs(1, 256, 20), synth.make_frame_pairs(1, 256, 21)]
    out = {}
    objects_prev, h = dict(), None
    with torch.no_grad():
        for fi, d in enumerate(frames):
            t = {k: torch.from_numpy(v) for k, v in d.items() if k != "gt_cls"}
            for k, v in d.items():
                out["f%d_in_%s" % (fi, k)] = v
            if h is None:
                h = torch.zeros(5, 1, 128)
            h, pc1_warp, cls, aff_list, aff_mat, indices1, confs, objects, timeout, objects_curr = net(
                t["pc1"], t["pc2"], t["feature1"], t["feature2"], h, objects_prev)
            p = "f%d_" % fi
            out[p + "pc1_warp"], out[p + "cls"], out[p + "h"] = npf(pc1_warp), npf(cls), npf(h)
            out[p + "n_objects_curr"] = np.int64(len(objects_curr))
            out[p + "object_sizes_curr"] = np.array([o.shape[2] for o in objects_curr], dtype=np.int64)
            out[p + "object_first_xyz"] = np.array([npf(o[0, 3:6, 0]) for o in objects_curr], dtype=np.float32).reshape(-1, 3)
            out[p + "aff_mat"] = npf(aff_mat) if torch.is_tensor(aff_mat) else np.zeros((1, 0, 0), np.float32)
            out[p + "indices1"] = indices1.numpy().astype(np.int64) if indices1 is not None else np.zeros((0,), np.int64)
            out[p + "confs"] = np.array([float(c) for c in confs], dtype=np.float64)
            out[p + "object_ids"] = np.array(list(objects.keys()), dtype=np.int64)
            out[p + "object_sizes"] = np.array([objects[k].shape[2] for k in objects], dtype=np.int64)
            out[p + "max_id"] = np.int64(net.max_id)
            objects_prev = {k: v.clone().detach() for k, v in objects.items()}
            margin = float((cls - 0.5).abs().min())
            assert margin > 2e-5, "a point sits on the mover threshold (margin %.2e): pick another FORWARD_CLS_BIAS_SHIFT" % margin
            print("  frame %d: movers %d (threshold margin %.1e), clusters %d, ids %s, confs %s" % (fi, int((cls > 0.5).sum()), margin, len(objects_curr),
                                                                         list(objects.keys()), [round(float(c), 4) for c in confs]))
    np.savez_compressed(os.path.join(outdir, name + ".npz"), **out)
    print("wrote %s (%d arrays)" % (name, len(out)))


FORWARD_CLS_BIAS_SHIFT = 0.09


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    torch.set_num_threads(8)
    rec, Track4D, args, mu, ref_loss, ref_main_utils = import_reference()

    # checkpoint-compatibility contract: the reference's state-dict keys, shapes, dtypes
    net = Track4D(args)
    sd = net.state_dict()
    spec = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()}
    live = sorted(k for k, p in net.named_parameters())
    with open(os.path.join(a.out, "state_dict_spec.json"), "w") as f:
        json.dump({"entries": spec, "parameters": live,
                   "n_params": int(sum(p.numel() for p in net.parameters()))}, f, indent=0, sort_keys=True)
    print("state_dict_spec.json: %d entries" % len(spec))

    cases = {
        "eval_b2_n256": lambda: eval_case("eval_b2_n256", synth.make_frame_pairs(2, 256, 0), Track4D, args, rec, mu, ref_main_utils, a.out),
        "eval_b1_n242": lambda: eval_case("eval_b1_n242", synth.make_frame_pairs(1, 242, 2), Track4D, args, rec, mu, ref_main_utils, a.out),
        "eval_b1_n1024": lambda: eval_case("eval_b1_n1024", synth.make_frame_pairs(1, 1024, 3), Track4D, args, rec, mu, ref_main_utils, a.out),
        "eval_b1_n256_dups": lambda: eval_case("eval_b1_n256_dups", special_cloud(256), Track4D, args, rec, mu, ref_main_utils, a.out),
        "train_b1_n256": lambda: train_case("train_b1_n256", synth.make_frame_pairs(1, 256, 1), Track4D, args, rec, mu, ref_loss, a.out),
        "forward_b1_n256": lambda: forward_case("forward_b1_n256", Track4D, args, rec, a.out),
    }
    for name, fn in cases.items():
        if a.only and a.only != name:
            continue
        fn()


if __name__ == "__main__":
    main()
