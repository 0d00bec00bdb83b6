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



def full_loss_case(name, Track4D, args, ref_loss, outdir):
    """Round 5: ONE TRAINING ITERATION AFTER PRE-TRAINING through the reference's forward() (main_utils.py:127-156, losses/loss.py:8-31,
    48-72): net.train(), frame 0 forward (its objects become objects_prev, detached, as the epoch loop does), frame 1 forward with
    them, total = 0.5 L_sf + 0.5 L_trk + L_seg (pretrain = False), backward.  L_trk is the BCE between the Affinity-MLP's list and the
    identity-match matrix of the two GT mappings, so its gradient reaches the Affinity MLP and, through the pooled object
    descriptors, the backbone.  The mappings are synthetic (GT key lists of the right lengths with two matching keys): the loss
    reads nothing but their keys (losses/loss.py:52-66)."""
    net = build_net(Track4D, args, train=True)
    with torch.no_grad():
        net.state_dict()["fd_layer.cp.linear.bias"].add_(FORWARD_CLS_BIAS_SHIFT)
    # (seeds: pairs on which two fp32 evaluations of the step flip no ReLU / max-pool decision in the decoder -- the hand-written
    # training path and the module path agree to 1e-5 on them, tools/experiments/dbg_seed_scan.py; on others one flipped activation
    # under B = 1 batch statistics moves every upstream gradient by ~1e-2, tests/test_varn_train_gpu.py -- with objects in both
    # frames, movers well off the 0.5 threshold and moving GT points)
    frames = [synth.make_frame_pairs(1, 256, 28), synth.make_frame_pairs(1, 256, 22)]
    out = {}
    for fi, d in enumerate(frames):
        for k, v in d.items():
            out["f%d_in_%s" % (fi, k)] = v
    t0 = {k: torch.from_numpy(v) for k, v in frames[0].items() if k != "gt_cls"}
    h = torch.zeros(5, 1, 128)
    h, _, cls0, _, _, _, _, objects, _, oc0 = net(t0["pc1"], t0["pc2"], t0["feature1"], t0["feature2"], h, dict())
    objects_prev = {k: v.clone().detach() for k, v in objects.items()}
    h = h.detach()
    out["f0_object_ids"] = np.array(list(objects_prev.keys()), dtype=np.int64)
    out["f0_object_sizes"] = np.array([objects_prev[k].shape[2] for k in objects_prev], dtype=np.int64)
    d = frames[1]
    t1 = {k: torch.from_numpy(v) for k, v in d.items() if k != "gt_cls"}
    h1, pc1_warp, cls, aff_list, aff_mat, indices1, confs, objects, _, objects_curr = net(t1["pc1"], t1["pc2"], t1["feature1"], t1["feature2"],
                                                                                          h, objects_prev)
    n_prev, n_curr = len(objects_prev), len(objects_curr)
    assert n_prev >= 2 and n_curr >= 2 and aff_list.numel() == n_prev * n_curr, (n_prev, n_curr, aff_list.shape)
    prev_keys = [10 + i for i in range(n_prev)]
    curr_keys = [10 + ((i + 1) % max(n_prev, n_curr)) if i < 3 else 500 + i for i in range(n_curr)]      # some keys match, at shifted places
    mp = {k: i for i, k in enumerate(prev_keys)}
    mc = {k: i for i, k in enumerate(curr_keys)}
    gt = torch.from_numpy(d["gt_warp"])
    gt_cls = torch.from_numpy(d["gt_cls"][0])
    if int(gt_cls.sum()) == 0 or int((~gt_cls).sum()) == 0:      # (this synthetic pair has no moving GT point: label every third one, so that
        gt_cls = torch.arange(gt_cls.numel()) % 3 == 0           # the segmentation term is defined)
    out["f1_gt_cls_used"] = gt_cls.numpy()
    total, items = ref_loss.track_4d_loss(objects_prev, objects, mp, mc, None, None, None, t1["pc1"], t1["pc2"], pc1_warp, cls, gt, aff_list,
                                          None, gt_cls, None, None, None, pretrain=False)
    keys = ["Loss", "SceneFlowLoss", "TrackingLoss", "SegLoss"]
    out["loss_keys"] = np.array(keys)
    out["loss_vals"] = np.array([float(items[k]) for k in keys], dtype=np.float64)
    assert float(items["TrackingLoss"]) > 0
    out["prev_keys"], out["curr_keys"] = np.array(prev_keys, dtype=np.int64), np.array(curr_keys, dtype=np.int64)
    out["aff_list"] = npf(aff_list)
    out["f1_cls"], out["f1_pc1_warp"] = npf(cls), npf(pc1_warp)
    out["f1_object_sizes_curr"] = np.array([o.shape[2] for o in objects_curr], dtype=np.int64)
    net.zero_grad()
    total.backward()
    grad_records([(k, p.grad) for k, p in net.named_parameters()], out)
    np.savez_compressed(os.path.join(outdir, name + ".npz"), **out)
    nz = sum(1 for k, p in net.named_parameters() if p.grad is not None and float(p.grad.abs().max()) > 0)
    print("wrote %s: %d x %d affinities, losses %s, %d parameters with a gradient (affinity.*: %s)"
          % (name, n_prev, n_curr, [round(float(items[k]), 5) for k in keys], nz,
             [round(float(p.grad.norm()), 6) for k, p in net.named_parameters() if k.startswith("affinity") and p.grad is not None][:4]))


# ------------------------------------------------------------------------------------------------
# round 3: gradient tensors, a batch that reaches the split-bf16 training kernels, real frames
# ------------------------------------------------------------------------------------------------

GRAD_SAMPLE = 1024          # elements kept per gradient tensor (flat, odd stride): every parameter, ~0.5 MB per fixture


def grad_stride(numel):
    s = -(-numel // GRAD_SAMPLE)
    return s | 1 if s > 1 else 1


def probe_vector(key, numel):
    """Fixed pseudo-random direction in [-1, 1) for a parameter (raw PCG64 -> uniform, as synth.tensor_for_key)."""
    return (2.0 * synth._uniform01("probe/" + key, numel, 77) - 1.0)


def grad_records(named_grads, out, prefix="", samples=True):
    """Per-parameter gradient records: L2 norm, the dot product with a fixed probe direction (covers the WHOLE tensor: a
    wrong-direction gradient with the right norm fails it) and a flat odd-stride sample of the tensor itself."""
    names, norms, probes = [], [], []
    for k, g in named_grads:
        names.append(k)
        if g is None:
            norms.append(-1.0)
            probes.append(0.0)
            continue
        a = g.detach().cpu().numpy().astype(np.float64).ravel()
        norms.append(float(np.sqrt((a * a).sum())))
        probes.append(float((a * probe_vector(k, a.size)).sum()))
        if samples:
            out[prefix + "grad/" + k] = a[::grad_stride(a.size)].astype(np.float32)
    if not prefix:
        out["grad_names"] = np.array(names)
    out[prefix + "grad_norms"] = np.array(norms, dtype=np.float64)
    out[prefix + "grad_probes"] = np.array(probes, dtype=np.float64)


def bn_records(sd, out, prefix="bn/"):
    for k, v in sd.items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            if k.startswith("pn_head.") or k.startswith("fd_layer.mse.") or k.startswith("fd_layer.fp.") or k.startswith("fd_layer.cp."):
                out[prefix + k] = npf(v)
        if k.endswith("num_batches_tracked") and (k.startswith("pn_head.sa1.mlps.0.layer0") or k.startswith("fd_layer.fp.sf_mlp.0")):
            out[prefix + k] = v.numpy()


def batch_mean_loss(ref_loss, pc1, pc2, pc1_warp, cls, gt, gt_cls, pretrain):
    """The reference's loss is B = 1 code (losses/loss.py:89 takes batch element 0, :131-142 index a (1,N) target); for a batch
    it is called per sample on B = 1 slices and averaged -- ratrack_amd.loss.backbone_loss is defined as exactly this."""
    keys = ["Loss", "SceneFlowLoss", "TrackingLoss", "SegLoss"]
    tot, acc = 0.0, {k: 0.0 for k in keys}
    B = pc1.shape[0]
    for b in range(B):
        s = slice(b, b + 1)
        t, it = ref_loss.track_4d_loss(None, None, {}, {}, None, None, None, pc1[s], pc2[s], pc1_warp[s], cls[s], gt[s], [],
                                       None, gt_cls[b], None, None, None, pretrain=pretrain)
        tot = tot + t / B
        for k in keys:
            acc[k] = acc[k] + (it[k] if torch.is_tensor(it[k]) else torch.tensor(float(it[k]))) / B
    return tot, acc, keys


def arbiter_train_step(d, out, pretrain=False, samples=True):
    """The same train step through oracle/track4d_ref.py in FLOAT64 (index ops on the fp32 coordinates: same geometry) -- the
    arbiter between two fp32 implementations.  Also the fp32 oracle's own distance to it (the noise floor of an fp32 train
    step at this batch size)."""
    from oracle import track4d_ref as R
    res = {}
    for tag, dt in (("f64", torch.float64), ("o32", torch.float32)):
        net = build_net(_REF["Track4D"], _REF["args"], train=True)
        sd = {k: (v.detach().clone().to(dt) if v.is_floating_point() else v.detach().clone()) for k, v in net.state_dict().items()}
        names = [k for k, _ in net.named_parameters()]
        for k in names:
            sd[k].requires_grad_(True)
        t = {k: torch.from_numpy(v).to(dt) for k, v in d.items() if k != "gt_cls"}
        flow, h, cls, *_ = R.backbone(sd, t["pc1"], t["pc2"], t["feature1"], t["feature2"], None, training=True)
        warp = t["pc1"] + flow
        gcls = torch.from_numpy(d["gt_cls"])
        B = warp.shape[0]
        total = 0.0
        for b in range(B):
            tb, _ = R.track_4d_loss(warp[b:b + 1], cls[b:b + 1], t["gt_warp"][b:b + 1], gcls[b], pretrain=pretrain)
            total = total + tb / B
        total.backward()
        res[tag] = (float(total), [(k, sd[k].grad) for k in names], flow.detach(), cls.detach())
    grad_records(res["f64"][1], out, prefix="arb_", samples=samples)
    out["arb_loss"] = np.float64(res["f64"][0])
    out["arb_flow"], out["arb_cls"] = res["f64"][2].numpy().astype(np.float32), res["f64"][3].numpy().astype(np.float32)
    # noise floor: fp32 oracle vs float64, per parameter, relative to the tensor's largest element
    floor = []
    for (k, g64), (_, g32) in zip(res["f64"][1], res["o32"][1]):
        if g64 is None:
            floor.append(-1.0)
            continue
        floor.append(float((g32.double() - g64).abs().max() / g64.abs().max().clamp_min(1e-300)))
    out["arb_fp32_oracle_relerr"] = np.array(floor, dtype=np.float64)
    return res


def train_batch_case(name, d, ref, outdir, pretrain=False):
    """B > 1 train step through the REFERENCE graph: batch-statistic BatchNorm over the batch, loss = batch mean of the
    reference's B = 1 loss.  At B = 8 x N = 256 = 2 048 query points the product's split-bf16 training kernels run."""
    rec, Track4D, args, mu, ref_loss = ref
    net = build_net(Track4D, args, train=True)
    out, tensors = run_backbone(net, rec, mu, d, capture=False)
    out = {k: v for k, v in out.items() if not k.endswith("_s8")}          # (the strided activation samples are eval-case material)
    flow, h_out, cls = tensors[0], tensors[1], tensors[2]
    pc1, pc2 = torch.from_numpy(d["pc1"]), torch.from_numpy(d["pc2"])
    gt, gt_cls = torch.from_numpy(d["gt_warp"]), torch.from_numpy(d["gt_cls"])
    total, items, keys = batch_mean_loss(ref_loss, pc1, pc2, pc1 + flow, cls, gt, gt_cls, pretrain)
    out["loss_keys"] = np.array(keys)
    out["loss_vals"] = np.array([float(items[k]) for k in keys], dtype=np.float64)
    net.zero_grad()
    total.backward()
    grad_records([(k, p.grad) for k, p in net.named_parameters()], out)
    bn_records(net.state_dict(), out)
    arbiter_train_step(d, out, pretrain)
    save(os.path.join(outdir, name + ".npz"), d, out)


REAL_FRAMES = ["00549", "01047", "01201"]


def real_pair(later, earlier, seed):
    """A frame pair of two of the radar frames the reference ships (different sizes, as every real consecutive pair) + seeded
    stand-in GT of the right shapes (GT generation is pinned separately, tools/make_golden_gt.py)."""
    from ratrack_amd import vod_io
    ex = os.path.join(ROOT, "tests", "golden", "vod_example")
    a = vod_io.load_radar_bin(os.path.join(ex, "radar_%s.bin" % later))
    b = vod_io.load_radar_bin(os.path.join(ex, "radar_%s.bin" % earlier))
    pc1, pc2, f1, f2 = vod_io.frame_pair_tensors(a, b)
    rng = np.random.Generator(np.random.PCG64(4321 + seed))
    n1 = pc1.shape[2]
    gt_warp = pc1.numpy() + np.array([-0.8, 0.0, 0.0], np.float32).reshape(1, 3, 1) + rng.normal(0, 0.2, (1, 3, n1)).astype(np.float32)
    gt_cls = (np.abs(a[:, 5]) > 1.0).reshape(1, n1)              # |compensated radial velocity| > 1 m/s
    assert 0 < gt_cls.sum() < n1
    return {"pc1": pc1.numpy(), "pc2": pc2.numpy(), "feature1": f1.numpy(), "feature2": f2.numpy(),
            "gt_warp": gt_warp.astype(np.float32), "gt_cls": gt_cls}


def real_case(name, later, earlier, seed, ref, ref_main_utils, outdir, arb_samples=True):
    """B = 1, N1 != N2 through the reference graph, as the reference's epoch loop runs it (main_utils.py:76-80,127): one
    eval-mode forward (everything eval_case captures) and one train step (loss, gradients, BatchNorm statistics)."""
    rec, Track4D, args, mu, ref_loss = ref
    d = real_pair(later, earlier, seed)
    net = build_net(Track4D, args, train=False)
    with torch.no_grad():
        out, tensors = run_backbone(net, rec, mu, d)
        flow, h_out, cls = tensors[0], tensors[1], tensors[2]
        out2, _ = run_backbone(net, rec, mu, d, h=h_out, capture=False)
        out["flow_step2"], out["h_out_step2"] = out2["flow"], out2["h_out"]
        pc1 = torch.from_numpy(d["pc1"])
        sf = ref_main_utils.eval_scene_flow(pc1, pc1 + flow, torch.from_numpy(d["gt_warp"]), torch.from_numpy(~d["gt_cls"]).float())
        out["metric_sf_keys"] = np.array(sorted(sf.keys()))
        out["metric_sf_vals"] = np.array([sf[k] for k in sorted(sf.keys())], dtype=np.float64)
    # train step on the same pair
    net = build_net(Track4D, args, train=True)
    tout, tensors = run_backbone(net, rec, mu, d, capture=False)
    flow, h_out, cls = tensors[0], tensors[1], tensors[2]
    pc1, pc2 = torch.from_numpy(d["pc1"]), torch.from_numpy(d["pc2"])
    total, items, keys = batch_mean_loss(ref_loss, pc1, pc2, pc1 + flow, cls, torch.from_numpy(d["gt_warp"]), torch.from_numpy(d["gt_cls"]), False)
    tr = {"flow": tout["flow"], "cls": tout["cls"], "h_out": tout["h_out"], "loss_keys": np.array(keys),
          "loss_vals": np.array([float(items[k]) for k in keys], dtype=np.float64)}
    net.zero_grad()
    total.backward()
    grad_records([(k, p.grad) for k, p in net.named_parameters()], tr)
    bn_records(net.state_dict(), tr)
    arbiter_train_step(d, tr, samples=arb_samples)
    out.update({"train/" + k: v for k, v in tr.items()})
    save(os.path.join(outdir, name + ".npz"), d, out)


_REF = {}

FORWARD_CLS_BIAS_SHIFT = 0.09


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    torch.set_num_threads(8)
    rec, Track4D, args, mu, ref_loss, ref_main_utils = import_reference()
    _REF.update(Track4D=Track4D, args=args)
    ref = (rec, Track4D, args, mu, ref_loss)

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
        # round 5: a training iteration with the tracking term (forward() in train mode, pretrain = False), gradients of every parameter
        "train_full_b1_n256": lambda: full_loss_case("train_full_b1_n256", Track4D, args, ref_loss, a.out),
        # round 3: 8 x 256 = 2 048 query points -> the split-bf16 training kernels; gradient tensors; float64 arbiter
        "train_b8_n256": lambda: train_batch_case("train_b8_n256", synth.make_frame_pairs(8, 256, 11), ref, a.out),
        # the reference's shipped radar frames, N = 322 / 352 / 242, every pair with N1 != N2
        "real_549_1047": lambda: real_case("real_549_1047", "00549", "01047", 0, ref, ref_main_utils, a.out),
        "real_1047_1201": lambda: real_case("real_1047_1201", "01047", "01201", 1, ref, ref_main_utils, a.out, arb_samples=False),
        "real_1201_549": lambda: real_case("real_1201_549", "01201", "00549", 2, ref, ref_main_utils, a.out, arb_samples=False),
    }
    for name, fn in cases.items():
        if a.only and a.only != name:
            continue
        fn()


if __name__ == "__main__":
    main()
