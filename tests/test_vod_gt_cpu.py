"""CPU: ground-truth generation and variable-N batching (ratrack_amd/vod_gt.py) on the three
View-of-Delft example frames the reference ships (data files copied to tests/golden/vod_example/), against
tests/golden/gt_example_set.npz = the reference's own transformation / label / GT-flow / mapping code run on the same
files (tools/make_golden_gt.py).  The oriented-box membership test restates Open3D 0.18 (absent here): it is checked against
its definition only, on constructed boxes."""
import os
import random
import types

import numpy as np
import torch

from _util import GOLDEN
from ratrack_amd import vod_gt, vod_io

EX = os.path.join(GOLDEN, "vod_example")
FRAMES = ["00549", "01047", "01201"]
G = dict(np.load(os.path.join(GOLDEN, "gt_example_set.npz")))


def _tf(f):
    return vod_gt.FrameTransforms(os.path.join(EX, "radar_calib_%s.txt" % f), os.path.join(EX, "lidar_calib_%s.txt" % f),
                                  os.path.join(EX, "pose_%s.json" % f))


def _tracking_lines(f):
    out = []
    for i, line in enumerate(open(os.path.join(EX, "label_%s.txt" % f)).read().splitlines()):
        t = line.split(" ")
        out.append(" ".join([t[0], str(i)] + t[2:15]))
    return out


def _det_lines(f):
    return open(os.path.join(EX, "label_%s.txt" % f)).read().splitlines()


def test_transforms_match_reference():
    for f in FRAMES:
        tf = _tf(f)
        for name in ("t_camera_radar", "t_radar_camera", "t_radar_lidar", "t_lidar_radar", "t_odom_camera"):
            np.testing.assert_allclose(getattr(tf, name), G["%s/%s" % (f, name)], rtol=0, atol=0, err_msg="%s %s" % (f, name))
    e = vod_gt.ego_motion(_tf(FRAMES[1]), _tf(FRAMES[0]))
    assert e.shape == (4, 4) and abs(np.linalg.det(e[:3, :3]) - 1) < 1e-4


def test_labels_match_reference():
    for f in FRAMES:
        lab = vod_gt.parse_tracking_labels(_tracking_lines(f))
        assert list(lab.keys()) == G["%s/label_ids" % f].tolist()
        vals = np.array([[o.h, o.w, o.l, o.x, o.y, o.z, o.ry] for o in lab.values()])
        assert np.array_equal(vals, G["%s/label_vals" % f])
        mov = vod_gt.filter_moving_labels(_det_lines(f), lab)
        assert list(mov.keys()) == G["%s/moving_ids" % f].tolist()


def test_points_in_box_definition():
    """Open3D semantics: closed slab test along the three box axes."""
    R = vod_gt.rot_z(0.3)
    box = vod_gt.Box(np.array([1.0, 2.0, 3.0]), R, np.array([4.0, 2.0, 1.0]))
    local = np.array([[0, 0, 0], [2.0, 1.0, 0.5], [2.0001, 0, 0], [-2.0, -1.0, -0.5], [0, 1.0001, 0], [0, 0, -0.5001], [1.9, 0.9, 0.4]])
    pts = local @ R.T + box.center
    assert vod_gt.points_in_box(box, pts).tolist() == [0, 1, 3, 6]
    lab = vod_gt.Label("Car", 0, 0, 0, 0, 0, 0, 0, 1.5, 2.0, 4.0, 1.0, 1.0, 10.0, 0.2)
    b = vod_gt.box_in_radar_frame(lab, _tf(FRAMES[0]))
    assert np.allclose(b.R.T @ b.R, np.eye(3), atol=1e-5) and b.extent.tolist() == [4.0, 2.0, 1.5]


def test_filter_object_points_on_example_frame():
    f = FRAMES[0]
    tf = _tf(f)
    labels = vod_gt.filter_moving_labels(_det_lines(f), vod_gt.parse_tracking_labels(_tracking_lines(f)))
    scan = vod_io.load_radar_bin(os.path.join(EX, "radar_%s.bin" % f))
    pc = torch.from_numpy(np.ascontiguousarray(scan[:, :3].T)).unsqueeze(0)
    pc_fil, cls, objs, objs_idx, objs_centre, cls_obj_id, boxes, objs_c, idx_c, centre_c = vod_gt.filter_object_points(2, labels, pc, tf)
    assert np.array_equal(cls.numpy(), G["flow/cls"]) and int(cls.sum()) == 40
    assert set(objs) >= set(objs_c) and all(o.shape[2] >= 2 for o in objs_c.values())
    for k, idx in objs_idx.items():
        assert torch.equal(objs[k], pc[:, :, idx]) or labels[k].type == "rider" or any(labels[j].type == "rider" for j in objs)
        assert torch.allclose(objs_centre[k], pc[:, :, idx].mean(dim=2))
    # every labelled point is inside the box whose id it carries; unlabelled points are in no box
    pts = pc[0].numpy().T
    for i in range(pts.shape[0]):
        inside = [k for k, b in boxes.items() if i in vod_gt.points_in_box(b, pts)]
        assert bool(cls[i]) == bool(inside)
        if inside:
            assert int(cls_obj_id[i]) == inside[-1]
    assert pc_fil.shape[2] == sum(len(v) for v in objs_idx.values())
    # rider merge + minimum size on a constructed case
    labs = {0: vod_gt.Label("rider", 0, 0, 0, 0, 0, 0, 0, 2.0, 1.0, 1.0, 0.0, 0.0, 10.0, 0.0),
            1: vod_gt.Label("bicycle", 1, 0, 0, 0, 0, 0, 0, 2.0, 1.0, 2.0, 0.3, 0.0, 10.0, 0.0),
            2: vod_gt.Label("Car", 2, 0, 0, 0, 0, 0, 0, 2.0, 2.0, 4.0, 8.0, 0.0, 20.0, 0.0)}
    cam = np.array([[0.0, 0.0, 10.0, 1], [0.2, 0.0, 10.0, 1], [0.9, 0.0, 10.0, 1], [8.0, 0.0, 20.0, 1], [30.0, 0, 30, 1]])
    radar = (tf.t_radar_camera @ cam.T)[:3].astype(np.float32)
    res = vod_gt.filter_object_points(2, labs, torch.from_numpy(radar).unsqueeze(0), tf)
    assert 0 in res[2] and 0 not in res[7]                 # the rider is merged away ...
    assert res[7][1].shape[2] == 3                          # ... into the bicycle (points 0,1 shared + point 2), duplicates removed
    assert 2 not in res[7]                                  # one point only: below min_obj_points
    assert res[1].tolist() == [True, True, True, True, False]


def test_gt_scene_flow_matches_reference():
    f = FRAMES[0]
    tf = _tf(f)
    labels = vod_gt.filter_moving_labels(_det_lines(f), vod_gt.parse_tracking_labels(_tracking_lines(f)))
    scan = vod_io.load_radar_bin(os.path.join(EX, "radar_%s.bin" % f))
    pc1 = torch.from_numpy(np.ascontiguousarray(scan[:, :3].T)).unsqueeze(0)
    res1 = vod_gt.filter_object_points(2, labels, pc1, tf)
    l2 = G["flow/labels2"]
    labels2 = {k: v._replace(x=float(l2[i, 0]), z=float(l2[i, 1]), ry=float(l2[i, 2])) for i, (k, v) in enumerate(labels.items())}
    res2 = vod_gt.filter_object_points(2, labels2, pc1, tf)
    gt = vod_gt.gt_scene_flow(res2[4], res1[1], res1[5], pc1, pc1 + 0.25, res1[6], res2[6])
    np.testing.assert_allclose(gt.numpy(), G["flow/gt"], rtol=1e-6, atol=1e-6)
    moved = (gt - (pc1 + 0.25)).abs().sum(1)[0] > 1e-6
    assert moved.any() and not moved[~res1[1]].any()        # only labelled points follow their box


def test_map_gt_objects_matches_reference():
    f = FRAMES[0]
    labels = vod_gt.filter_moving_labels(_det_lines(f), vod_gt.parse_tracking_labels(_tracking_lines(f)))
    scan = vod_io.load_radar_bin(os.path.join(EX, "radar_%s.bin" % f))
    pc1 = torch.from_numpy(np.ascontiguousarray(scan[:, :3].T)).unsqueeze(0)
    res1 = vod_gt.filter_object_points(2, labels, pc1, _tf(f))
    objs = {100 + i: torch.cat([v, v, v[:, :1].expand(-1, 3, -1)], dim=1)[:, :9] for i, (k, v) in enumerate(res1[7].items())}
    objs[999] = torch.randn(1, 9, 4, generator=torch.Generator().manual_seed(1))
    random.seed(7)
    m, minv = vod_gt.map_gt_objects(res1[9], res1[7], objs)
    assert np.array_equal(np.array(list(m.keys()), dtype=np.float64), G["map/keys"])
    assert np.array_equal(np.array(list(m.values())), G["map/vals"])
    assert np.array_equal(np.array(list(minv.keys())), G["map/inv_keys"])
    assert np.array_equal(np.array(list(minv.values()), dtype=np.float64), G["map/inv_vals"])


def test_pad_frame_pairs():
    scans = [vod_io.load_radar_bin(os.path.join(EX, "radar_%s.bin" % f)) for f in FRAMES]
    assert sorted(s.shape[0] for s in scans) == [242, 322, 352]
    pairs = [vod_io.frame_pair_tensors(scans[i], scans[(i + 1) % 3]) for i in range(3)]
    pc1, pc2, f1, f2, nv = vod_gt.pad_frame_pairs(pairs)
    assert pc1.shape == (3, 3, 352) and f2.shape == (3, 2, 352) and nv.tolist() == [[322, 352, 242], [352, 242, 322]]
    for b, p in enumerate(pairs):
        n1 = p[0].shape[2]
        assert torch.equal(pc1[b, :, :n1], p[0][0]) and torch.equal(f1[b, :, :n1], p[2][0])
        assert (pc1[b, :, n1:] == p[0][0, :, :1]).all() and (f1[b, :, n1:] == p[2][0, :, :1]).all()
    m = vod_gt.valid_mask(nv[0], 352)
    assert m.sum(1).tolist() == [322, 352, 242]


def test_frame_pair_gt_matches_the_reference_epoch_loop_preamble():
    """vod_gt.frame_pair_gt (files of a frame pair -> everything the epoch loop hands to the network, the mapping and the loss)
    against tests/golden/train_gt_real.npz: the reference's get_gt_flow_new on the same files (tools/make_golden_gt.py --train)."""
    g = np.load(os.path.join(GOLDEN, "train_gt_real.npz"), allow_pickle=False)
    frame = lambda f: dict(radar=os.path.join(EX, "radar_%s.bin" % f), radar_calib=os.path.join(EX, "radar_calib_%s.txt" % f),
                           lidar_calib=os.path.join(EX, "lidar_calib_%s.txt" % f), pose=os.path.join(EX, "pose_%s.json" % f),
                           labels=os.path.join(EX, "label_%s.txt" % f), tracking=_tracking_lines(f))
    for it, (later, earlier) in enumerate([("01047", "01201"), ("00549", "01047")]):
        r = vod_gt.frame_pair_gt(frame(later), frame(earlier))
        pre = "p%d_" % it
        assert r.pc1.shape[2] != r.pc2.shape[2]
        assert np.array_equal(r.gt_cls.numpy(), g[pre + "gt_cls"]) and 0 < int(r.gt_cls.sum()) < r.gt_cls.numel()
        assert np.abs(r.pc1_compensated.numpy() - g[pre + "pc1_comp"]).max() <= 2e-5
        assert np.abs(r.gt_flow.numpy() - g[pre + "gt"]).max() <= 2e-5
        assert list(r.gt_objs.keys()) == g[pre + "gt_obj_ids"].tolist()
        assert set(r.objs_idx) == set(r.gt_objs) == set(r.objs_centre)
