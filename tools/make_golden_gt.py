#!/usr/bin/env python3
"""Generate tests/golden/gt_example_set.npz: the reference's own GT machinery run (in the build container, CPU) on the three
View-of-Delft example frames shipped in the reference tree (copied as DATA files into tests/golden/vod_example/).

Pinned through the imported reference (import recipe of tools/make_golden.py):
  * vod/frame/transformations.py FrameTransformMatrix: t_camera_radar, t_radar_camera, t_radar_lidar, t_lidar_radar,
    t_odom_camera of every frame;
  * dataset_classes/kitti/kitti_trk_vod.py Tracklet_3D: parsing of tracking-label lines (synthesised from the shipped
    detection labels: `type id occ alpha bbox h w l x y z ry`, ids 0..);
  * models/utils/track4d_utils.py filter_moving_boxes_det, get_bbx_transformation, get_gt_flow_new, map_gt_objects,
    iou_points.
NOT pinned by the reference: Open3D's OrientedBoundingBox (open3d 0.18.0 is absent; see ratrack_amd/vod_gt.py) -- the boxes
handed to get_gt_flow_new are namespace objects carrying the (center, R) ratrack_amd.vod_gt computes.
"""
import os
import random
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden as MG  # noqa: E402
from ratrack_amd import vod_gt, vod_io  # noqa: E402

EX = os.path.join(ROOT, "tests", "golden", "vod_example")
FRAMES = ["00549", "01047", "01201"]
REF_EX = os.path.join(MG.REF_SRC, "dataset_classes", "vod", "example_set")


def tracking_lines(frame):
    """Detection label lines -> tracking-format lines with ids in file order."""
    out = []
    for i, line in enumerate(open(os.path.join(EX, "label_%s.txt" % frame)).read().splitlines()):
        t = line.split(" ")
        out.append(" ".join([t[0], str(i)] + t[2:15]))
    return out


def main():
    MG.import_reference()
    from vod.configuration import KittiLocations
    from vod.frame import FrameDataLoader, FrameTransformMatrix
    import models.utils.track4d_utils as TU
    from dataset_classes.kitti.kitti_trk_vod import Tracklet_3D
    out = {}
    loc = KittiLocations(root_dir=REF_EX, output_dir="/tmp")
    tfs, labels = {}, {}
    for f in FRAMES:
        fd = FrameDataLoader(kitti_locations=loc, frame_number=f)
        tf = FrameTransformMatrix(fd)
        for name in ("t_camera_radar", "t_radar_camera", "t_radar_lidar", "t_lidar_radar", "t_odom_camera"):
            out["%s/%s" % (f, name)] = np.asarray(getattr(tf, name), dtype=np.float64)
        tfs[f] = tf
        lines = tracking_lines(f)
        trk = Tracklet_3D(lines, int(f))
        lab = trk.data[int(f)]
        out["%s/label_ids" % f] = np.array(list(lab.keys()))
        out["%s/label_vals" % f] = np.array([[o.h, o.w, o.l, o.x, o.y, o.z, o.ry] for o in lab.values()], dtype=np.float64)
        det = open(os.path.join(EX, "label_%s.txt" % f)).read().splitlines()
        mov = TU.filter_moving_boxes_det(det, lab)
        out["%s/moving_ids" % f] = np.array(list(mov.keys()))
        labels[f] = mov
    # ---- GT flow: frame 00549 as the later frame, a displaced copy of its boxes as the earlier one ----------------------------
    f = FRAMES[0]
    my_tf = vod_gt.FrameTransforms(os.path.join(EX, "radar_calib_%s.txt" % f), os.path.join(EX, "lidar_calib_%s.txt" % f),
                                   os.path.join(EX, "pose_%s.json" % f))
    my_labels = vod_gt.filter_moving_labels(open(os.path.join(EX, "label_%s.txt" % f)).read().splitlines(),
                                            vod_gt.parse_tracking_labels(tracking_lines(f)))
    scan = vod_io.load_radar_bin(os.path.join(EX, "radar_%s.bin" % f))
    pc1 = torch.from_numpy(np.ascontiguousarray(scan[:, :3].T)).unsqueeze(0)
    res1 = vod_gt.filter_object_points(2, my_labels, pc1, my_tf)
    rng = np.random.default_rng(0)
    labels2 = {k: v._replace(x=v.x + float(rng.normal(0, 0.5)), z=v.z + float(rng.normal(0, 0.5)), ry=v.ry + float(rng.normal(0, 0.05)))
               for k, v in my_labels.items()}
    res2 = vod_gt.filter_object_points(2, labels2, pc1, my_tf)
    ns = lambda b: types.SimpleNamespace(R=b.R, center=b.center)
    boxes1 = {k: ns(b) for k, b in res1[6].items()}
    boxes2 = {k: ns(b) for k, b in res2[6].items()}
    pc1_comp = pc1 + 0.25
    ref_flow = TU.get_gt_flow_new(res1[4], res2[4], res1[1], res1[5], res2[5], pc1, pc1_comp, boxes1, boxes2)
    out["flow/labels2"] = np.array([[v.x, v.z, v.ry] for v in labels2.values()], dtype=np.float64)
    out["flow/gt"] = ref_flow.numpy()
    out["flow/cls"] = res1[1].numpy()
    # ---- GT-object mapping -----------------------------------------------------------------------------------------------------
    objs = {k: torch.cat([v, v, v[:, :1].expand(-1, 3, -1)], dim=1)[:, :9] for k, v in list(res1[7].items())}      # predicted objects: (1, >=6, n), channels 3:6 = points
    objs = {100 + i: o for i, (k, o) in enumerate(objs.items())}
    objs[999] = torch.randn(1, 9, 4, generator=torch.Generator().manual_seed(1))               # matches nothing
    random.seed(7)
    m, minv = TU.map_gt_objects(res1[9], res1[7], objs)
    out["map/keys"] = np.array(list(m.keys()), dtype=np.float64)
    out["map/vals"] = np.array(list(m.values()))
    out["map/inv_keys"] = np.array(list(minv.keys()))
    out["map/inv_vals"] = np.array(list(minv.values()), dtype=np.float64)
    path = os.path.join(ROOT, "tests", "golden", "gt_example_set.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays; moving objects in frame %s: %d, labelled points %d / %d" % (
        f, len(my_labels), int(res1[1].sum()), pc1.shape[2]))


if __name__ == "__main__" and "--affinity" not in sys.argv and "--train" not in sys.argv:
    main()


def affinity_loss_golden():
    """tests/golden/affinity_loss.npz: losses/loss.py:48-72 affinity_loss on three mapping configurations + the empty case."""
    _, _, _, _, ref_loss, _ = MG.import_reference()
    g = torch.Generator().manual_seed(3)
    out = {}
    for ci, (prev, curr) in enumerate([([5, 7, 9], [7, 5, 11, 9]), ([1], [2, 3]), ([4, 4.5, 6], [6, 4])]):
        mp = {k: 100 + i for i, k in enumerate(prev)}
        mc = {k: 200 + i for i, k in enumerate(curr)}
        aff = torch.rand(len(prev) * len(curr), generator=g) * 0.98 + 0.01
        out["aff%d/prev" % ci], out["aff%d/curr" % ci] = np.array(prev, dtype=np.float64), np.array(curr, dtype=np.float64)
        out["aff%d/aff" % ci], out["aff%d/val" % ci] = aff.numpy(), np.float64(float(ref_loss.affinity_loss(mp, mc, aff)))
    out["aff_empty"] = np.float64(float(ref_loss.affinity_loss({}, {1: 2}, torch.rand(0))))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "affinity_loss.npz"), **out)


if __name__ == "__main__" and "--affinity" in sys.argv:
    affinity_loss_golden()


# ---- round 5: the shipped frames -> labels -> GT -> one training iteration after pre-training ----------------------------------------
GT_TRAIN_PAIRS = [("01047", "01201"), ("00549", "01047")]       # (later = pc1, earlier = pc2): iteration 0 supplies objects_prev / mappings_prev


def frame_files(f):
    return dict(radar=os.path.join(EX, "radar_%s.bin" % f), radar_calib=os.path.join(EX, "radar_calib_%s.txt" % f),
                lidar_calib=os.path.join(EX, "lidar_calib_%s.txt" % f), pose=os.path.join(EX, "pose_%s.json" % f),
                labels=os.path.join(EX, "label_%s.txt" % f), tracking=tracking_lines(f))


def frame_gt(later, earlier, ref_tu):
    """What one pass of the reference's epoch loop derives from the files of a frame pair before it calls the network
    (main_utils.py:66-122), with the labels through the reference's Tracklet_3D / filter_moving_boxes_det and the GT flow through
    its get_gt_flow_new (boxes handed over as (center, R) namespaces -- Open3D is absent, the point-in-box test is
    ratrack_amd.vod_gt's).  The product's own path is ratrack_amd.vod_gt.frame_pair_gt."""
    from dataset_classes.kitti.kitti_trk_vod import Tracklet_3D
    tf1, tf2 = (vod_gt.FrameTransforms(os.path.join(EX, "radar_calib_%s.txt" % f), os.path.join(EX, "lidar_calib_%s.txt" % f),
                                       os.path.join(EX, "pose_%s.json" % f)) for f in (later, earlier))
    a = vod_io.load_radar_bin(os.path.join(EX, "radar_%s.bin" % later))
    b = vod_io.load_radar_bin(os.path.join(EX, "radar_%s.bin" % earlier))
    pc1, pc2, f1, f2 = vod_io.frame_pair_tensors(a, b)
    comp = vod_io.compensate_ego_motion(a[:, :3], vod_gt.ego_motion(tf1, tf2))
    pc1_comp = torch.from_numpy(np.ascontiguousarray(comp[:, :3].T.astype(np.float32))).unsqueeze(0)
    lab = {}
    for f in (later, earlier):
        det = open(os.path.join(EX, "label_%s.txt" % f)).read().splitlines()
        ref_lab = ref_tu.filter_moving_boxes_det(det, Tracklet_3D(tracking_lines(f), int(f)).data[int(f)])
        mine = vod_gt.filter_moving_labels(det, vod_gt.parse_tracking_labels(tracking_lines(f)))
        assert list(ref_lab.keys()) == list(mine.keys())
        lab[f] = mine                # (same ids and values: tests/test_vod_gt_cpu.py test_labels_match_reference)
    r1 = vod_gt.filter_object_points(2, lab[later], pc1, tf1)
    r2 = vod_gt.filter_object_points(2, lab[earlier], pc2, tf2)
    ns = lambda bx: types.SimpleNamespace(R=bx.R, center=bx.center)
    gt = ref_tu.get_gt_flow_new(r1[4], r2[4], r1[1], r1[5], r2[5], pc1, pc1_comp, {k: ns(v) for k, v in r1[6].items()},
                                {k: ns(v) for k, v in r2[6].items()})
    return dict(pc1=pc1, pc2=pc2, f1=f1, f2=f2, pc1_comp=pc1_comp, gt=gt.float(), gt_cls=r1[1], gt_objs=r1[7], objs_idx=r1[8], objs_centre=r1[9],
                gt_mov_pts=r1[0], lbl1=lab[later], lbl2=lab[earlier])


def gt_train_case():
    """tests/golden/train_gt_real.npz: TWO consecutive passes of the reference's epoch loop after pre-training (main_utils.py:66-156)
    on the shipped frames -- files -> labels -> GT flow / gt_cls / GT objects -> net.train() forward() -> map_gt_objects ->
    track_4d_loss(pretrain=False) -> backward -- the second pass with the first one's objects and mappings as objects_prev /
    mappings_prev.  Recorded: the GT tensors, the mappings' keys, the loss items and every parameter's gradient of the second pass.
    (The three shipped frames are not consecutive in time: the ego motion between them is tens of metres, so the GT warped
    positions are far from the cloud and the scene-flow term is large; the arithmetic is the epoch loop's.)"""
    rec, Track4D, args, mu, ref_loss, _ = MG.import_reference()
    import models.utils.track4d_utils as TU
    net = MG.build_net(Track4D, args, train=True)
    with torch.no_grad():
        net.state_dict()["fd_layer.cp.linear.bias"].add_(MG.FORWARD_CLS_BIAS_SHIFT)
    out = {}
    objects_prev, mappings_prev, h = dict(), dict(), None
    for it, (later, earlier) in enumerate(GT_TRAIN_PAIRS):
        g = frame_gt(later, earlier, TU)
        mine = vod_gt.frame_pair_gt(frame_files(later), frame_files(earlier))
        print("pass %d (%s, %s): %d / %d GT moving points, GT objects %s, |gt - pc1| max %.1f m, mine vs reference GT: %.2e"
              % (it, later, earlier, int(g["gt_cls"].sum()), g["gt_cls"].numel(), sorted(g["gt_objs"].keys()),
                 float((g["gt"] - g["pc1"]).abs().max()), float((g["gt"] - mine.gt_flow).abs().max())))
        if h is None:
            h = torch.zeros(5, 1, 128)
        h, pc1_warp, cls, aff_list, aff_mat, assig, confs, objects, _, objects_curr = net(g["pc1"], g["pc2"], g["f1"], g["f2"], h, objects_prev)
        random.seed(100 + it)
        mappings_curr, mappings_inv = TU.map_gt_objects(g["objs_centre"], g["gt_objs"], objects)
        total, items = ref_loss.track_4d_loss(objects_prev, objects, mappings_prev, mappings_curr, mappings_inv, g["lbl1"], g["lbl2"], g["pc1"],
                                              g["pc2"], pc1_warp, cls, g["gt"], aff_list, g["gt_mov_pts"], g["gt_cls"], g["gt_objs"], g["objs_idx"],
                                              g["objs_centre"], pretrain=False)
        keys = ["Loss", "SceneFlowLoss", "TrackingLoss", "SegLoss"]
        pre = "p%d_" % it
        out[pre + "gt"], out[pre + "gt_cls"], out[pre + "pc1_comp"] = g["gt"].numpy(), g["gt_cls"].numpy(), g["pc1_comp"].numpy()
        out[pre + "gt_obj_ids"] = np.array(list(g["gt_objs"].keys()), dtype=np.int64)
        out[pre + "object_ids"] = np.array(list(objects.keys()), dtype=np.int64)
        out[pre + "object_sizes"] = np.array([objects[k].shape[2] for k in objects], dtype=np.int64)
        out[pre + "map_keys"] = np.array(list(mappings_curr.keys()), dtype=np.float64)
        out[pre + "map_vals"] = np.array(list(mappings_curr.values()), dtype=np.int64)
        out[pre + "loss_vals"] = np.array([float(items[k]) for k in keys], dtype=np.float64)
        out[pre + "cls"], out[pre + "pc1_warp"] = MG.npf(cls), MG.npf(pc1_warp)
        out[pre + "aff_list"] = MG.npf(aff_list) if torch.is_tensor(aff_list) else np.zeros(0, np.float32)
        print("   objects %s (prev %d), mapping keys %s, losses %s" % (list(objects.keys()), len(objects_prev), list(mappings_curr.keys()),
                                                                    [round(float(items[k]), 5) for k in keys]))
        if it == len(GT_TRAIN_PAIRS) - 1:
            assert float(items["TrackingLoss"]) > 0 and len(mappings_prev) > 0
            net.zero_grad()
            total.backward()
            MG.grad_records([(k, p.grad) for k, p in net.named_parameters()], out)
        objects_prev = {k: v.clone().detach() for k, v in objects.items()}
        mappings_prev = mappings_curr
        h = h.detach()
    out["loss_keys"] = np.array(keys)
    path = os.path.join(ROOT, "tests", "golden", "train_gt_real.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__" and "--train" in sys.argv:
    gt_train_case()
