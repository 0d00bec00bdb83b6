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


if __name__ == "__main__" and "--affinity" not in sys.argv:
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
