"""Ground-truth generation from View-of-Delft labels and variable-N batching (hard part H7).

Host-side numpy / torch code, like the reference's (the epoch loop builds the GT on the CPU per frame, main_utils.py:86-124):

* calibration / pose files -> homogeneous transforms (vod/frame/transformations.py:16-283): `FrameTransforms`;
* KITTI-style tracking-label lines -> `Label` records (dataset_classes/kitti/kitti_trk_vod.py:8-48), the detection
  file's second column as the "moving" flag (models/utils/track4d_utils.py:581-590);
* oriented GT boxes in the radar frame (track4d_utils.py:528-546 `get_bbx_param`) and the points inside them
  (track4d_utils.py:105-176 `filter_object_points`: motion-segmentation labels, per-object point sets, rider+bicycle merge,
  minimum object size);
* GT scene flow (track4d_utils.py:337-359 `get_gt_flow_new`: rigid box-to-box motion for labelled points, ego-motion
  compensated position for the rest), GT-object to predicted-object mapping for the tracking loss (:74-102);
* `pad_frame_pairs`: real VoD frames have a different number of radar points each (242 / 322 / 352 in the shipped example
  set) and the reference only ever runs B = 1.  Clouds are padded to the batch maximum with COPIES OF THEIR POINT 0 and the
  true counts travel with the batch (`n_valid`): duplicates of point 0 are never picked by furthest point sampling before
  the cloud is exhausted, fill ball-query slots only where the reference fills them with the first hit (= point 0 itself,
  the scan is in index order), leave every max-pool unchanged, and are excluded from the kNN candidate set and from the FPS
  tie rule (which depends on the cloud size) by the kernels -- so the valid part of every padded sample equals its own B = 1
  run (tests/test_vod_gt_cpu.py, tests/test_fused_gpu.py::test_padded_variable_n_batch).

Third-party dependency of the reference that is absent here: Open3D 0.18.0 (src/environment.yml).  Its
`OrientedBoundingBox(center, R, extent).get_point_indices_within_bounding_box(points)` is restated from its published
implementation (cpp/open3d/geometry/BoundingVolume.cpp): with the box axes d_k = R e_k, a point p is inside iff
|(p - center) . d_k| <= extent_k / 2 for k = 0, 1, 2, evaluated in float64; indices are returned in ascending order.
`scipy.spatial.transform.Rotation.from_euler('XYZ', [0, 0, a])` (intrinsic rotations, only the z angle non-zero) is the
plain rotation about z by a.
"""
import json
import math
import random
from collections import namedtuple

import numpy as np
import torch

Label = namedtuple("Label", "type id occ alpha xmin ymin xmax ymax h w l x y z ry")
Box = namedtuple("Box", "center R extent")          # the three fields of open3d's OrientedBoundingBox the reference reads


# ---- calibration and poses ---------------------------------------------------------------------------------------------

class FrameTransforms:
    """The homogeneous transforms the GT code needs, from one frame's calibration and pose files
    (vod/frame/transformations.py: `get_sensor_transforms` :233-258 reads the intrinsic from line 3 and the sensor-to-camera
    extrinsic from line 6 of the KITTI calibration file, both as float32; `get_world_transform` :260-283 reads one JSON
    object per line of the pose file)."""

    def __init__(self, radar_calib_path, lidar_calib_path=None, pose_path=None):
        self.camera_projection_matrix, self.t_camera_radar = self._sensor(radar_calib_path)
        self.t_radar_camera = np.linalg.inv(self.t_camera_radar)
        if lidar_calib_path is not None:
            _, self.t_camera_lidar = self._sensor(lidar_calib_path)
            self.t_lidar_camera = np.linalg.inv(self.t_camera_lidar)
            self.t_lidar_radar = np.dot(self.t_lidar_camera, self.t_camera_radar)
            self.t_radar_lidar = np.dot(self.t_radar_camera, self.t_camera_lidar)
        if pose_path is not None:
            rows = [json.loads(line) for line in open(pose_path, "r")]
            self.t_odom_camera = np.array(rows[0]["odomToCamera"], dtype=np.float32).reshape(4, 4)
            self.t_map_camera = np.array(rows[1]["mapToCamera"], dtype=np.float32).reshape(4, 4)
            self.t_utm_camera = np.array(rows[2]["UTMToCamera"], dtype=np.float32).reshape(4, 4)

    @staticmethod
    def _sensor(path):
        with open(path, "r") as f:
            lines = f.readlines()
        intrinsic = np.array(lines[2].strip().split(" ")[1:], dtype=np.float32).reshape(3, 4)
        extrinsic = np.array(lines[5].strip().split(" ")[1:], dtype=np.float32).reshape(3, 4)
        return intrinsic, np.concatenate([extrinsic, [[0, 0, 0, 1]]], axis=0)


def ego_motion(tf_later, tf_earlier):
    """Radar-frame motion between two frames (dataset_classes/track_vod_3d.py:98-105): inv(odom<-radar(later)) . odom<-radar(earlier)."""
    odom_radar_0 = np.dot(tf_later.t_odom_camera, tf_later.t_camera_radar)
    odom_radar_1 = np.dot(tf_earlier.t_odom_camera, tf_earlier.t_camera_radar)
    return np.dot(np.linalg.inv(odom_radar_0), odom_radar_1)


# ---- labels ------------------------------------------------------------------------------------------------------------

def parse_tracking_labels(lines):
    """Tracking-label lines `type id occ alpha xmin ymin xmax ymax h w l x y z ry` -> {id: Label}, in file order
    (kitti_trk_vod.py:24-48; a later line with the same id replaces the earlier one, as the dict assignment does)."""
    out = {}
    for line in lines:
        tok = line.split(" ")
        if len(tok) < 15:
            continue
        rest = [float(x) for x in tok[1:15]]
        obj_id = int(tok[1])
        out[obj_id] = Label(tok[0], obj_id, rest[1], rest[2], rest[3], rest[4], rest[5], rest[6], rest[7], rest[8], rest[9],
                            rest[10], rest[11], rest[12], rest[13])
    return out


def filter_moving_labels(detection_lines, labels):
    """Keep the i-th label iff the i-th detection line's second column is 1 (track4d_utils.py:581-590: the detection file
    lists the same objects in the same order and carries the annotators' moving flag in the truncation column)."""
    keys = list(labels.keys())
    return {keys[i]: labels[keys[i]] for i, line in enumerate(detection_lines) if int(line.split(" ")[1]) == 1}


# ---- boxes -------------------------------------------------------------------------------------------------------------

def rot_z(angle):
    c, s = math.cos(angle), math.sin(angle)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def box_in_radar_frame(label, tf):
    """get_bbx_param(..., 'radar') (track4d_utils.py:528-546): centre = t_radar_camera . (x, y, z, 1); extent = (l, w, h);
    R = t_radar_lidar[:3,:3] . Rz(-(ry + pi/2))."""
    center = (tf.t_radar_camera @ np.array([label.x, label.y, label.z, 1]))[:3]
    extent = np.array([label.l, label.w, label.h], dtype=np.float64)
    R = tf.t_radar_lidar[:3, :3] @ rot_z(-(label.ry + np.pi / 2))
    return Box(np.asarray(center, dtype=np.float64), np.asarray(R, dtype=np.float64), extent)


def points_in_box(box, pts):
    """Ascending indices of the points (N,3) inside the oriented box (Open3D's OrientedBoundingBox semantics, see the module
    docstring): |(p - c) . (R e_k)| <= extent_k / 2 on all three axes, closed interval, float64."""
    d = np.asarray(pts, dtype=np.float64) - box.center
    proj = d @ box.R                                    # column k of R is the box axis d_k
    return np.nonzero((np.abs(proj) <= box.extent / 2).all(axis=1))[0]


def box_transform(box):
    """get_bbx_transformation (track4d_utils.py:549-556): the box pose as a 4x4 matrix."""
    t = np.zeros((4, 4))
    t[:3, :3] = box.R
    t[:3, 3] = box.center
    t[3, 3] = 1
    return t


# ---- per-frame ground truth -----------------------------------------------------------------------------------------------

def filter_object_points(min_obj_points, labels, pc, tf):
    """track4d_utils.py:105-176.  labels {id: Label} (moving objects), pc (1,3,N) tensor, tf FrameTransforms ->
    the reference's 10-tuple (pc_fil, cls, objs, objs_idx, objs_centre, cls_obj_id, boxes, objs_combined,
    objs_idx_combined, objs_centre_combined); tensors live on pc's device."""
    dev = pc.device
    N = pc.shape[2]
    boxes = {lab.id: box_in_radar_frame(lab, tf) for lab in labels.values()}
    pts = pc[0].detach().cpu().numpy().T
    cls = torch.zeros(N, dtype=torch.bool, device=dev)
    cls_obj_id = torch.full((N,), -1, dtype=torch.int64, device=dev)
    objs, objs_idx, objs_centre, parts = {}, {}, {}, []
    for obj_id, box in boxes.items():
        idx = points_in_box(box, pts)
        if len(idx) == 0:
            continue
        idx_t = torch.from_numpy(idx).to(dev)
        cls[idx_t] = True
        cls_obj_id[idx_t] = obj_id                       # a point inside two boxes keeps the later box's id (:132-133)
        objs[obj_id] = pc[:, :, idx_t]
        objs_centre[obj_id] = objs[obj_id].mean(dim=2)
        objs_idx[obj_id] = idx_t
        parts.append(objs[obj_id])
    pc_fil = torch.cat(parts, dim=2) if parts else None
    # riders are merged into the nearest other object (their bicycle), duplicates removed (:143-158)
    ids_to_pop = []
    for obj_id, c1 in objs_centre.items():
        if labels[obj_id].type != "rider":
            continue
        nearest, best = -1, float("inf")
        for other, c2 in objs_centre.items():
            if other == obj_id:
                continue
            dist = float((c1 - c2).pow(2).sum(dim=1).sqrt())
            if dist < best:
                best, nearest = dist, other
        if nearest == -1:
            continue
        ids_to_pop.append(obj_id)
        objs[nearest] = torch.unique(torch.cat((objs[obj_id], objs[nearest]), dim=2), dim=2)
    for obj_id, obj in objs.items():                      # GT objects with too few points are dropped (:160-163)
        if obj.size(2) < min_obj_points:
            ids_to_pop.append(obj_id)
    keep = [k for k in objs if k not in ids_to_pop]
    return (pc_fil, cls, objs, objs_idx, objs_centre, cls_obj_id, boxes, {k: objs[k] for k in keep},
            {k: objs_idx[k] for k in keep}, {k: objs_centre[k] for k in keep})


def gt_scene_flow(objs_centre2, cls1, cls_obj_id1, pc1, pc1_comp, boxes1, boxes2):
    """get_gt_flow_new (track4d_utils.py:337-359), vectorised per object: a labelled point of pc1 whose object also has points
    in the other frame moves rigidly with its box, T = T_box2 . inv(T_box1) (float64 product, float32 application as in the
    reference); every other point gets its ego-motion compensated position.  pc1 (1,3,N), pc1_comp (1,3,N) -> (1,3,N):
    GT WARPED POSITIONS (what the reference calls gt_flow)."""
    out = pc1_comp[:, :3, :].clone().to(torch.float32)
    ids = cls_obj_id1.to(pc1.device)
    for obj_id in torch.unique(ids[cls1.to(pc1.device)]).tolist():
        if obj_id not in objs_centre2:
            continue
        t = np.dot(box_transform(boxes2[obj_id]), np.linalg.inv(box_transform(boxes1[obj_id])))
        t = torch.tensor(t, dtype=torch.float32, device=pc1.device)
        sel = (ids == obj_id) & cls1.to(pc1.device)
        p = pc1[0, :3, sel]
        hom = torch.cat((p, torch.ones(1, p.shape[1], device=pc1.device)), dim=0)
        out[0, :, sel] = (t @ hom)[:3]
    return out


def iou_points(pred_pts, gt_pts):
    """track4d_utils.py:50-71: points of the predicted object (its un-warped coordinates, columns 3:6) that coincide
    (distance < 1e-5) with GT-object points, over the union."""
    a = np.asarray(pred_pts)[:, 3:6]
    b = np.asarray(gt_pts)
    total = a.shape[0] + b.shape[0]
    common = int((np.linalg.norm(a[:, None, :] - b[None, :, :], axis=2) < 0.00001).sum())
    return 0 if total - common == 0 else common / (total - common)


def map_gt_objects(gt_obj_centres, gt_objs, objects, rng=random):
    """track4d_utils.py:74-102: {gt id: predicted id}, {predicted id: gt id} by best point-IoU; unmatched predictions get a
    random negative key (drawn from `rng`, the module `random` in the reference)."""
    if len(gt_obj_centres) == 0:
        return {}, {}
    mapping, mapping_inv, used = {}, {}, []
    gt_key = None
    for key, obj in objects.items():
        best_iou, best_gt = 0, -1
        for gt_key, gt in gt_objs.items():
            iou = iou_points(obj[0].detach().cpu().numpy().T, gt[0].detach().cpu().numpy().T)
            if iou > best_iou:
                best_iou, best_gt = iou, gt_key
        if best_gt == -1 or best_gt in used:
            mapping[-rng.randint(99999, 9999999999999999)] = key
            mapping_inv[key] = -rng.randint(99999, 9999999999999999)
            continue
        mapping[best_gt] = key
        mapping_inv[key] = gt_key        # sic: the reference stores the LAST gt key of the inner loop (:100), not best_gt
        used.append(best_gt)
    return mapping, mapping_inv


# ---- one pass of the epoch loop, up to the network call ----------------------------------------------------------------------

FramePairGT = namedtuple("FramePairGT", "pc1 pc2 feature1 feature2 pc1_compensated gt_flow gt_cls gt_objs objs_idx objs_centre gt_mov_pts "
                                        "labels1 labels2")


def frame_pair_gt(later, earlier, min_obj_points=2, device="cpu"):
    """What the reference's epoch loop derives from the files of a frame pair before it calls the network (main_utils.py:66-122,
    dataset_classes/track_vod_3d.py:98-119): the clouds and features of both frames, the ego-motion compensated later cloud, the
    moving labels, the per-frame GT (`filter_object_points`; the loss and the mapping take the rider-merged, size-filtered object
    sets, main_utils.py:118-120) and the GT warped positions (`gt_scene_flow`).
    later / earlier: dict(radar=, radar_calib=, lidar_calib=, pose=, labels=) of file paths -- `labels` the detection-format label
    file whose second column is the moving flag -- + tracking=<list of tracking-format label lines of the frame>."""
    from . import vod_io
    tfs = [FrameTransforms(f["radar_calib"], f.get("lidar_calib"), f.get("pose")) for f in (later, earlier)]
    scans = [vod_io.load_radar_bin(f["radar"]) for f in (later, earlier)]
    pc1, pc2, f1, f2 = vod_io.frame_pair_tensors(scans[0], scans[1], device=device)
    comp = vod_io.compensate_ego_motion(scans[0][:, :3], ego_motion(tfs[0], tfs[1]))
    pc1_comp = torch.from_numpy(np.ascontiguousarray(comp[:, :3].T.astype(np.float32))).unsqueeze(0).to(device)
    labels = []
    for f in (later, earlier):
        with open(f["labels"], "r") as fh:
            labels.append(filter_moving_labels(fh.read().splitlines(), parse_tracking_labels(f["tracking"])))
    r1 = filter_object_points(min_obj_points, labels[0], pc1, tfs[0])
    r2 = filter_object_points(min_obj_points, labels[1], pc2, tfs[1])
    gt = gt_scene_flow(r2[4], r1[1], r1[5], pc1, pc1_comp, r1[6], r2[6])
    return FramePairGT(pc1, pc2, f1, f2, pc1_comp, gt, r1[1], r1[7], r1[8], r1[9], r1[0], labels[0], labels[1])


# ---- variable-N batches --------------------------------------------------------------------------------------

def pad_frame_pairs(pairs, device="cpu"):
    """pairs: list of (pc1 (1,3,N1), pc2 (1,3,N2), feature1 (1,2,N1), feature2 (1,2,N2)) as `vod_io.frame_pair_tensors` makes
    them.  -> pc1, pc2 (B,3,Nmax), feature1, feature2 (B,2,Nmax), n_valid (2,B) int32 (row 0: frame 1, row 1: frame 2).
    Padding columns repeat column 0 of the same cloud (coordinates AND features), see the module docstring."""
    nmax = max(max(p[0].shape[2], p[1].shape[2]) for p in pairs)

    def pad(t):
        n = t.shape[2]
        return t if n == nmax else torch.cat([t, t[:, :, :1].expand(-1, -1, nmax - n)], dim=2)
    cols = [torch.cat([pad(p[i]) for p in pairs], dim=0).contiguous().to(device) for i in range(4)]
    n_valid = torch.tensor([[p[0].shape[2] for p in pairs], [p[1].shape[2] for p in pairs]], dtype=torch.int32, device=device)
    return cols[0], cols[1], cols[2], cols[3], n_valid


def valid_mask(n_valid_row, nmax):
    """(B,) counts -> (B,nmax) bool."""
    return torch.arange(nmax, device=n_valid_row.device).unsqueeze(0) < n_valid_row.unsqueeze(1)
