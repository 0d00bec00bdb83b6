"""File formats at the two ends of the hot path.

* radar frames: View-of-Delft `.bin` files, float32 (N, 7) = x, y, z, RCS, v_r, v_r_compensated, time
  (reference: vod/frame/data_loader.py:164-180); `frame_pair_tensors` builds the (1,3,N) / (1,2,N) tensors the epoch loop
  feeds to the model (main_utils.py:75-79: xyz = columns 0:3, features = RCS and v_r = columns 3:5) and
  `compensate_ego_motion` is the homogeneous transform of dataset_classes/track_vod_3d.py:107-108.
* tracking results: one text file per frame, one line per tracked object,
      NA 1 -1 -1 <confidence> <track id> x0 y0 z0 x1 y1 z1 ...
  with the object's points taken from channels 3:6 of its (1, C, n) tensor (main_utils.py:165-184); consumed by the
  authors' AB3DMOT-style evaluation.  Numbers are written with Python's repr of the float32 value promoted to float, as the
  reference's str(float(t)) does, so files compare byte for byte.

Host-side code (numpy / file I/O); not part of the GPU path.  GT generation (tracking labels, oriented boxes, GT flow and object
mappings from the VoD label / calibration / pose files) is ratrack_amd/vod_gt.py.
"""
import os

import numpy as np
import torch


def load_radar_bin(path):
    """-> float32 (N, 7); raises FileNotFoundError like open() (the reference logs and returns None)."""
    scan = np.fromfile(path, dtype=np.float32)
    if scan.size % 7:
        raise ValueError("%s: %d float32 values is not a whole number of 7-column radar points" % (path, scan.size))
    return scan.reshape(-1, 7)


def save_radar_bin(path, scan):
    scan = np.ascontiguousarray(scan, dtype=np.float32)
    assert scan.ndim == 2 and scan.shape[1] == 7
    scan.tofile(path)


def frame_pair_tensors(scan_later, scan_earlier, device="cpu"):
    """(pc1 (1,3,N1), pc2 (1,3,N2), feature1 (1,2,N1), feature2 (1,2,N2)): pc1 is the LATER frame (t+1), pc2 the earlier (t)
    (dataset_classes/track_vod_3d.py:73-84,119; main_utils.py:68,75-78)."""
    def one(scan):
        t = torch.from_numpy(np.ascontiguousarray(scan[:, :6], dtype=np.float32)).unsqueeze(0).permute(0, 2, 1)
        return t[:, :3, :].contiguous().to(device), t[:, 3:5, :].contiguous().to(device)
    pc1, f1 = one(scan_later)
    pc2, f2 = one(scan_earlier)
    return pc1, pc2, f1, f2


def compensate_ego_motion(xyz, ego_motion):
    """xyz (N,3), ego_motion (4,4) -> (N,4) homogeneous points  [x y z 1] . inv(ego_motion^T)  (track_vod_3d.py:107-108)."""
    hom = np.hstack((xyz, np.ones((xyz.shape[0], 1))))
    return np.dot(hom, np.linalg.inv(ego_motion.T))


def format_track_line(obj_id, conf, obj):
    """obj (1, C>=6, n) tensor of one tracked object; its points are channels 3:6 (the un-warped coordinates)."""
    parts = ["NA", "1", "-1", "-1", str(float(conf)), str(obj_id)]
    for i in range(obj.size(2)):
        parts += [str(float(obj[0, 3, i])), str(float(obj[0, 4, i])), str(float(obj[0, 5, i]))]
    return " ".join(parts) + "\n"


def write_track_results(root, seq, index, objects, confs):
    """Writes <root>/<seq>/<index:05d>.txt; objects: {track id: (1,C,n) tensor} in association order, confs aligned with
    that order (main_utils.py:167-182).  Returns the path."""
    d = os.path.join(root, str(seq))
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, str(int(index)).zfill(5) + ".txt")
    with open(path, "w+") as f:
        for k, (obj_id, obj) in enumerate(objects.items()):
            f.write(format_track_line(obj_id, confs[k], obj))
    return path


def read_track_results(path):
    """-> list of (track id, confidence, points (n,3) float64) -- the inverse of write_track_results."""
    out = []
    with open(path) as f:
        for line in f:
            tok = line.split()
            if not tok:
                continue
            assert tok[:4] == ["NA", "1", "-1", "-1"], "not a RaTrack result line: %r" % line[:40]
            pts = np.array([float(v) for v in tok[6:]], dtype=np.float64).reshape(-1, 3)
            out.append((int(tok[5]), float(tok[4]), pts))
    return out
