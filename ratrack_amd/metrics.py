"""Scene-flow and motion-segmentation metrics (reference: main_utils.py:272-389), numpy on the host."""
import numpy as np


def _cartesian_res(pc, sensor):
    """Per-point x/y/z measurement resolution from range/elevation/azimuth resolution
    (main_utils.py:272-311).  pc (B,3,N) numpy."""
    if sensor == "radar":      # LRR30
        res = np.array([0.2, 1.0 * np.pi / 180, 1.6 * np.pi / 180])
    else:                      # lidar, HDL-64E
        res = np.array([0.04, 0.4 * np.pi / 180, 0.08 * np.pi / 180])
    x, y, z = pc[:, 0], pc[:, 1], pc[:, 2]
    r = np.sqrt(x ** 2 + y ** 2 + z ** 2)
    theta = np.arcsin(z / r)
    phi = np.arctan2(y, x)
    gx = np.stack((np.cos(phi) * np.cos(theta), -r * np.sin(theta) * np.cos(phi), -r * np.cos(theta) * np.sin(phi)), axis=2)
    gy = np.stack((np.sin(phi) * np.cos(theta), -r * np.sin(phi) * np.sin(theta), r * np.cos(theta) * np.cos(phi)), axis=2)
    gz = np.stack((np.sin(theta), r * np.cos(theta), np.zeros_like(x)), axis=2)
    return np.stack((np.sum(abs(gx) * res, axis=2), np.sum(abs(gy) * res, axis=2), np.sum(abs(gz) * res, axis=2)), axis=2)


def eval_scene_flow(pc, pred, labels, mask):
    """EPE, resolution-normalised error (RNE), strict/relaxed accuracy (main_utils.py:342-374).
    pc, pred (= warped points), labels (B,3,N); mask (B,N) or (N,): 1 = static, 0 = moving."""
    to_np = lambda t: t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)
    pc, pred, labels, mask = to_np(pc), to_np(pred), to_np(labels), to_np(mask)
    if mask.ndim == 2:
        mask = mask[0]
    error = np.sqrt(np.sum((pred - labels) ** 2, 1) + 1e-20)
    epe = np.mean(error)
    gt_len = np.sqrt(np.sum(labels * labels, 1) + 1e-20)
    res_r = np.sqrt(np.sum(_cartesian_res(pc, "radar"), 2) + 1e-20)
    res_l = np.sqrt(np.sum(_cartesian_res(pc, "lidar"), 2) + 1e-20)
    rn_error = error / (res_r / res_l)
    rne = np.mean(rn_error)
    mov_rne = np.sum(rn_error[:, mask == 0]) / (np.sum(mask == 0) + 1e-6)
    stat_rne = np.mean(rn_error[:, mask == 1])
    count = np.size(pred, 0) * np.size(pred, 2)
    sas = np.sum(np.logical_or(rn_error <= 0.10, rn_error / gt_len <= 0.10)) / count
    ras = np.sum(np.logical_or(rn_error <= 0.20, rn_error / gt_len <= 0.20)) / count
    return {"rne": rne, "50-50 rne": (mov_rne + stat_rne) / 2, "mov_rne": mov_rne, "stat_rne": stat_rne,
            "sas": sas, "ras": ras, "epe": epe}


def eval_motion_seg(pre, gt):
    """accuracy, mIoU, sensitivity (main_utils.py:377-389)."""
    to_np = lambda t: t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)
    pre, gt = to_np(pre), to_np(gt)
    tp = np.logical_and(pre == 1, gt == 1).sum() + 1e-20
    tn = np.logical_and(pre == 0, gt == 0).sum() + 1e-20
    fp = np.logical_and(pre == 1, gt == 0).sum() + 1e-20
    fn = np.logical_and(pre == 0, gt == 1).sum() + 1e-20
    return {"acc": (tp + tn) / (tp + tn + fp + fn), "sen": tp / (tp + fn),
            "miou": 0.5 * (tp / (tp + fp + fn + 1e-4) + tn / (tn + fp + fn + 1e-4))}
