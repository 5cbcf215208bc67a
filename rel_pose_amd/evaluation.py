"""Error metrics of the reference's evaluation scripts, as functions over arrays so they can be
tested without datasets: test_matterport.py:27-68 (`eval_camera`) and test_streetlearn_interiornet.py:26-122."""
import os

import numpy as np
from scipy.spatial.transform import Rotation

DEPTH_SCALE = 5          # test_matterport.py:25 (undoes matterport.py:44's division)


def matterport_prediction(pred7):
    """model output slot 1 (t/5, q xyzw) -> (t in metres [3], q wxyz [4]) as test_matterport.py:147-153 does"""
    p = np.array(pred7, dtype=np.float64, copy=True)
    p[3], p[6] = p[6], p[3]
    p[:3] *= DEPTH_SCALE
    return p[:3], p[3:]


def matterport_gt_rotation(q_wxyz):
    """ground truth with w >= 0 (test_matterport.py:140-145)"""
    q = np.array(q_wxyz, dtype=np.float64, copy=True)
    return -q if q[0] < 0 else q


def camera_metrics_matterport(pred_tran, pred_rot, gt_tran, gt_rot, out_dir=None):
    """top-1 accuracy at (1 m, 30 deg), mean / median errors; rotation error 2 acos |<q_pred, q_gt>| in degrees"""
    pred_tran, pred_rot = np.vstack(pred_tran), np.vstack(pred_rot)
    gt_tran, gt_rot = np.vstack(gt_tran), np.vstack(gt_rot)
    err_t = np.linalg.norm(gt_tran - pred_tran, axis=1)
    err_r = 2 * np.arccos(np.clip(np.abs(np.sum(pred_rot * gt_rot, axis=1)), -1.0, 1.0)) * 180 / np.pi
    metrics = {
        "top1 T err < 1.0": (err_t < 1.0).sum() / len(err_t) * 100,
        "top1 R err < 30": (err_r < 30).sum() / len(err_r) * 100,
        "T mean err": np.mean(err_t), "R mean err": np.mean(err_r),
        "T median err": np.median(err_t), "R median err": np.median(err_r),
    }
    if out_dir is not None:
        os.makedirs(out_dir, exist_ok=True)
        mag_t = np.linalg.norm(gt_tran, axis=1)
        mag_r = 2 * np.arccos(np.clip(gt_rot[:, 0], -1.0, 1.0)) * 180 / np.pi
        np.savetxt(os.path.join(out_dir, "gt_translation_magnitude_vs_error.csv"), np.stack([mag_t, err_t], 1), delimiter=",", fmt="%1.5f")
        np.savetxt(os.path.join(out_dir, "gt_rotation_magnitude_vs_error.csv"), np.stack([mag_r, err_r], 1), delimiter=",", fmt="%1.5f")
    return metrics


def _angle_deg(m):
    cos = np.clip((np.trace(m, axis1=1, axis2=2) - 1) / 2, -1.0, 1.0)
    return np.arccos(cos) * 180 / np.pi


def rotation_metrics_panorama(pred_quat, gt_quat, out_dir=None):
    """geodesic rotation error (degrees) split by the ground-truth angle: overlap_large < 45 deg <= overlap_small < 90 deg;
    mean / median / fraction within 10 deg per bucket (test_streetlearn_interiornet.py:71-122).  Quaternions xyzw."""
    r_pred = Rotation.from_quat(np.vstack(pred_quat)).as_matrix()
    r_gt = Rotation.from_quat(np.vstack(gt_quat)).as_matrix()
    err = _angle_deg(r_pred @ np.transpose(r_gt, (0, 2, 1)))
    gt_angle = _angle_deg(r_gt)
    buckets = {"rotation_geodesic_error_overlap_large": err[gt_angle < 45],
               "rotation_geodesic_error_overlap_small": err[(gt_angle >= 45) & (gt_angle < 90)]}
    if out_dir is not None:
        os.makedirs(out_dir, exist_ok=True)
        keep = gt_angle < 90
        np.savetxt(os.path.join(out_dir, "all_rotation_err_degrees.csv"), err[keep].astype(np.float32), delimiter=",", fmt="%1.5f")
        np.savetxt(os.path.join(out_dir, "all_gt_rot_degrees.csv"), gt_angle[keep].astype(np.float32), delimiter=",", fmt="%1.5f")
    out = {}
    for k, v in buckets.items():
        if v.size:
            out.update({k + "/mean": np.mean(v), k + "/median": np.median(v), k + "/10deg": (v <= 10).sum() / v.shape[0]})
    return out
