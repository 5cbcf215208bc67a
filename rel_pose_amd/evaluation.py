"""Error metrics of the reference's evaluation scripts, as functions over arrays so they can be
tested without datasets: test_matterport.py:27-68 (`eval_camera`) and test_streetlearn_interiornet.py:26-122."""
import os

import numpy as np
from scipy.spatial.transform import Rotation

DEPTH_SCALE = 5          # test_matterport.py:25 (undoes matterport.py:44's division)


def matterport_prediction(pred7):
    """model output slot 1 (t/5, q xyzw) -> (t in metres [3], q wxyz [4]) as test_matterport.py:147-153 does -- in the
    array's own dtype (the reference scales the float32 model output in place; pinned by tests/golden/reference_metrics.npz)"""
    p = np.array(pred7, copy=True)
    p[3], p[6] = pred7[6], pred7[3]
    p[:3] = p[:3] * DEPTH_SCALE
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
        with np.errstate(invalid="ignore"):
            mag_r = 2 * np.arccos(gt_rot[:, 0]) * 180 / np.pi         # unclipped like test_matterport.py:58: w > 1 by rounding -> nan
        np.savetxt(os.path.join(out_dir, "gt_translation_magnitude_vs_error.csv"), np.stack([mag_t, err_t], 1), delimiter=",", fmt="%1.5f")
        np.savetxt(os.path.join(out_dir, "gt_rotation_magnitude_vs_error.csv"), np.stack([mag_r, err_r], 1), delimiter=",", fmt="%1.5f")
    return metrics


def _angle_rad(m):
    """rotation angle of [n,3,3] float64 torch matrices: acos of the clamped (trace - 1) / 2, summed in the reference's order
    (test_streetlearn_interiornet.py:26-35) so that results agree to the last bit with its torch arithmetic"""
    import torch
    cos = (m[:, 0, 0] + m[:, 1, 1] + m[:, 2, 2] - 1) / 2
    one = torch.ones_like(cos)
    return torch.acos(torch.max(torch.min(cos, one), -one))


def rotation_metrics_panorama(pred_quat, gt_quat, out_dir=None):
    """geodesic rotation error (degrees) split by the ground-truth angle: overlap_large < pi/4 <= overlap_small < pi/2;
    mean / median / fraction within 10 deg per bucket (test_streetlearn_interiornet.py:71-122).  Quaternions xyzw.
    float64 torch-CPU arithmetic in the reference's operation order: bit-identical to its outputs on the same quaternions
    (tests/golden/reference_metrics.npz)."""
    import torch
    r_pred = torch.from_numpy(Rotation.from_quat(np.copy(pred_quat)).as_matrix()).view(-1, 3, 3)
    r_gt = torch.from_numpy(Rotation.from_quat(np.copy(gt_quat)).as_matrix()).view(-1, 3, 3)
    err = _angle_rad(torch.bmm(r_pred, r_gt.transpose(1, 2))) / np.pi * 180
    gt_angle = _angle_rad(r_gt)
    buckets = {"rotation_geodesic_error_overlap_large": err[gt_angle < (np.pi / 4)],
               "rotation_geodesic_error_overlap_small": err[(gt_angle >= np.pi / 4) & (gt_angle < np.pi / 2)]}
    if out_dir is not None:
        os.makedirs(out_dir, exist_ok=True)
        keep = gt_angle < (np.pi / 2)
        np.savetxt(os.path.join(out_dir, "all_rotation_err_degrees.csv"), err[keep].numpy().astype(np.float32), delimiter=",", fmt="%1.5f")
        np.savetxt(os.path.join(out_dir, "all_gt_rot_degrees.csv"), (gt_angle[keep] / np.pi * 180).numpy().astype(np.float32),
                   delimiter=",", fmt="%1.5f")
    out = {}
    for k, v in buckets.items():
        v = v.numpy()
        if v.size:
            out.update({k + "/mean": np.mean(v), k + "/median": np.median(v), k + "/10deg": np.true_divide((v <= 10).sum(axis=0), v.shape[0])})
    return out
