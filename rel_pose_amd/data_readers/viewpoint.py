"""Ground-truth rotation of the panorama-crop datasets (InteriorNet / StreetLearn): each image is tagged with the pitch `x`
and yaw `y` of its virtual camera; the pair's relative rotation is R(view 2) R(view 1)^T
(reference src/data_readers/interiornet.py:25-51, test_streetlearn_interiornet.py:53-128).

The arithmetic is torch float32 on the CPU in the reference's order (sin / cos of the float32 angles, a float32 3x3 product), so
the quaternions are bit-identical to the ones the reference's readers and evaluation script build
(tests/golden/reference_metrics.npz: `pano_script_gt_rot`, `pano_gt_quat_for_viewpoints`)."""
import numpy as np
import torch
from scipy.spatial.transform import Rotation


def rotation_matrix_from_viewpoint(pitch, yaw):
    """pitch then (negated) yaw -> [3,3] float32 tensor"""
    rx = torch.tensor(float(pitch), dtype=torch.float64).to(torch.float32)
    ry = -torch.tensor(float(yaw), dtype=torch.float64).to(torch.float32)
    c1, s1, c2, s2 = torch.cos(rx), torch.sin(rx), torch.cos(ry), torch.sin(ry)
    zero = torch.zeros(())
    return torch.stack([torch.stack([c2, s1 * s2, c1 * s2]),
                        torch.stack([zero, c1, -s1]),
                        torch.stack([-s2, s1 * c2, c1 * c2])])


def relative_rotation(x1, y1, x2, y2):
    m1, m2 = rotation_matrix_from_viewpoint(x1, y1), rotation_matrix_from_viewpoint(x2, y2)
    return torch.bmm(m2[None], m1[None].transpose(1, 2))[0].numpy()


def relative_quaternion(x1, y1, x2, y2):
    """xyzw quaternion of the pair's relative rotation (scipy convention, as the reference uses R.from_matrix(...).as_quat())"""
    return Rotation.from_matrix(relative_rotation(x1, y1, x2, y2)).as_quat()
