"""Ground-truth rotation of the panorama-crop datasets (InteriorNet / StreetLearn): each image is tagged with the pitch `x`
and yaw `y` of its virtual camera; the pair's relative rotation is R(view 2) R(view 1)^T
(reference src/data_readers/interiornet.py:25-51, test_streetlearn_interiornet.py:53-128)."""
import numpy as np
from scipy.spatial.transform import Rotation


def rotation_matrix_from_viewpoint(pitch, yaw):
    """pitch then (negated) yaw, float32 like the reference's torch.FloatTensor arithmetic -> [3,3] float32"""
    rx = np.float32(pitch)
    ry = -np.float32(yaw)
    c1, s1, c2, s2 = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry)
    return np.array([[c2, s1 * s2, c1 * s2],
                     [0.0, c1, -s1],
                     [-s2, s1 * c2, c1 * c2]], dtype=np.float32)


def relative_rotation(x1, y1, x2, y2):
    return rotation_matrix_from_viewpoint(x2, y2) @ rotation_matrix_from_viewpoint(x1, y1).T


def relative_quaternion(x1, y1, x2, y2):
    """xyzw quaternion of the pair's relative rotation (scipy convention, as the reference uses R.from_matrix(...).as_quat())"""
    return Rotation.from_matrix(relative_rotation(x1, y1, x2, y2).astype(np.float64)).as_quat()
