"""CPU oracle for the essential-matrix auxiliary (TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product).

There is no reference code for this row (the reference regresses R,t and holds no SVD), so the pin is
LAPACK: numpy.linalg.svd in float64 on the same float32 inputs, and E = [t]x R(q) from the textbook quaternion formula.
"""
import numpy as np


def essential_from_pose(pose):
    """pose [n,7] (t, q xyzw) -> E [n,3,3] = [t]x R(q), float64."""
    pose = np.asarray(pose, dtype=np.float64)
    t, q = pose[:, :3], pose[:, 3:]
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    x, y, z, w = q.T
    R = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                  np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                  np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], 1)
    tx = np.zeros((len(t), 3, 3))
    tx[:, 0, 1], tx[:, 0, 2] = -t[:, 2], t[:, 1]
    tx[:, 1, 0], tx[:, 1, 2] = t[:, 2], -t[:, 0]
    tx[:, 2, 0], tx[:, 2, 1] = -t[:, 1], t[:, 0]
    return tx @ R


def singular_values(A):
    return np.linalg.svd(np.asarray(A, dtype=np.float64), compute_uv=False)
