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


def rotation_from_quat(q):
    q = np.asarray(q, dtype=np.float64)
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    x, y, z, w = np.moveaxis(q, -1, 0)
    return np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                     np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                     np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], -2)


def decode_essential(E, x1, x2):
    """LAPACK restatement of the textbook E -> (R, t) decode with the cheirality vote (Hartley & Zisserman, Result 9.19):
    E [n,3,3], x1 / x2 [n,P,2] -> (R [n,3,3], t [n,3] unit, count [n]); float64."""
    E = np.asarray(E, dtype=np.float64)
    out_R, out_t, out_c = [], [], []
    W = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    for e, p1, p2 in zip(E, np.asarray(x1, dtype=np.float64), np.asarray(x2, dtype=np.float64)):
        U, _, Vt = np.linalg.svd(e)
        best = None
        for R in (U @ W @ Vt, U @ W.T @ Vt):
            if np.linalg.det(R) < 0:
                R = -R
            for t in (U[:, 2], -U[:, 2]):
                c = 0
                for a, b in zip(p1, p2):
                    m, x = R @ np.array([a[0], a[1], 1.0]), np.array([b[0], b[1], 1.0])
                    lam, *_ = np.linalg.lstsq(np.stack([m, -x], 1), -t, rcond=None)
                    c += int(lam[0] > 0 and lam[1] > 0)
                if best is None or c > best[2]:
                    best = (R, t, c)
        out_R.append(best[0]); out_t.append(best[1]); out_c.append(best[2])
    return np.stack(out_R), np.stack(out_t), np.array(out_c)
