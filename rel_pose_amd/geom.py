"""Essential-matrix helpers (auxiliary, off the model's output path -- see include/relpose_hip.h and csrc/geom.hip).

The reference regresses R,t directly (src/model.py:91-98,145-159) and contains no SVD; BASELINE.json's north_star asks for
a one-wavefront Jacobi 3x3 SVD, so it exists here as a tool for consumers of the predicted pose (epipolar checks, the
(s, s, 0) structure of E = [t]x R), never between the regressor and ViTEss.forward's return value.
"""
import torch

from . import _lib
from .ops import _chk, _p, _st


def essential_from_pose(poses):
    """poses [...,7] = (t, q xyzw) -> E [...,3,3] = [t]x R(q)."""
    lib = _lib.load()
    flat = poses.reshape(-1, 7).contiguous()
    _chk(flat)
    E = torch.empty(flat.shape[0], 3, 3, device=flat.device, dtype=torch.float32)
    _lib.check(lib.rp_essential_from_pose(_p(flat), _p(E), flat.shape[0], _st()), "rp_essential_from_pose")
    return E.view(*poses.shape[:-1], 3, 3)


def svd3x3(A):
    """A [...,3,3] -> (U [...,3,3], S [...,3] descending, V [...,3,3]) with A = U diag(S) V^T."""
    lib = _lib.load()
    flat = A.reshape(-1, 9).contiguous()
    _chk(flat)
    n = flat.shape[0]
    U = torch.empty(n, 3, 3, device=flat.device, dtype=torch.float32)
    S = torch.empty(n, 3, device=flat.device, dtype=torch.float32)
    V = torch.empty(n, 3, 3, device=flat.device, dtype=torch.float32)
    _lib.check(lib.rp_svd3x3(_p(flat), _p(U), _p(S), _p(V), n, _st()), "rp_svd3x3")
    lead = A.shape[:-2]
    return U.view(*lead, 3, 3), S.view(*lead, 3), V.view(*lead, 3, 3)


def pose_from_essential(E, x1, x2):
    """E [n,3,3], correspondences x1, x2 [n,P,2] (normalised image coordinates of the same 3-D points in camera 1 / camera 2, with
    X2 = R X1 + t) -> (pose [n,7] = (t unit-norm, q xyzw with w >= 0), count [n] int32 = points in front of both cameras).
    The four-fold ambiguity of the SVD decode is resolved by the cheirality vote (csrc/geom.hip)."""
    lib = _lib.load()
    Ef = E.reshape(-1, 9).contiguous()
    n = Ef.shape[0]
    x1f, x2f = x1.reshape(n, -1, 2).contiguous(), x2.reshape(n, -1, 2).contiguous()
    _chk(Ef, x1f, x2f)
    if x1f.shape != x2f.shape:
        raise ValueError("x1 and x2 must have the same shape [n,P,2]")
    pose = torch.empty(n, 7, device=Ef.device, dtype=torch.float32)
    count = torch.empty(n, device=Ef.device, dtype=torch.int32)
    import ctypes
    _lib.check(lib.rp_pose_from_essential(_p(Ef), _p(x1f), _p(x2f), x1f.shape[1], _p(pose), ctypes.c_void_p(count.data_ptr()), n, _st()),
               "rp_pose_from_essential")
    return pose, count
