"""Minimal SE3 type standing in for lietorch.SE3 on the model path and in the loss.

The reference leans on the un-vendored third-party extension lietorch (pinned lietorch==0.2, reference
environment.yml:19).  Call sites: src/model.py:9,146-152,164; train.py:144-146; src/geom/losses.py:8-10.
On the model path SE3 is only a wrapper around a [...,7] tensor (t(3), q xyzw(4)); real group arithmetic
(mul / inv / log) happens only in the geodesic loss.  This file restates that arithmetic from the published
SE(3) formulas in plain PyTorch (autograd supplies Euclidean gradients of the 7-vector; lietorch's custom
backward uses a tangent-space convention) => loss VALUE follows the textbook definition, its parity with
lietorch is UNPINNED (no lietorch here, no test in the reference) -- SURVEY.md 8c, 8f-2; DESIGN.md.
"""
import torch


def _qmul(a, b):
    ax, ay, az, aw = a.unbind(-1)
    bx, by, bz, bw = b.unbind(-1)
    return torch.stack([aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx,
                        aw * bz + ax * by - ay * bx + az * bw,
                        aw * bw - ax * bx - ay * by - az * bz], dim=-1)


def _qconj(q):
    return torch.cat([-q[..., :3], q[..., 3:]], dim=-1)


def _qrot(q, v):
    """rotate v by unit quaternion q (xyzw)."""
    u, w = q[..., :3], q[..., 3:]
    t = 2.0 * torch.cross(u, v, dim=-1)
    return v + w * t + torch.cross(u, t, dim=-1)


def _so3_log(q):
    """unit quaternion -> rotation vector phi (angle * axis), numerically safe near 0."""
    u, w = q[..., :3], q[..., 3]
    n = u.norm(dim=-1)
    sign = torch.where(w < 0, -torch.ones_like(w), torch.ones_like(w))     # shortest arc
    w, n_s = w * sign, n
    small = n_s < 1e-6
    n_safe = torch.where(small, torch.ones_like(n_s), n_s)
    fac = torch.where(small, 2.0 / w.clamp_min(1e-12) - (2.0 / 3.0) * n_s * n_s / w.clamp_min(1e-12) ** 3,
                      2.0 * torch.atan2(n_safe, w) / n_safe)
    return u * (sign * fac).unsqueeze(-1)


def _hat(v):
    x, y, z = v.unbind(-1)
    o = torch.zeros_like(x)
    return torch.stack([torch.stack([o, -z, y], -1), torch.stack([z, o, -x], -1), torch.stack([-y, x, o], -1)], -2)


class SE3:
    """data[..., :3] = translation, data[..., 3:] = unit quaternion (x, y, z, w)."""

    def __init__(self, data):
        self.data = data

    @staticmethod
    def IdentityLike(G):
        d = torch.zeros_like(G.data)
        d[..., 6] = 1.0
        return SE3(d)

    def __getitem__(self, idx):
        return SE3(self.data[idx])

    @property
    def shape(self):
        return self.data.shape[:-1]

    def detach(self):
        return SE3(self.data.detach())

    def __mul__(self, other):
        t1, q1 = self.data[..., :3], self.data[..., 3:]
        t2, q2 = other.data[..., :3], other.data[..., 3:]
        return SE3(torch.cat([t1 + _qrot(q1, t2), _qmul(q1, q2)], dim=-1))

    def inv(self):
        t, q = self.data[..., :3], self.data[..., 3:]
        qi = _qconj(q)
        return SE3(torch.cat([-_qrot(qi, t), qi], dim=-1))

    def log(self):
        """[..., 6] = (tau, phi) with t = V(phi) tau."""
        t, q = self.data[..., :3], self.data[..., 3:]
        phi = _so3_log(q)
        th = phi.norm(dim=-1)
        K = _hat(phi)
        th2 = th * th
        small = th < 1e-4
        ths = torch.where(small, torch.ones_like(th), th)
        c = torch.where(small, 1.0 / 12.0 + th2 / 720.0,
                        (1.0 - ths * torch.cos(ths / 2) / (2.0 * torch.sin(ths / 2))) / (ths * ths))
        eye = torch.eye(3, dtype=t.dtype, device=t.device).expand(K.shape)
        Vinv = eye - 0.5 * K + c[..., None, None] * (K @ K)
        tau = (Vinv @ t.unsqueeze(-1)).squeeze(-1)
        return torch.cat([tau, phi], dim=-1)


def tangent_gradient(data, grad):
    """Euclidean gradient of a scalar w.r.t. the 7-vector (t, q xyzw) -> the tangent-space gradient in lietorch's storage
    convention: [dL/dtau (3), dL/dphi (3), 0] with X <- Exp((tau, phi)) * X (left perturbation), evaluated at 0:
        t' = t + tau + phi x t,   q' = q + 1/2 (phi, 0) (x) q      (first order)
    =>  dL/dtau = g_t,   dL/dphi = t x g_t + 1/2 M(q)^T g_q.
    lietorch (the reference's SE3, un-vendored) returns gradients of group elements in this embedded-tangent form
    [recalled, not verifiable here: SURVEY.md 8c]; this function is the exact chain rule for THAT definition and is
    checked against finite differences of Exp(xi) * X in tests/test_host_cpu.py.  Used only when
    rel_pose_amd.losses.GRADIENT_CONVENTION == "tangent" (default: "euclidean", plain autograd of the 7-vector)."""
    t, q = data[..., :3], data[..., 3:]
    gt, gq = grad[..., :3], grad[..., 3:]
    bx, by, bz, bw = q.unbind(-1)
    gx, gy, gz, gw = gq.unbind(-1)
    gphi_q = 0.5 * torch.stack([bw * gx - bz * gy + by * gz - bx * gw,
                                bz * gx + bw * gy - bx * gz - by * gw,
                                -by * gx + bx * gy + bw * gz - bz * gw], dim=-1)
    gphi = torch.cross(t, gt, dim=-1) + gphi_q
    return torch.cat([gt, gphi, torch.zeros_like(gw).unsqueeze(-1)], dim=-1)


class _TangentGrad(torch.autograd.Function):
    """identity in the forward; converts the incoming Euclidean gradient to the embedded-tangent form in the backward"""

    @staticmethod
    def forward(ctx, data):
        ctx.save_for_backward(data)
        return data.view_as(data)

    @staticmethod
    def backward(ctx, grad):
        (data,) = ctx.saved_tensors
        return tangent_gradient(data, grad)


def with_tangent_gradient(G):
    """SE3 whose data back-propagates tangent-space gradients (see tangent_gradient)"""
    return SE3(_TangentGrad.apply(G.data))
