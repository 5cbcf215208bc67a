"""Geodesic pose loss (reference src/geom/losses.py:3-21), over rel_pose_amd.se3.SE3.  Parity with lietorch's
arithmetic is unpinned (see se3.py).

The reference indexes with ii=[0,1], jj=[1,0] (tensor indices -> host-to-device copies every call); here the same
selection is ``flip(1)``, which keeps the step free of host syncs and capturable in a HIP graph."""
import os

import torch

from .se3 import SE3, with_tangent_gradient

# What flows back from the loss into the predicted poses [B,2,7]:
#   "euclidean" (default): the plain derivative w.r.t. the seven stored numbers (t, q);
#   "tangent": the embedded tangent-space gradient [dL/dtau, dL/dphi, 0] of a left perturbation Exp(xi) * G, which is what
#              lietorch's SE3 backward is recalled to return -- the definition is verified against finite
#              differences (tests/test_host_cpu.py), lietorch's use of it is NOT verifiable offline: parity stays unpinned.
GRADIENT_CONVENTION = os.environ.get("RP_LIE_GRADIENT", "euclidean")


def _pair_swap(G):
    """G[:, [1, 0]]"""
    return SE3(G.data.flip(1))


def geodesic_loss_tensors(Ps, Gs):
    """(translation, rotation) geodesic losses as device scalars -- no .item(), no host sync.  fp32 GPU poses take the fused
    kernel (ops.GeodesicLossFn: same formulas, one launch, exact derivatives); anything else the PyTorch formulation below."""
    if GRADIENT_CONVENTION == "tangent":
        Gs = [with_tangent_gradient(Gs[0])]
    elif GRADIENT_CONVENTION != "euclidean":
        raise ValueError("RP_LIE_GRADIENT must be 'euclidean' or 'tangent'")
    g = Gs[0].data
    if g.is_cuda and g.dtype == torch.float32 and g.dim() == 3 and Ps.data.dtype == torch.float32:
        from . import ops
        return ops.GeodesicLossFn.apply(Ps.data, g)
    return geodesic_loss_tensors_torch(Ps, Gs)


def geodesic_loss_tensors_torch(Ps, Gs):
    """reference formulation on rel_pose_amd.se3.SE3 (any device / dtype; autograd supplies the gradient)"""
    dP = _pair_swap(Ps) * Ps.inv()
    dG = _pair_swap(Gs[0]) * Gs[0].inv()
    tau, phi = (dG * dP.inv()).log().split([3, 3], dim=-1)
    return tau.norm(dim=-1).mean(), phi.norm(dim=-1).mean()


class LazyMetrics(dict):
    """The reference's metrics dict ({name: python float}, src/geom/losses.py:16-19) whose floats are only fetched from the device
    when somebody LOOKS at them: the reference's two `.item()` calls per step are two host syncs per step, which at its own batch
    of 6 pairs per GPU serialise the host enqueue time with the kernels.  train.py prints the metrics every 20
    steps; the other 19 never synchronise.  Behaves as a plain dict for every read access."""

    def __init__(self, tensors):
        super().__init__(tensors)
        self._pending = True

    def _fetch(self):
        if self._pending:
            self._pending = False
            for k in list(dict.keys(self)):
                v = dict.__getitem__(self, k)
                dict.__setitem__(self, k, v.item() if hasattr(v, "item") else v)

    def __getitem__(self, k):
        self._fetch()
        return dict.__getitem__(self, k)

    def get(self, k, default=None):
        self._fetch()
        return dict.get(self, k, default)

    # CPython's dict fast paths (`dict(m)`, `{}.update(m)`, `{**m}`, `m | other`) read a dict SUBCLASS's storage directly
    # unless `__iter__` is overridden (dictobject.c: dict_merge tests `tp_iter == dict_iter`); the reference loop does
    # `metrics.update(geo_metrics)` (train.py:168) before handing the dict to its Logger, so the override is what keeps device
    # tensors from leaking into a plain dict.
    def __iter__(self):
        self._fetch()
        return dict.__iter__(self)

    def keys(self):
        self._fetch()
        return dict.keys(self)

    def items(self):
        self._fetch()
        return dict.items(self)

    def values(self):
        self._fetch()
        return dict.values(self)

    def copy(self):
        self._fetch()
        return dict(self)

    def __repr__(self):
        self._fetch()
        return dict.__repr__(self)

    __str__ = __repr__

    def __reduce__(self):
        self._fetch()
        return (dict, (dict(dict.items(self)),))


def geodesic_loss(Ps, Gs, train_val="train"):
    loss_tr, loss_rot = geodesic_loss_tensors(Ps, Gs)
    metrics = LazyMetrics({train_val + "_geo_loss_tr": loss_tr.detach(), train_val + "_geo_loss_rot": loss_rot.detach()})
    return loss_tr, loss_rot, metrics
