"""Geodesic pose loss (reference src/geom/losses.py:3-21), over rel_pose_amd.se3.SE3.  Parity with lietorch's
arithmetic is unpinned (see se3.py)."""
import torch


def geodesic_loss(Ps, Gs, train_val="train"):
    ii, jj = torch.tensor([0, 1]), torch.tensor([1, 0])
    dP = Ps[:, jj] * Ps[:, ii].inv()
    dG = Gs[0][:, jj] * Gs[0][:, ii].inv()
    d = (dG * dP.inv()).log()
    tau, phi = d.split([3, 3], dim=-1)
    loss_tr = tau.norm(dim=-1).mean()
    loss_rot = phi.norm(dim=-1).mean()
    metrics = {train_val + "_geo_loss_tr": loss_tr.detach().item(),
               train_val + "_geo_loss_rot": loss_rot.detach().item()}
    return loss_tr, loss_rot, metrics


def geodesic_loss_tensors(Ps, Gs):
    """Same without the .item() host syncs (bench / graph-friendly)."""
    ii, jj = [0, 1], [1, 0]
    dP = Ps[:, jj] * Ps[:, ii].inv()
    dG = Gs[0][:, jj] * Gs[0][:, ii].inv()
    tau, phi = (dG * dP.inv()).log().split([3, 3], dim=-1)
    return tau.norm(dim=-1).mean(), phi.norm(dim=-1).mean()
