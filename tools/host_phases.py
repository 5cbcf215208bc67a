#!/usr/bin/env python3
"""Host-side (CPU) time of each phase of a launch-bound training step (--batch 6): time.perf_counter around the calls, no device sync
inside the step, so a phase's figure is what the Python / ATen / autograd host code costs to ENQUEUE it."""
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rel_pose_amd._env  # noqa: F401
import torch
from rel_pose_amd import parallel
from rel_pose_amd.losses import geodesic_loss_tensors
from rel_pose_amd.model import ViTEss
from rel_pose_amd.se3 import SE3

B = int(os.environ.get("BATCH", "6"))
args = types.SimpleNamespace(fusion_transformer=True, transformer_depth=6, fc_hidden_size=512, cross_features=False,
                             use_single_softmax=False, no_pos_encoding=False, l1_pos_encoding=False, noess=False,
                             feature_resolution=(24, 24), num_heads=3, total_num_features=192, pool_size=60)
torch.manual_seed(0)
net = ViTEss(args).cuda().train()
opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=1e-4, fused=True)
images = torch.floor(torch.rand(B, 2, 3, 384, 384, device="cuda") * 255.0)
poses = torch.zeros(B, 2, 7, device="cuda"); poses[:, :, 6] = 1.0; poses[:, 1, :3] = 0.3
intr = torch.tensor([[192.0, 192.0, 192.0, 192.0]], device="cuda").repeat(B, 2, 1)
acc = dict(forward=0.0, loss=0.0, backward=0.0, clip=0.0, adam=0.0, zero_grad=0.0)
N = 60
for it in range(N + 10):
    t = [time.perf_counter()]
    opt.zero_grad(set_to_none=True); t.append(time.perf_counter())
    Ps = SE3(poses); Gs = SE3.IdentityLike(Ps)
    est = net(images, Gs, intrinsics=intr.clone()); t.append(time.perf_counter())
    ltr, lrot = geodesic_loss_tensors(Ps, est); loss = 10 * ltr + 10 * lrot; t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    parallel.clip_grad_norm_(net.parameters(), 2.5); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    if it >= 10:
        for k, (a, b) in zip(("zero_grad", "forward", "loss", "backward", "clip", "adam"), zip(t[:-1], t[1:])):
            acc[k] += b - a
    if it == 9:
        torch.cuda.synchronize(); t0 = time.perf_counter()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / N
print("batch %d: wall %.2f ms per step; host enqueue time per phase (ms): %s ; sum %.2f" %
      (B, wall * 1e3, ", ".join("%s %.2f" % (k, v / N * 1e3) for k, v in acc.items()), sum(acc.values()) / N * 1e3))
