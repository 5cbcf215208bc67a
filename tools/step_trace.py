#!/usr/bin/env python3
"""Per-step GPU time of the first N training steps of a fresh process (events recorded after every step, read at the end): how long
does the bench configuration take to reach its steady state?  usage: tools/step_trace.py [steps] [batch]"""
import os, sys, types, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rel_pose_amd._env  # noqa: F401
import torch
from rel_pose_amd import parallel
from rel_pose_amd.losses import geodesic_loss_tensors
from rel_pose_amd.model import ViTEss
from rel_pose_amd.se3 import SE3
N = int(sys.argv[1]) if len(sys.argv) > 1 else 80
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
args = types.SimpleNamespace(fusion_transformer=True, transformer_depth=6, fc_hidden_size=512, cross_features=False,
                             use_single_softmax=False, no_pos_encoding=False, l1_pos_encoding=False, noess=False,
                             feature_resolution=(24, 24), num_heads=3, total_num_features=192, pool_size=60)
torch.manual_seed(0)
net = ViTEss(args).cuda().train()
opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=1e-4, fused=True)
images = torch.floor(torch.rand(B, 2, 3, 384, 384, device="cuda") * 255.0)
poses = torch.zeros(B, 2, 7, device="cuda"); poses[:, :, 6] = 1.0; poses[:, 1, :3] = 0.3
intr = torch.tensor([[192.0, 192.0, 192.0, 192.0]], device="cuda").repeat(B, 2, 1)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
host = []
torch.cuda.synchronize()
ev[0].record()
for it in range(N):
    t = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    Ps = SE3(poses); Gs = SE3.IdentityLike(Ps)
    est = net(images, Gs, intrinsics=intr.clone())
    ltr, lrot = geodesic_loss_tensors(Ps, est); loss = 10 * ltr + 10 * lrot
    loss.backward()
    parallel.clip_grad_norm_(net.parameters(), 2.5)
    opt.step()
    ev[it + 1].record()
    host.append((time.perf_counter() - t) * 1e3)
torch.cuda.synchronize()
gpu = [ev[i].elapsed_time(ev[i + 1]) for i in range(N)]
for i in range(0, N, 8):
    print("steps %3d..%3d  gpu ms: %s   host enqueue ms: %s" % (i, i + 7, " ".join("%6.1f" % v for v in gpu[i:i + 8]), " ".join("%6.1f" % v for v in host[i:i + 8])))
print("reserved MB", torch.cuda.memory_reserved() >> 20, " num alloc retries", torch.cuda.memory_stats().get("num_alloc_retries"))
