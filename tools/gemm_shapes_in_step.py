#!/usr/bin/env python3
"""Every rp_gemm call of one training step (64 pairs), timed one by one (events + sync around each call: not a step time), against a
floor = max(flops / 130 TF, bytes / 4.5 TB/s).  Finds mis-tiled or mis-split shapes.  Tuning aid."""
import os, sys, types, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rel_pose_amd._env  # noqa
import torch
from rel_pose_amd import ops
from rel_pose_amd.losses import geodesic_loss_tensors
from rel_pose_amd.model import ViTEss
from rel_pose_amd.se3 import SE3
B = int(os.environ.get("BATCH", "64"))
args = types.SimpleNamespace(fusion_transformer=True, transformer_depth=6, fc_hidden_size=512, cross_features=False,
                             use_single_softmax=False, no_pos_encoding=False, l1_pos_encoding=False, noess=False,
                             feature_resolution=(24, 24), num_heads=3, total_num_features=192, pool_size=60)
torch.manual_seed(0)
net = ViTEss(args).cuda().train()
images = torch.floor(torch.rand(B, 2, 3, 384, 384, device="cuda") * 255.0)
poses = torch.zeros(B, 2, 7, device="cuda"); poses[:, :, 6] = 1.0; poses[:, 1, :3] = 0.3
intr = torch.tensor([[192.0, 192.0, 192.0, 192.0]], device="cuda").repeat(B, 2, 1)
def step():
    for p in net.parameters(): p.grad = None
    Ps = SE3(poses); Gs = SE3.IdentityLike(Ps)
    est = net(images, Gs, intrinsics=intr.clone())
    ltr, lrot = geodesic_loss_tensors(Ps, est); (10 * ltr + 10 * lrot).backward()
for _ in range(3): step()
torch.cuda.synchronize()
rec = collections.OrderedDict()
orig = ops.gemm
def timed_gemm(A, Bm, M, N, K, **kw):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); out = orig(A, Bm, M, N, K, **kw); e.record(); torch.cuda.synchronize()
    key = (M, N, K, kw.get("a_layout", 0), kw.get("b_layout", 0), kw.get("batch", 1), kw.get("split_k"), bool(kw.get("ln")), kw.get("dact", 0), kw.get("act", 0))
    r = rec.setdefault(key, [0, 0.0]); r[0] += 1; r[1] += s.elapsed_time(e) * 1e3
    return out
ops.gemm = timed_gemm
step()
ops.gemm = orig
print("%-52s %5s %9s %9s %6s" % ("M, N, K, aL, bL, batch, split, ln, dact, act", "calls", "us/call", "floor us", "ratio"))
for key, (n, t) in sorted(rec.items(), key=lambda kv: -kv[1][1]):
    M, N, K, al, bl, batch = key[:6]
    fl = 2.0 * M * N * K * batch; by = 4.0 * batch * (M * K + N * K + M * N)
    floor = max(fl / 130e12, by / 4.5e12) * 1e6
    print("%-52s %5d %9.1f %9.1f %6.2f" % (str(key), n, t / n, floor, t / n / floor))
