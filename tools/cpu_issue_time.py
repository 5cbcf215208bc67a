import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from rel_pose_amd.model import ViTEss
from rel_pose_amd.se3 import SE3
from rel_pose_amd.losses import geodesic_loss_tensors
dev = torch.device("cuda", 0)
model = ViTEss(bench.model_args()).to(dev).train()
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=5e-4)
B = 64
images, poses, intr = bench.synthetic_batch(B, 384, dev, 1)
fmap = torch.rand(2 * B, 192, 24, 24, device=dev)
Ps = SE3(poses); Gs = SE3.IdentityLike(Ps); i24 = (intr / 16).contiguous()
def step():
    opt.zero_grad(set_to_none=True)
    out = model.forward_tokens(fmap, Gs.data, i24)
    ltr, lrot = geodesic_loss_tensors(Ps, [SE3(out)])
    (10 * ltr + 10 * lrot).backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 2.5)
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
# CPU issue time with an empty GPU queue at the start of each step
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t0))
print("per step: cpu-issue %.1f ms, wall(with sync) %.1f ms" % (1e3 * sum(a for a, _ in ts) / 5, 1e3 * sum(b for _, b in ts) / 5))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); step(); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
