#!/usr/bin/env python3
"""Hand-written stem convolution (csrc/conv_stem.hip) against MIOpen's, 128 images of 224x224 (64 pairs), forward."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from rel_pose_amd import ops, _env  # noqa: F401
from tools.rows_time import timeit

N = int(os.environ.get("IMAGES", "128"))
torch.manual_seed(0)
x = torch.randn(N, 3, 224, 224, device="cuda").contiguous(memory_format=torch.channels_last)
w = (torch.randn(64, 3, 7, 7, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
xp = F.pad(x.permute(0, 2, 3, 1), (0, 0, 3, 3, 3, 3)).contiguous()
y = ops.conv_stem_fwd(xp, w)
ref = F.conv2d(x.double(), w.double(), None, 2, 3).permute(0, 2, 3, 1)
ref32 = F.conv2d(x, w, None, 2, 3).permute(0, 2, 3, 1)
print("max |hand - fp64| / max %.2e ; MIOpen fp32 vs fp64 %.2e" % (float((y - ref).abs().max() / ref.abs().max()),
                                                                   float((ref32 - ref).abs().max() / ref.abs().max())))
fl = 2.0 * N * 112 * 112 * 64 * 147
th, tm = timeit(lambda: ops.conv_stem_fwd(xp, w)), timeit(lambda: F.conv2d(x, w, None, 2, 3))
ts = timeit(lambda: ops.conv_stem_fwd(xp, w, want_stats=True))
y2, st = ops.conv_stem_fwd(xp, w, want_stats=True)
print('with statistics partials: %.1f us; sum(y) rel err %.2e, sum(y^2) rel err %.2e' % (ts, float(((st[:, 0].sum(0) - ref.sum((0, 1, 2))).abs() / ref.sum((0, 1, 2)).abs().max()).max()), float(((st[:, 1].sum(0) - (ref * ref).sum((0, 1, 2))).abs() / (ref * ref).sum((0, 1, 2))).max())))
tp = timeit(lambda: F.pad(x.permute(0, 2, 3, 1), (0, 0, 3, 3, 3, 3)))
print("stem conv forward, %d images: hand-written %.1f us (%.1f TF algorithmic) | MIOpen %.1f us (%.1f TF) | zero-frame copy (if not fused into "
      "the preprocessing kernel) %.1f us" % (N, th, fl / th / 1e6, tm, fl / tm / 1e6, tp))
