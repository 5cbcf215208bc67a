#!/usr/bin/env python3
"""lab: rp_conv_stem_wgrad_f32 against MIOpen's fp32 backward-weights for resnet.conv1 (7x7 / 2, 3 -> 64) at Z images of 224 x 224."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from rel_pose_amd import ops
torch.backends.cudnn.benchmark = True
Z = int(os.environ.get("Z", "128"))
CL = torch.channels_last
x = torch.randn(Z, 3, 224, 224, device="cuda").contiguous(memory_format=CL)
xp = F.pad(x.permute(0, 2, 3, 1), (0, 0, 3, 3, 3, 3)).contiguous()
dy = torch.randn(Z, 64, 112, 112, device="cuda").contiguous(memory_format=CL)
w = torch.randn(64, 3, 7, 7, device="cuda").contiguous(memory_format=CL)
dn = dy.permute(0, 2, 3, 1)
def t(fn, n=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
own = lambda: ops.conv_stem_wgrad_f32(xp, dn)
mio = lambda: torch.ops.aten.convolution_backward(dy, xp.permute(0, 3, 1, 2), w, None, [2, 2], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1]
a, b = own().permute(0, 3, 1, 2), mio()
err = float((a.double() - b.double()).abs().max() / b.double().abs().max())
gf = 2.0 * Z * 112 * 112 * 64 * 147 * 1e-9
to, tm = t(own), t(mio)
print("Z=%d  own %.1f us (%.1f TF algorithmic, incl. space-to-depth + reduce)   MIOpen %.1f us (%.1f TF)   max rel diff %.2e" % (Z, to, gf / to * 1e3, tm, gf / tm * 1e3, err))
