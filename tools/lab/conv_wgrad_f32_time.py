#!/usr/bin/env python3
"""lab: rp_conv3x3_c64_wgrad_f32 against MIOpen's fp32 backward-weights for resnet.layer1's 3x3 64 -> 64 convolution at Z images."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rel_pose_amd import ops
torch.backends.cudnn.benchmark = True
Z = int(os.environ.get("Z", "128"))
CL = torch.channels_last
x = torch.randn(Z, 64, 56, 56, device="cuda").contiguous(memory_format=CL)
dy = torch.randn(Z, 64, 56, 56, device="cuda").contiguous(memory_format=CL)
w = torch.randn(64, 64, 3, 3, device="cuda").contiguous(memory_format=CL)
xn, dn = x.permute(0, 2, 3, 1), dy.permute(0, 2, 3, 1)
def t(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
own = lambda: ops.conv3x3_c64_wgrad_f32(xn, dn)
mio = lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
a, b = own().permute(0, 3, 1, 2), mio()
err = float((a.double() - b.double()).abs().max() / b.double().abs().max())
gf = 2.0 * Z * 56 * 56 * 64 * 64 * 9 * 1e-9
to, tm = t(own), t(mio)
print("Z=%d  own %.1f us (%.1f TF incl. the reduce)   MIOpen %.1f us (%.1f TF)   max rel diff %.2e" % (Z, to, gf / to * 1e3, tm, gf / tm * 1e3, err))
