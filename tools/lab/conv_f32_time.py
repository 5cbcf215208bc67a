#!/usr/bin/env python3
"""rp_conv3x3_c64_f32 against MIOpen at N images (default 128): forward and input gradient of the layer1 3x3 64 -> 64 convolution."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rel_pose_amd._env  # noqa
import torch
from rel_pose_amd import ops, _lib
_lib.load()
N = int(os.environ.get("Z", "128"))
CL = torch.channels_last
torch.manual_seed(0)
x = torch.randn(N, 64, 56, 56, device="cuda").contiguous(memory_format=CL)
w = (torch.randn(64, 64, 3, 3, device="cuda") * 0.04).contiguous(memory_format=CL)
dy = torch.randn(N, 64, 56, 56, device="cuda").contiguous(memory_format=CL)
xr, wr, dyr = x.permute(0, 2, 3, 1), w.permute(0, 2, 3, 1), dy.permute(0, 2, 3, 1)
wb = w.flip(2, 3).permute(1, 2, 3, 0).contiguous()


def timeit(fn, n=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

gf = 2.0 * N * 56 * 56 * 64 * 64 * 9
for rnd in range(2):
    t = timeit(lambda: ops.conv3x3_c64_f32(xr, wr))
    print("own forward          %8.1f us  %6.1f TF" % (t, gf / t * 1e-6))
    t = timeit(lambda: torch.nn.functional.conv2d(x, w, None, 1, 1))
    print("MIOpen forward       %8.1f us  %6.1f TF" % (t, gf / t * 1e-6))
    t = timeit(lambda: ops.conv3x3_c64_f32(dyr, wr, input_gradient=True))
    print("own input gradient   %8.1f us  %6.1f TF" % (t, gf / t * 1e-6))
    t = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False]))
    print("MIOpen bwd-data      %8.1f us  %6.1f TF" % (t, gf / t * 1e-6))
