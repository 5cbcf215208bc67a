"""rp_conv3x3_c128_f32 against MIOpen at the step's shapes (128 images): layer2 128 -> 128 forward / input gradient, tail conv1 128 -> 192"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from rel_pose_amd import ops
N = int(os.environ.get("N", "128"))
CL = torch.channels_last


def timeit(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for CO in (128, 192):
    x = torch.randn(N, 128, 28, 28, device="cuda").contiguous(memory_format=CL)
    w = (torch.randn(CO, 128, 3, 3, device="cuda") * 0.03).contiguous(memory_format=CL)
    b = torch.randn(CO, device="cuda") if CO == 192 else None
    dy = torch.randn(N, CO, 28, 28, device="cuda").contiguous(memory_format=CL)
    xr, wr, dyr = x.permute(0, 2, 3, 1), w.permute(0, 2, 3, 1), dy.permute(0, 2, 3, 1)
    gf = 2.0 * N * 784 * 128 * CO * 9
    t_own = timeit(lambda: ops.conv3x3_c128_f32(xr, wr, b))
    t_mi = timeit(lambda: F.conv2d(x, w, b, 1, 1))
    print("CO=%d forward        own %6.1f us (%5.1f TF = %.2f)   MIOpen %6.1f us (%5.1f TF)" % (CO, t_own, gf / t_own * 1e-6, gf / t_own * 1e-6 / 157.3, t_mi, gf / t_mi * 1e-6), flush=True)
    if CO == 128:
        t_own = timeit(lambda: ops.conv3x3_c128_f32(dyr, wr, input_gradient=True))
        t_mi = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False]))
        print("CO=%d input gradient own %6.1f us (%5.1f TF = %.2f)   MIOpen %6.1f us (%5.1f TF)" % (CO, t_own, gf / t_own * 1e-6, gf / t_own * 1e-6 / 157.3, t_mi, gf / t_mi * 1e-6), flush=True)
