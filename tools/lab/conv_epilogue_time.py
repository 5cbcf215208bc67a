"""what do the epilogue options of the two fp32 convolution kernels cost?  input gradient at 128 images: plain | + res | + bn mask | forward + stats"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rel_pose_amd import ops
N = int(os.environ.get("N", "128"))


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


for C, HW, conv in ((64, 56, ops.conv3x3_c64_f32), (128, 28, ops.conv3x3_c128_f32)):
    x = torch.randn(N, HW, HW, C, device="cuda")
    dy = torch.randn(N, HW, HW, C, device="cuda")
    w = torch.randn(C, 3, 3, C, device="cuda") * 0.03
    res = torch.randn(N, HW, HW, C, device="cuda")
    mean, rstd, gamma, beta = (torch.randn(C, device="cuda") * 0.1, torch.rand(C, device="cuda") + 0.5, torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1)
    rows = [("forward", lambda: conv(x, w)), ("forward + stats", lambda: conv(x, w, want_stats=True)),
            ("input gradient", lambda: conv(dy, w, input_gradient=True)), ("input gradient + res", lambda: conv(dy, w, input_gradient=True, res=res)),
            ("input gradient + bn mask", lambda: conv(dy, w, input_gradient=True, want_stats=True, bn=(x, mean, rstd, gamma, beta)))]
    for rep in range(2):
        print("C=%d  " % C + "   ".join("%s %.1f us" % (k, timeit(f)) for k, f in rows), flush=True)
