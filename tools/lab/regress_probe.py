"""the pose regressor's big Linear (26880 -> 512, 64 rows): forward, input gradient and weight gradient against their HBM floor (55 MB of weight)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rel_pose_amd import ops


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


B, K, H = 64, 26880, 512
x = torch.randn(B, K, device="cuda")
w0 = torch.randn(H, K, device="cuda") * 0.01
b0 = torch.randn(H, device="cuda")
dh = torch.randn(B, H, device="cuda")
ops.register_transposed(w0)
print("forward  linear(x, w0, relu)   %.1f us" % timeit(lambda: ops.linear(x, w0, b0, act=2)))
print("dx       linear_dx(dh, w0)     %.1f us" % timeit(lambda: ops.linear_dx(dh, w0)))
print("dw       linear_dw(dh, x)      %.1f us" % timeit(lambda: ops.linear_dw(dh, x)))
print("torch    x @ w0.t()            %.1f us" % timeit(lambda: x @ w0.t()))
print("torch    dh @ w0               %.1f us" % timeit(lambda: dh @ w0))
print("torch    dh.t() @ x            %.1f us" % timeit(lambda: dh.t() @ x))
print("floor: 55 MB at 5 TB/s = 11 us")
