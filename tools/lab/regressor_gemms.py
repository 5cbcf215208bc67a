import os, sys, torch
sys.path.insert(0, os.getcwd())
from rel_pose_amd import ops, _lib
_lib.load()
def timeit(fn, n=30, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
B, F_, Hd = 64, 26880, 512
x = torch.randn(B, F_, device="cuda"); W0 = torch.randn(Hd, F_, device="cuda") * 0.01; b0 = torch.zeros(Hd, device="cuda")
dh = torch.randn(B, Hd, device="cuda")
print("W0 = %.1f MB; streaming it once at 5 TB/s = %.1f us" % (W0.numel() * 4 / 1e6, W0.numel() * 4 / 5e6))
print("fwd   y = relu(x W0^T + b)  [64,26880]x[26880,512]: %6.1f us" % timeit(lambda: ops.linear(x, W0, b0, act=2)))
print("dX    dx = dh W0            [64,512]x[512,26880]  : %6.1f us" % timeit(lambda: ops.linear_dx(dh, W0)))
print("dW    dW0 = dh^T x          [512,64]x[64,26880]   : %6.1f us" % timeit(lambda: ops.linear_dw(dh, x)))
# CrossBlock-sized Linears: M = 128 images * 70 rows
M = 128 * 70
g = torch.randn(M, 224, device="cuda"); Wp = torch.randn(192, 224, device="cuda") * 0.05; bp = torch.zeros(192, device="cuda"); df = torch.randn(M, 192, device="cuda")
print("pf fwd [8960,224]x[224,192]: %6.1f us   dX: %6.1f us   dW: %6.1f us" % (timeit(lambda: ops.linear(g, Wp, bp)), timeit(lambda: ops.linear_dx(df, Wp)), timeit(lambda: ops.linear_dw(df, g))))
for sk in (1, 2, 4, 8):
    t = timeit(lambda: ops.gemm(dh, W0, B, F_, Hd, b_layout=1, split_k=sk))
    print("dX with split_k=%d: %6.1f us" % (sk, t))
W0t = W0.t().contiguous()
print("dX on a transposed copy (K-contiguous B [26880,512]): %6.1f us" % timeit(lambda: ops.gemm(dh, W0t, B, F_, Hd)))
for sk in (1, 2, 4, 8, 16):
    print("fwd with split_k=%d: %6.1f us" % (sk, timeit(lambda: ops.gemm(x, W0, B, Hd, F_, bias=None, split_k=sk))))
