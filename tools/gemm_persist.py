import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rel_pose_amd import ops, _lib
_lib.load()
M = 128 * 576
def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
x = torch.randn(M, 192, device="cuda"); h = torch.randn(M, 768, device="cuda")
W1 = torch.randn(768, 192, device="cuda") * 0.05; b1 = torch.zeros(768, device="cuda")
Wq = torch.randn(576, 192, device="cuda") * 0.05; bq = torch.zeros(576, device="cuda")
W2 = torch.randn(192, 768, device="cuda") * 0.05; b2 = torch.zeros(192, device="cuda")
Wp = torch.randn(192, 192, device="cuda") * 0.05
dY = torch.randn(M, 768, device="cuda")
for pc in sys.argv[1:]:
    os.environ["RP_GEMM_PIPE"] = pc
    r = [timeit(lambda: ops.linear(x, Wq, bq)), timeit(lambda: ops.linear(x, Wp, b2, residual=x)),
         timeit(lambda: ops.linear(x, W1, b1, act=1, want_pre=True)), timeit(lambda: ops.linear(h, W2, b2, residual=x)),
         timeit(lambda: ops.linear_dx(h, W1)), timeit(lambda: ops.linear_dx(x, W2, dact=1, aux=h)),
         timeit(lambda: ops.linear_dw(dY, x)), timeit(lambda: ops.linear_dw(x, h))]
    print("pipe=%s  qkv %.0f  proj %.0f  fc1(gelu+pre) %.0f  fc2 %.0f | dX(768->192) %.0f  dX(192->768,dgelu) %.0f | dW fc1 %.0f dW fc2 %.0f | sum %.0f us" % ((pc,) + tuple(r) + (sum(r),)), flush=True)
