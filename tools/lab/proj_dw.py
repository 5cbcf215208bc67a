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
M = 73728
dy = torch.randn(M, 192, device="cuda"); x = torch.randn(M, 192, device="cuda")
for sk in (None, 64, 96, 128):
    print("proj dW [192,192] K=73728 split %s tile %s: %6.1f us" % (sk, os.environ.get("RP_GEMM_TILE"), timeit(lambda: ops.gemm(dy, x, 192, 192, M, a_layout=1, b_layout=1, split_k=sk))))
