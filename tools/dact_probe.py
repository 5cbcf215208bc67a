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
    return s.elapsed_time(e) / n * 1e6
do = torch.randn(M, 192, device="cuda"); W2 = torch.randn(192, 768, device="cuda") * 0.03
h = torch.randn(M, 768, device="cuda"); b = torch.zeros(768, device="cuda")
for tile in ("1,1", "2,1", "1,3", "2,3"):
    os.environ["RP_GEMM_TILE"] = tile
    print(tile, "NN raw   %.1f" % timeit(lambda: ops.gemm(do, W2, M, 768, 192, b_layout=1, split_k=1)),
          "dgelu %.1f" % timeit(lambda: ops.gemm(do, W2, M, 768, 192, b_layout=1, split_k=1, dact=1, aux=h)),
          "drelu %.1f" % timeit(lambda: ops.gemm(do, W2, M, 768, 192, b_layout=1, split_k=1, dact=2, aux=h)),
          "res %.1f" % timeit(lambda: ops.gemm(do, W2, M, 768, 192, b_layout=1, split_k=1, residual=h)),
          "bias %.1f" % timeit(lambda: ops.gemm(do, W2, M, 768, 192, b_layout=1, split_k=1, bias=b)),
          "bias+gelu+pre %.1f" % timeit(lambda: ops.gemm(do, W2, M, 768, 192, b_layout=1, split_k=1, bias=b, act=1, pre_out=h)), flush=True)
