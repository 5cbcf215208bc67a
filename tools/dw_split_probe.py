"""weight-gradient GEMM (TN, split-K): time vs split factor and tile, per operand precision"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rel_pose_amd import ops
M = 128 * 576


def timeit(fn, n=10, warm=3):
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


for name, (N, K) in {"qkv": (576, 192), "fc1": (768, 192), "proj": (192, 192)}.items():
    x = torch.randn(M, K, device="cuda")
    dy = torch.randn(M, N, device="cuda")
    for prec in (3, 0):
        for tile in ("1,3", "2,1", "2,2"):
            os.environ["RP_GEMM_TILE"] = tile
            row = []
            for sk in (24, 32, 48, 64, 80, 96, 128):
                t = timeit(lambda: ops.gemm(dy, x, N, K, M, a_layout=1, b_layout=1, split_k=sk, precision=prec))
                row.append("%d:%.0f" % (sk, t))
            print("%-4s prec=%d tile=%s  " % (name, prec, tile) + "  ".join(row), flush=True)
