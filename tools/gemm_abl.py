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
h = torch.randn(M, 768, device="cuda"); x = torch.randn(M, 192, device="cuda")
W2 = torch.randn(192, 768, device="cuda") * 0.05; W1 = torch.randn(768, 192, device="cuda") * 0.05
os.environ["RP_GEMM_WGS_PER_CU"] = "0"
for tile in ("2,1", "1,1", "2,2"):
    os.environ["RP_GEMM_TILE"] = tile
    for abl in (0, 1, 2, 3, 4, 7, 8, 15):
        os.environ["RP_GEMM_ABL"] = str(abl)
        t2 = timeit(lambda: ops.gemm(h, W2, M, 192, 768, split_k=1)); t1 = timeit(lambda: ops.gemm(x, W1, M, 768, 192, split_k=1))
        f = 2.0 * M * 768 * 192
        print("tile %s abl=%2d  fc2-shape(K=768) %.0f us %.0f TF | fc1-shape(K=192) %.0f us %.0f TF" % (tile, abl, t2, f / t2 / 1e6, t1, f / t1 / 1e6), flush=True)
