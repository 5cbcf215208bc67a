"""qkv input gradient + LayerNorm backward (gemm_dma_kernel<0,1,1,3>: 64 x 192 tiles, K = 576, three workgroups per CU = 768 slots): is
the 1152-tile launch of the 64-pair step paying for a half-empty second round?  time vs number of tiles"""
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


W = torch.randn(576, 192, device="cuda") * 0.05
gm = torch.randn(192, device="cuda")
for tiles in (384, 512, 768, 896, 1024, 1152, 1280, 1536, 2304):
    M = tiles * 64
    dy, x, add = torch.randn(M, 576, device="cuda"), torch.randn(M, 192, device="cuda"), torch.randn(M, 192, device="cuda")
    _, mean, rstd = ops.layernorm_fwd(x, gm, gm)
    t = timeit(lambda: ops.linear_dx_lnbwd(dy, W, x, gm, mean, rstd, add=add))
    t2 = timeit(lambda: ops.gemm(dy, W, M, 192, 576, b_layout=1))
    fl = 2.0 * M * 576 * 192
    print("tiles %5d (%.2f rounds of 768)  dx+lnbwd %7.1f us (%5.1f TF, %.3f us per tile)   plain gemm %7.1f us (%5.1f TF)" % (tiles, tiles / 768, t, fl / t * 1e-6, t / tiles, t2, fl / t2 * 1e-6), flush=True)
