#!/usr/bin/env python3
"""Tile-shape sweep for the hot GEMM shapes (tuning aid; RP_GEMM_TILE override)."""
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
    return s.elapsed_time(e) / n * 1e-3

shapes = {"qkv": (M, 576, 192), "proj": (M, 192, 192), "fc1": (M, 768, 192), "fc2": (M, 192, 768)}
only = sys.argv[1:] or list(shapes)
for name, (m, n, k) in shapes.items():
    if name not in only: continue
    A = torch.randn(m, k, device="cuda"); W = torch.randn(n, k, device="cuda") * 0.05
    Wt = W.t().contiguous()
    for lay in ("NT", "NN", "TN"):
        for tile in ("1,1", "1,2", "1,3", "2,1", "2,2", "2,3"):
            os.environ["RP_GEMM_TILE"] = tile
            if lay == "NT":
                f = lambda: ops.gemm(A, W, m, n, k, split_k=1)
            elif lay == "NN":
                f = lambda: ops.gemm(A, Wt, m, n, k, b_layout=1, split_k=1)
            else:
                continue
            t = timeit(f)
            print("%-5s %s tile %s: %7.1f us %6.1f TF" % (name, lay, tile, t * 1e6, 2.0 * m * n * k / t / 1e12), flush=True)

# weight-gradient form (TN, split-K): dW[n,k] = dY[M,n]^T X[M,k]
for name, (n, k) in {"dW_qkv": (576, 192), "dW_fc1": (768, 192), "dW_fc2": (192, 768), "dW_proj": (192, 192)}.items():
    dY = torch.randn(M, n, device="cuda"); X = torch.randn(M, k, device="cuda")
    for tile in ("1,1", "1,2", "1,3", "2,1", "2,2", "2,3"):
        os.environ["RP_GEMM_TILE"] = tile
        for sk in (32, 64, 128, 256):
            t = timeit(lambda: ops.gemm(dY, X, n, k, M, a_layout=1, b_layout=1, split_k=sk))
            print("%-7s TN tile %s sk=%3d: %7.1f us %6.1f TF" % (name, tile, sk, t * 1e6, 2.0 * M * n * k / t / 1e12), flush=True)
