#!/usr/bin/env python3
"""Weight-gradient GEMM shapes of a Block at 64 pairs (contraction over 73728 tokens) under tile / split-K overrides.  Tuning aid."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    from rel_pose_amd import ops, _lib
    _lib.load()
    M = 128 * 576
    def timeit(fn, n=30, warm=5):
        for _ in range(warm): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n * 1e3
    out = []
    for name, N, K in (("qkv", 576, 192), ("proj", 192, 192), ("fc1", 768, 192), ("fc2", 192, 768)):
        dy, x = torch.randn(M, N, device="cuda"), torch.randn(M, K, device="cuda")
        sk = os.environ.get("DW_SPLIT")
        if sk:
            f = (lambda: ops.gemm(x, dy, K, N, M, a_layout=1, b_layout=1, split_k=int(sk), trans_c=True, ldc=K, out=torch.empty(N, K, device="cuda"))) if (K > N) else (lambda: ops.gemm(dy, x, N, K, M, a_layout=1, b_layout=1, split_k=int(sk)))
        else:
            f = lambda: ops.linear_dw(dy, x)
        t = timeit(f)
        out.append("%s %6.1f us %5.1f TF" % (name, t, 2.0 * M * N * K / t * 1e-6))
    print("TILE=%s SPLIT=%s | " % (os.environ.get("RP_GEMM_TILE"), os.environ.get("DW_SPLIT")) + " | ".join(out))
else:
    for env in ({}, {"RP_GEMM_TILE": "2,3"}, {"RP_GEMM_TILE": "2,1"}, {"RP_GEMM_TILE": "1,2"}, {"RP_GEMM_TILE": "2,2"}, {"DW_SPLIT": "48"}, {"DW_SPLIT": "64"}, {"DW_SPLIT": "96"}, {"DW_SPLIT": "128"},
                {"RP_GEMM_TILE": "2,3", "DW_SPLIT": "48"}, {"RP_GEMM_TILE": "2,3", "DW_SPLIT": "96"}):
        subprocess.run([sys.executable, __file__, "x"], env=dict(os.environ, **env))
