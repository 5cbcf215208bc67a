#!/usr/bin/env python3
"""Row-resident K = 192 Linear (csrc/linear_rows.hip) against LayerNorm kernel + generic LDS-DMA GEMM, per ViT shape, sustained."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rel_pose_amd import ops


def timeit(fn, iters=int(os.environ.get('ITERS', '200'))):
    for _ in range(min(30, iters)):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    pairs = int(os.environ.get("PAIRS", "64"))
    M = pairs * 2 * 576
    g = torch.Generator(device="cuda").manual_seed(0)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)          # noqa: E731
    x, gm, bt, res = r(M, 192), 1 + 0.1 * r(192), 0.1 * r(192), r(M, 192)
    print("%d pairs, M = %d" % (pairs, M))
    only = os.environ.get("ONLY")
    for name, N, ln, act, pre, rs in (("qkv  (LN fused, train outputs)", 576, True, 0, False, False),
                                      ("qkv  (LN fused, inference)", 576, True, 0, False, False),
                                      ("proj (+residual)", 192, False, 0, False, True),
                                      ("fc1  (LN fused, GELU + pre, train)", 768, True, 1, True, False),
                                      ("fc1  (LN fused, GELU, inference)", 768, True, 1, False, False)):
        if only and not name.startswith(only):
            continue
        W, b = r(N, 192) * 192 ** -0.5, 0.1 * r(N)
        train = "train" in name
        new = lambda: ops.linear_rows(x, W, b, act=act, want_pre=pre, residual=res if rs else None,          # noqa: E731
                                      ln=(gm, bt) if ln else None, want_ln_out=train and ln)

        def old():
            xin = ops.layernorm_fwd(x, gm, bt, want_stats=train)[0] if ln else x
            pre_t = ops._empty(M, N, like=x) if pre else None
            return ops.gemm(xin, W, M, N, 192, bias=b, act=act, pre_out=pre_t, residual=res if rs else None)

        def old_gemm_only():
            pre_t = ops._empty(M, N, like=x) if pre else None
            return ops.gemm(x, W, M, N, 192, bias=b, act=act, pre_out=pre_t, residual=res if rs else None)

        tn, to, tg = timeit(new), timeit(old), timeit(old_gemm_only)
        fl = 2.0 * M * 192 * N
        print("%-38s N=%3d  rows kernel %7.1f us (%.1f TF, %.2f of 157.3) | LN + rp_gemm %7.1f us | rp_gemm alone %7.1f us (%.1f TF, %.2f)" %
              (name, N, tn, fl / tn / 1e6, fl / tn / 1e6 / 157.3, to, tg, fl / tg / 1e6, fl / tg / 1e6 / 157.3), flush=True)


if __name__ == "__main__":
    main()
