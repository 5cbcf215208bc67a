#!/usr/bin/env python3
"""Fused MLP backward-data kernel (rp_mlp_fused_bwd + LayerNorm backward) against the two GEMMs it replaces (fc2 input gradient with
GELU' epilogue + column sums; fc1 input gradient with the fused LayerNorm backward), sustained, 64 pairs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rel_pose_amd import ops


def timeit(fn, iters=100):
    for _ in range(20):
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
    dy, hpre, x1 = r(M, 192), r(M, 768), r(M, 192)
    w1, w2, gm = r(768, 192) * 192 ** -0.5, r(192, 768) * 768 ** -0.5, 1 + 0.1 * r(192)
    _, mean, rstd = ops.layernorm_fwd(x1, gm, 0.1 * r(192))

    def fused():
        dh, dxn, part = ops.mlp_fused_bwd(dy, hpre, w1, w2)
        return ops.layernorm_bwd(dxn, x1, gm, mean, rstd, add=dy), ops.colsum(part)

    def fused_kernel_only():
        return ops.mlp_fused_bwd(dy, hpre, w1, w2)

    def chain():
        dh, db1 = ops.linear_dx(dy, w2, dact=1, aux=hpre, want_colsum=True)
        return ops.linear_dx_lnbwd(dh, w1, x1, gm, mean, rstd, add=dy)

    tf, tk, tc = timeit(fused), timeit(fused_kernel_only), timeit(chain)
    fl = 4.0 * M * 192 * 768
    print("%d pairs: fused kernel + transposes %.1f us (%.1f TF); + LayerNorm backward + db1 column sums %.1f us | two GEMMs with fused "
          "epilogues %.1f us" % (pairs, tk, fl / tk / 1e6, tf, tc))


if __name__ == "__main__":
    main()
