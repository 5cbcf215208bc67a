#!/usr/bin/env python3
"""Fused inference MLP (csrc/mlp_fused.hip) against the kernel chain it replaces (LayerNorm + fc1/GELU GEMM + fc2/residual GEMM),
per variant (RP_MLP_VARIANT: 0 = 8 waves x 2 per SIMD, 1 = 4 x 3, 2 = 12 x 3, 3 = 8 x 4 with spills), sustained timing."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    import torch
    from rel_pose_amd import ops
    pairs = int(os.environ.get("PAIRS", "64"))
    M = pairs * 2 * 576
    g = torch.Generator(device="cuda").manual_seed(0)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)          # noqa: E731
    x, gm, bt = r(M, 192), 1 + 0.1 * r(192), 0.1 * r(192)
    w1, b1, w2, b2 = r(768, 192) * 192 ** -0.5, 0.1 * r(768), r(192, 768) * 768 ** -0.5, 0.1 * r(192)

    def fused():
        return ops.mlp_fused(x, gm, bt, w1, b1, w2, b2)

    def chain():
        xn, _, _ = ops.layernorm_fwd(x, gm, bt, want_stats=False)
        return ops.linear(ops.linear(xn, w1, b1, act=1), w2, b2, residual=x)

    err = float((fused() - chain()).abs().max())
    out = []
    for name, fn in (("fused", fused), ("chain", chain)):
        iters = int(os.environ.get('ITERS', '200'))
        for _ in range(min(30, iters)):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        out.append("%s %.1f us (%.1f TF)" % (name, us, 4.0 * M * 192 * 768 / us / 1e6))
    print("variant %s, %d pairs (M = %d): %s | max |fused - chain| %.2e" % (os.environ.get("RP_MLP_VARIANT", "0"), pairs, M,
                                                                          "; ".join(out), err), flush=True)


if __name__ == "__main__":
    if os.environ.get("MLP_TIME_CHILD"):
        child()
    else:
        for pairs in ("64", "1", "8"):
            for v in ("0", "1", "2", "3"):
                subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, MLP_TIME_CHILD="1", RP_MLP_VARIANT=v, PAIRS=pairs))
