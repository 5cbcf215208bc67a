#!/usr/bin/env python3
"""Run-to-run bit equality of the bf16-path kernels on fixed inputs (M = 128 pairs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rel_pose_amd import ops
ops.set_gemm_precision(1); ops.set_attention_precision(1)
M = int(os.environ.get("PAIRS", "128")) * 2 * 576
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)
x, gm, bt = r(M, 192), 1 + 0.1 * r(192), 0.1 * r(192)
W = r(576, 192) * 192 ** -0.5
dy = r(M, 576).to(torch.bfloat16)
add = r(M, 192)
_, mean, rstd = ops.layernorm_fwd(x, gm, bt)
def same(a, b):
    if isinstance(a, (tuple, list)):
        return all(same(u, v) for u, v in zip(a, b) if u is not None)
    return torch.equal(a, b)
def diffinfo(a, b, name):
    if isinstance(a, (tuple, list)):
        for i, (u, v) in enumerate(zip(a, b)):
            if u is not None: diffinfo(u, v, "%s[%d]" % (name, i))
        return
    d = (a.float() - b.float()).abs()
    n = int((d > 0).sum())
    if n:
        idx = (d > 0).nonzero()
        print("   %s: %d elements differ, max %.3e, first at %s, last at %s, shape %s" % (name, n, float(d.max()), idx[0].tolist(), idx[-1].tolist(), list(a.shape)))
ops.register_transposed(W) if hasattr(ops, "register_transposed") else None
tests = {
    "dx_lnbwd": lambda: ops.linear_dx_lnbwd(dy, W, x, gm, mean, rstd, add=add),
}
w1, b1, w2, b2 = r(768, 192) * 192 ** -0.5, 0.1 * r(768), r(192, 768) * 768 ** -0.5, 0.1 * r(192)
tests["mlp_fwd"] = lambda: ops.mlp_fused(x, gm, bt, w1, b1, w2, b2, train=True, out_dtype=torch.bfloat16, xn_dtype=torch.bfloat16)
hpre = tests["mlp_fwd"]()[5]
tests["mlp_bwd"] = lambda: ops.mlp_fused_bwd(add, hpre, w1, w2, out_dtype=torch.bfloat16)
for name, fn in tests.items():
    ref = fn()
    torch.cuda.synchronize()
    bad = badprev = 0
    prev = ref
    for i in range(20):
        out = fn()
        torch.cuda.synchronize()
        if not same(ref, out):
            bad += 1
            if bad == 1: diffinfo(ref, out, name)
        if not same(prev, out):
            badprev += 1
            if badprev <= 2: diffinfo(prev, out, name + " vs previous")
        prev = out
    print("%s: %d of 20 repeats differ from the first run, %d from their predecessor" % (name, bad, badprev), flush=True)
