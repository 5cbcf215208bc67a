#!/usr/bin/env python3
"""bf16-configuration fused MLP forward (training form) / backward, sustained timing at PAIRS pairs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rel_pose_amd import ops
ops.set_gemm_precision(1)
pairs = int(os.environ.get("PAIRS", "128"))
M = pairs * 2 * 576
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)
x, gm, bt = r(M, 192), 1 + 0.1 * r(192), 0.1 * r(192)
w1, b1, w2, b2 = r(768, 192) * 192 ** -0.5, 0.1 * r(768), r(192, 768) * 768 ** -0.5, 0.1 * r(192)
dy = r(M, 192)
def t(fn, n=int(os.environ.get('ITERS', '60'))):
    for _ in range(min(10, n)): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
fwd = lambda: ops.mlp_fused(x, gm, bt, w1, b1, w2, b2, train=True, out_dtype=torch.bfloat16, xn_dtype=torch.bfloat16)
hpre = fwd()[5]
bwd = lambda: ops.mlp_fused_bwd(dy, hpre, w1, w2, out_dtype=torch.bfloat16)
print("RP_MLP_DEBUG=%s  pairs %d: fwd %.1f us  bwd %.1f us" % (os.environ.get("RP_MLP_DEBUG", "0"), pairs, t(fwd), t(bwd)), flush=True)
