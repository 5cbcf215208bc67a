#!/usr/bin/env python3
"""Per-kernel timings at BASELINE sizes (B pairs): HIP events on torch's current stream."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rel_pose_amd import ops, _lib

_lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
Z = 2 * B
M = Z * 576
PEAK = 157.3e12


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
    return s.elapsed_time(e) / n * 1e-3


def line(name, t, flops):
    print("%-28s %9.1f us  %7.1f TFLOP/s  %5.1f%% of fp32-MFMA peak" % (name, t * 1e6, flops / t / 1e12, 100 * flops / t / PEAK), flush=True)


x = torch.randn(M, 192, device="cuda")
g, b = torch.ones(192, device="cuda"), torch.zeros(192, device="cuda")
Wqkv, Wp = torch.randn(576, 192, device="cuda") * 0.07, torch.randn(192, 192, device="cuda") * 0.07
W1, W2 = torch.randn(768, 192, device="cuda") * 0.07, torch.randn(192, 768, device="cuda") * 0.03
bq, b1 = torch.zeros(576, device="cuda"), torch.zeros(768, device="cuda")
qkv = ops.linear(x, Wqkv, bq)
h = ops.linear(x, W1, b1, act=1)
o, lse = ops.attn_fwd(qkv, Z)
line("layernorm_fwd", timeit(lambda: ops.layernorm_fwd(x, g, b)), 0.0 + 1)
line("gemm qkv  M x576x192", timeit(lambda: ops.linear(x, Wqkv, bq)), 2.0 * M * 576 * 192)
line("gemm proj M x192x192", timeit(lambda: ops.linear(x, Wp, b, residual=x)), 2.0 * M * 192 * 192)
line("gemm fc1  M x768x192 gelu", timeit(lambda: ops.linear(x, W1, b1, act=1)), 2.0 * M * 768 * 192)
line("gemm fc2  M x192x768", timeit(lambda: ops.linear(h, W2, b, residual=x)), 2.0 * M * 768 * 192)
line("attn_fwd", timeit(lambda: ops.attn_fwd(qkv, Z)), 4.0 * Z * 3 * 576 * 576 * 64)
line("attn_stats", timeit(lambda: ops.attn_fwd(qkv, Z, stats_only=True)), 2.0 * Z * 3 * 576 * 576 * 64)
do = torch.randn(M, 192, device="cuda")
line("attn_bwd (5 GEMM alg.)", timeit(lambda: ops.attn_bwd(qkv, o, lse, do, Z)), 10.0 * Z * 3 * 576 * 576 * 64)
line("gemm dX fc2 (dact)", timeit(lambda: ops.linear_dx(do, W2, dact=1, aux=h)), 2.0 * M * 768 * 192)
line("gemm dW fc1 splitK", timeit(lambda: ops.linear_dw(h, x)), 2.0 * M * 768 * 192)
line("colsum M x768", timeit(lambda: ops.colsum(h)), 1)
pos = ops.posenc(None, B, x.device)
rl, cl = ops.emm_stats(qkv, Z)
xa = ops.emm_build_x(qkv, pos, Z)
emm_alg = Z * 3 * (2.0 * 576 * 576 * 70)
line("emm_apply (T,F)", timeit(lambda: ops.emm_apply(qkv, xa, rl, cl, Z)), Z * 3 * 2.0 * 576 * 576 * 64 + emm_alg + Z * 3 * 2.0 * 576 * 70 * 70)
t, fp = ops.emm_apply(qkv, xa, rl, cl, Z)
dF = torch.randn(Z, 3, 96, 96, device="cuda")
line("emm_backward (all)", timeit(lambda: ops.emm_backward(qkv, xa, t, rl, cl, dF, Z), n=5), 1)
feats = torch.randn(B, 26880, device="cuda")
W0 = torch.randn(512, 26880, device="cuda") * 0.01
line("gemm regressor0 splitK", timeit(lambda: ops.linear(feats, W0, act=2)), 2.0 * B * 26880 * 512)
