#!/usr/bin/env python3
"""Per-shape time of the bf16-operand rp_gemm launches of a Block at 128 pairs (256 images): bytes moved, achieved GB/s, and the
effect of bf16 storage of the hidden tensors.  Tuning aid for the bf16 configuration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rel_pose_amd import ops, _lib
_lib.load()
bf = torch.bfloat16
M = 256 * 576
def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
x = torch.randn(M, 192, device="cuda"); Wq = torch.randn(576, 192, device="cuda") * .07; W1 = torch.randn(768, 192, device="cuda") * .07
W2 = torch.randn(192, 768, device="cuda") * .05; Wp = torch.randn(192, 192, device="cuda") * .07; b768 = torch.zeros(768, device="cuda"); b576 = torch.zeros(576, device="cuda"); b192 = torch.zeros(192, device="cuda")
h32 = torch.randn(M, 768, device="cuda"); h16 = h32.to(bf); dy192 = torch.randn(M, 192, device="cuda"); dqkv = torch.randn(M, 576, device="cuda")
pre32 = torch.empty(M, 768, device="cuda"); pre16 = torch.empty(M, 768, device="cuda", dtype=bf)
rows = [
 ("qkv fwd        fp32 out", lambda: ops.gemm(x, Wq, M, 576, 192, bias=b576, precision=1), 4 * M * (192 + 576)),
 ("proj fwd + res fp32 out", lambda: ops.gemm(x, Wp, M, 192, 192, bias=b192, residual=dy192, precision=1), 4 * M * (192 * 3)),
 ("fc1 fwd  h,hpre fp32   ", lambda: ops.gemm(x, W1, M, 768, 192, bias=b768, act=1, pre_out=pre32, precision=1), 4 * M * (192 + 2 * 768)),
 ("fc1 fwd  h,hpre bf16   ", lambda: ops.gemm(x, W1, M, 768, 192, bias=b768, act=1, pre_out=pre16, precision=1, out_dtype=bf), 4 * M * 192 + 2 * M * 2 * 768),
 ("fc2 fwd  A fp32        ", lambda: ops.gemm(h32, W2, M, 192, 768, bias=b192, residual=dy192, precision=1), 4 * M * (768 + 2 * 192)),
 ("fc2 fwd  A bf16        ", lambda: ops.gemm(h16, W2, M, 192, 768, bias=b192, residual=dy192, precision=1), 2 * M * 768 + 4 * M * 2 * 192),
 ("dX fc2 gelu' fp32      ", lambda: ops.gemm(dy192, W2, M, 768, 192, b_layout=1, dact=1, aux=h32, precision=1), 4 * M * (192 + 2 * 768)),
 ("dX fc2 gelu' bf16 io   ", lambda: ops.gemm(dy192, W2, M, 768, 192, b_layout=1, dact=1, aux=h16, precision=1, out_dtype=bf), 4 * M * 192 + 2 * M * 2 * 768),
 ("dX qkv (N=192) fp32    ", lambda: ops.gemm(dqkv, Wq, M, 192, 576, b_layout=1, precision=1), 4 * M * (576 + 192)),
 ("dW qkv                 ", lambda: ops.gemm(dqkv, x, 576, 192, M, a_layout=1, b_layout=1, precision=1), 4 * M * (576 + 192)),
 ("dW fc1  A fp32         ", lambda: ops.gemm(h32, x, 768, 192, M, a_layout=1, b_layout=1, precision=1), 4 * M * (768 + 192)),
 ("dW fc1  A bf16         ", lambda: ops.gemm(h16, x, 768, 192, M, a_layout=1, b_layout=1, precision=1), 2 * M * 768 + 4 * M * 192),
]
ops.set_gemm_precision(1)
g192, be192 = torch.ones(192, device="cuda"), torch.zeros(192, device="cuda")
W2t = W2.t().contiguous()
rows += [
 ("ROWS qkv fwd fp32 out  ", lambda: ops.linear_rows(x, Wq, b576), 4 * M * (192 + 576)),
 ("ROWS LN+qkv train      ", lambda: ops.linear_rows(x, Wq, b576, ln=(g192, be192), want_ln_out=True), 4 * M * (192 * 2 + 576)),
 ("ROWS proj + res        ", lambda: ops.linear_rows(x, Wp, b192, residual=dy192), 4 * M * 192 * 3),
 ("ROWS LN+fc1 h,hpre fp32", lambda: ops.linear_rows(x, W1, b768, act=1, want_pre=True, ln=(g192, be192), want_ln_out=True), 4 * M * (192 * 2 + 2 * 768)),
 ("ROWS LN+fc1 h,hpre bf16", lambda: ops.linear_rows(x, W1, b768, act=1, want_pre=True, ln=(g192, be192), want_ln_out=True, out_dtype=bf), 4 * M * 192 * 2 + 2 * M * 2 * 768),
 ("ROWS dX fc2 gelu' fp32 ", lambda: ops.linear_rows(dy192, W2t, dact_aux=h32, want_colsum=True), 4 * M * (192 + 2 * 768)),
 ("ROWS dX fc2 gelu' bf16 ", lambda: ops.linear_rows(dy192, W2t, dact_aux=h16, want_colsum=True, out_dtype=bf), 4 * M * 192 + 2 * M * 2 * 768),
]
for name, fn, nbytes in rows:
    t = timeit(fn)
    print("%s %8.1f us   %7.1f MB  %5.2f TB/s" % (name, t, nbytes / 1e6, nbytes / t * 1e-6))
