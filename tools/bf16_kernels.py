#!/usr/bin/env python3
"""Launch the kernels of the bf16 data path at the 128-pair shapes (BASELINE configs[4] per GPU) a few times -- one process for all
rocprofv3 --pmc passes (tools/pmc2.sh).  usage: bf16_kernels.py [what] [reps]     what: all | attn_fwd | attn_bwd | dw | rows"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rel_pose_amd import ops, _lib
_lib.load()
what = sys.argv[1] if len(sys.argv) > 1 else "all"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.manual_seed(0)
Z = 256
M = Z * 576
dev = "cuda"
bf = torch.bfloat16
qkv = (torch.randn(M, 576, device=dev)).to(bf)
do = torch.randn(M, 192, device=dev).to(bf)
for _ in range(reps):
    if what in ("all", "attn_fwd"):
        o, lse = ops.attn_fwd_bf16(qkv, Z)
        ops.attn_fwd_bf16(qkv, Z, stats_only=True)
    if what in ("all", "attn_bwd") and hasattr(ops, "attn_bwd_bf16"):
        o, lse = ops.attn_fwd_bf16(qkv, Z)
        ops.attn_bwd_bf16(qkv, o, lse, do, Z)
    if what in ("all", "dw"):
        ops.set_gemm_precision(1)
        dy768 = torch.randn(M, 768, device=dev).to(bf)
        x192 = torch.randn(M, 192, device=dev)
        ops.linear_dw(qkv, x192.to(bf))            # qkv: A [M,576] bf16, B bf16
        ops.linear_dw(dy768, x192)                 # fc1: A [M,768] bf16, B fp32
        ops.linear_dw(x192, do)                    # proj: A = o [M,192] bf16, B = dx1 fp32 (transposed reduce)
        ops.set_gemm_precision(0)
torch.cuda.synchronize()
print("ok")
