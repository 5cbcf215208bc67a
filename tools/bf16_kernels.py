#!/usr/bin/env python3
"""Launch the kernels of the bf16 configuration at the 128-pair shapes (BASELINE configs[4] per GPU) a few times -- one process for all
rocprofv3 --pmc passes (tools/pmc2.sh).  usage: bf16_kernels.py [what] [reps]
what: all (a Block and a CrossBlock, forward + backward, through the product autograd Functions) | attn_fwd | attn_bwd | dw"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rel_pose_amd import ops, _lib
_lib.load()
what = sys.argv[1] if len(sys.argv) > 1 else "all"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.manual_seed(0)
Z = int(os.environ.get("RP_Z", "256"))
M = Z * 576
dev = "cuda"
bf = torch.bfloat16
if what == "all":
    ops.set_gemm_precision(1)
    ops.set_attention_precision(1)
    x = torch.randn(Z, 576, 192, device=dev, requires_grad=True)
    def P(*s, sc=0.07):
        return (torch.randn(*s, device=dev) * sc).requires_grad_(True)
    n1w, n1b, n2w, n2b = (torch.ones(192, device=dev, requires_grad=True), torch.zeros(192, device=dev, requires_grad=True),
                          torch.ones(192, device=dev, requires_grad=True), torch.zeros(192, device=dev, requires_grad=True))
    qkv_w, qkv_b, proj_w, proj_b = P(576, 192), P(576), P(192, 192), P(192)
    fc1_w, fc1_b, fc2_w, fc2_b = P(768, 192), P(768), P(192, 768), P(192)
    pf_w, pf_b = P(192, 210), P(192)
    intr = torch.tensor([192.0, 192.0, 192.0, 192.0], device=dev).repeat(Z // 2, 2, 1).contiguous()
    pos = ops.posenc(intr, Z // 2, dev)
    for _ in range(reps):
        y = ops.BlockFn.apply(x, n1w, n1b, qkv_w, qkv_b, proj_w, proj_b, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b)
        y.backward(torch.randn_like(y))
        f = ops.CrossBlockFn.apply(x, pos, n1w, n1b, qkv_w, qkv_b, pf_w, pf_b, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b)
        f.backward(torch.randn_like(f))
else:
    qkv = (torch.randn(M, 576, device=dev)).to(bf)
    do = torch.randn(M, 192, device=dev).to(bf)
    for _ in range(reps):
        if what == "attn_fwd":
            o, lse = ops.attn_fwd_bf16(qkv, Z)
            ops.attn_fwd_bf16(qkv, Z, stats_only=True)
        if what == "attn_bwd":
            o, lse = ops.attn_fwd_bf16(qkv, Z)
            ops.attn_bwd_bf16(qkv, o, lse, do, Z)
        if what == "dw":
            ops.set_gemm_precision(1)
            dy768 = torch.randn(M, 768, device=dev).to(bf)
            x192 = torch.randn(M, 192, device=dev)
            ops.linear_dw(qkv, x192.to(bf))            # qkv: A [M,576] bf16, B bf16
            ops.linear_dw(dy768, x192)                 # fc1: A [M,768] bf16, B fp32
            ops.linear_dw(x192, do)                    # proj: A = o [M,192] bf16, B = dx1 fp32 (transposed reduce)
            ops.set_gemm_precision(0)
torch.cuda.synchronize()
print("ok")
