#!/usr/bin/env python3
"""Launch one hot kernel a few times (for rocprofv3 --pmc passes).  usage: one_kernel.py <what> [tile]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rel_pose_amd import ops, _lib
_lib.load()
what = sys.argv[1]
if len(sys.argv) > 2:
    os.environ["RP_GEMM_TILE"] = sys.argv[2]
Z = 128
M = Z * 576
x = torch.randn(M, 192, device="cuda")
W1 = torch.randn(768, 192, device="cuda") * 0.07
b1 = torch.zeros(768, device="cuda")
Wq = torch.randn(576, 192, device="cuda") * 0.07
bq = torch.zeros(576, device="cuda")
for _ in range(3):
    if what == "qkv":
        ops.linear(x, Wq, bq)
    elif what == "fc1":
        ops.linear(x, W1, b1, act=1)
    elif what == "dw":
        dY = torch.randn(M, 768, device="cuda")
        ops.linear_dw(dY, x)
    elif what == "dx":
        dY = torch.randn(M, 768, device="cuda")
        ops.linear_dx(dY, W1)
    elif what == "attn":
        qkv = ops.linear(x, Wq, bq)
        ops.attn_fwd(qkv, Z)
    elif what == "attn_bwd":
        qkv = ops.linear(x, Wq, bq)
        o, lse = ops.attn_fwd(qkv, Z)
        ops.attn_bwd(qkv, o, lse, torch.randn(M, 192, device="cuda"), Z)
torch.cuda.synchronize()
