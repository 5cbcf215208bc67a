#!/usr/bin/env python3
"""Launch every hot-path kernel at the 64-pair shapes a few times (one process for all rocprofv3 --pmc passes).
usage: all_kernels.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rel_pose_amd import ops, _lib
_lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
torch.manual_seed(0)
Z = 128
M = Z * 576
dev = "cuda"
x = torch.randn(M, 192, device=dev)
Wq = torch.randn(576, 192, device=dev) * 0.07; bq = torch.zeros(576, device=dev)
Wp = torch.randn(192, 192, device=dev) * 0.07; bp = torch.zeros(192, device=dev)
W1 = torch.randn(768, 192, device=dev) * 0.07; b1 = torch.zeros(768, device=dev)
W2 = torch.randn(192, 768, device=dev) * 0.07; b2 = torch.zeros(192, device=dev)
dY768 = torch.randn(M, 768, device=dev)
dY192 = torch.randn(M, 192, device=dev)
intr = torch.tensor([192.0, 192.0, 192.0, 192.0], device=dev).repeat(Z // 2, 2, 1).contiguous()
g1, be1 = torch.ones(192, device=dev), torch.zeros(192, device=dev)
for _ in range(reps):
    qkv = ops.ln_linear(x, g1, be1, Wq, bq)[0]        # linear_rows_kernel<true>: LayerNorm + qkv
    ops.linear(x, Wp, bp, residual=x)                 # linear_rows_kernel<false>: proj + residual
    h, hpre = ops.ln_linear(x, g1, be1, W1, b1, act=1, want_pre=True)[:2]
    ops.linear(h, W2, b2, residual=x)                 # gemm_dma<0,0,2,1>: fc2 forward
    ops.mlp_fused(x, g1, be1, W1, b1, W2, b2)         # inference MLP, one kernel
    ops.mlp_fused_bwd(dY192, hpre, W1, W2)            # MLP backward-data, one kernel
    ops.linear_dx(dY768, W1)                          # dX (K = 768): gemm_dma<0,1,1,3>
    ops.linear_dx(dY192, W2, dact=1, aux=hpre)        # dX with GELU' (row-resident form)
    ops.gemm(qkv, Wq, M, 192, 576, b_layout=1)        # dX of qkv: gemm_dma<0,1,...>
    ops.linear_dw(dY768, x)                           # dW split-K
    ops.linear_dw(dY192, h)
    ops.mlp_fused(x, g1, be1, W1, b1, W2, b2, train=True)      # training MLP forward (stores xn, h, h_pre)
    ops.attn_fwd(qkv, Z)                                       # inference attention
    o, lse, pst, mrun = ops.attn_fwd(qkv, Z, save_p=True)      # training attention: stored-P forward
    ops.attn_bwd(qkv, o, lse, dY192, Z, want_bias_partials=True, saved_p=(pst, mrun))      # dkdv_p + ds_matmul_t
    # EMM (stored-S form)
    pos = ops.posenc(intr, Z // 2, dev)
    X = ops.emm_build_x(qkv, pos, Z)
    rlse, clse, sc = ops.emm_stats(qkv, Z, want_s=True)
    t, f = ops.emm_apply(qkv, X, rlse, clse, Z, s=sc)
    df = torch.randn(Z, 3, 96, 96, device=dev) * 0.01
    ops.emm_backward(qkv, X, t, rlse, clse, df, Z, s=sc)
    y, mu, rs = ops.layernorm_fwd(x, torch.ones(192, device=dev), torch.zeros(192, device=dev))
    ops.layernorm_bwd(dY192, x, torch.ones(192, device=dev), mu, rs)
torch.cuda.synchronize()
print("ok")
