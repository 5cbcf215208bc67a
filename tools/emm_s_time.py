#!/usr/bin/env python3
"""Stored-S EMM (rp_emm_stats(s_out), rp_emm_apply / rp_emm_grad_ds(s_in)) against the recompute form at the 64-pair shape: isolated
sustained launch times of the statistics pass, the forward apply, and the whole backward."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rel_pose_amd import ops, _lib
_lib.load()
Z = 128
torch.manual_seed(0)
qkv = torch.randn(Z * 576, 576, device="cuda")


def timeit(fn, n=40, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

intr = torch.tensor([192.0, 192.0, 192.0, 192.0], device="cuda").repeat(Z // 2, 2, 1).contiguous()
pos = ops.posenc(intr, Z // 2, "cuda")
X = ops.emm_build_x(qkv, pos, Z)
df = torch.randn(Z, 3, 96, 96, device="cuda") * 0.01
for rnd in range(2):
    print("stats            %8.1f us   + S store %8.1f us" % (timeit(lambda: ops.emm_stats(qkv, Z)), timeit(lambda: ops.emm_stats(qkv, Z, want_s=True))))
    rlse, clse, sc = ops.emm_stats(qkv, Z, want_s=True)
    print("apply            %8.1f us   stored S  %8.1f us" % (timeit(lambda: ops.emm_apply(qkv, X, rlse, clse, Z)), timeit(lambda: ops.emm_apply(qkv, X, rlse, clse, Z, s=sc))))
    print("apply (swap)     %8.1f us   stored S  %8.1f us" % (timeit(lambda: ops.emm_apply(qkv, X, rlse, clse, Z, swap=True, want_f=False)),
                                                              timeit(lambda: ops.emm_apply(qkv, X, rlse, clse, Z, swap=True, want_f=False, s=sc))))
    tt, f = ops.emm_apply(qkv, X, rlse, clse, Z, s=sc)
    print("backward (all)   %8.1f us   stored S  %8.1f us" % (timeit(lambda: ops.emm_backward(qkv, X, tt, rlse, clse, df, Z), n=20),
                                                              timeit(lambda: ops.emm_backward(qkv, X, tt, rlse, clse, df, Z, s=sc), n=20)))
