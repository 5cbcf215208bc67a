#!/usr/bin/env python3
"""Stored-P attention (rp_attn_fwd_savep / rp_attn_bwd_dkdv_p) against the recompute form at the 64-pair shape: isolated, sustained
launch times of the forward, the dK/dV pass and the whole backward (delta + dK/dV + rp_ds_matmul)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rel_pose_amd import ops, _lib
lib = _lib.load()
Z = int(os.environ.get("Z", "128"))
torch.manual_seed(0)
qkv = torch.randn(Z * 576, 576, device="cuda")
do = torch.randn(Z * 576, 192, device="cuda")


def timeit(fn, n=40, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

g = 2.0 * 576 * 576 * 64 * 3 * Z       # one 576 x 576 x 64 product over all heads and images
for rnd in range(int(os.environ.get("ROUNDS", "2"))):
    t = timeit(lambda: ops.attn_fwd(qkv, Z))
    print("attn_fwd              %8.1f us  %6.1f TF" % (t, 2 * g / t * 1e-6))
    t = timeit(lambda: ops.attn_fwd(qkv, Z, save_p=True))
    print("attn_fwd_savep        %8.1f us  %6.1f TF" % (t, 2 * g / t * 1e-6))
    o, lse, pst, mrun = ops.attn_fwd(qkv, Z, save_p=True)
    t0 = timeit(lambda: ops.attn_bwd(qkv, o, lse, do, Z, want_bias_partials=True), n=30)
    print("attn_bwd recompute    %8.1f us  %6.1f TF algorithmic (4 products)" % (t0, 4 * g / t0 * 1e-6))
    t1 = timeit(lambda: ops.attn_bwd(qkv, o, lse, do, Z, want_bias_partials=True, saved_p=(pst, mrun)), n=30)
    print("attn_bwd stored P     %8.1f us  %6.1f TF algorithmic (4 products)" % (t1, 4 * g / t1 * 1e-6))
