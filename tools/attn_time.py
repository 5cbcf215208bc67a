#!/usr/bin/env python3
"""Sustained timing of the attention / EMM kernels at the 64-pair shapes (tuning aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rel_pose_amd import ops, _lib
_lib.load()
Z = 128
torch.manual_seed(0)
qkv = torch.randn(Z * 576, 576, device="cuda")
do = torch.randn(Z * 576, 192, device="cuda")


def timeit(fn, n=60, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

fl_fwd = 4.0 * 576 * 576 * 64 * 3 * Z
t = timeit(lambda: ops.attn_fwd(qkv, Z))
print("attn_fwd          %8.1f us  %6.1f TF" % (t, fl_fwd / t * 1e-6))
t = timeit(lambda: ops.attn_fwd(qkv, Z, stats_only=True, q_off=0, k_off=192, q_xor=1))
print("attn_fwd stats    %8.1f us  %6.1f TF" % (t, fl_fwd / 2 / t * 1e-6))
o, lse = ops.attn_fwd(qkv, Z)
t = timeit(lambda: ops.attn_bwd(qkv, o, lse, do, Z), n=30)
print("attn_bwd (all)    %8.1f us  %6.1f TF (5 GEMMs algorithmic)" % (t, fl_fwd * 2.5 / t * 1e-6))
intr = torch.tensor([192.0, 192.0, 192.0, 192.0], device="cuda").repeat(Z // 2, 2, 1).contiguous()
pos = ops.posenc(intr, Z // 2, "cuda")
X = ops.emm_build_x(qkv, pos, Z)
rlse, clse = ops.emm_stats(qkv, Z)
t = timeit(lambda: ops.emm_apply(qkv, X, rlse, clse, Z), n=30)
print("emm_apply         %8.1f us" % t)
tt, f = ops.emm_apply(qkv, X, rlse, clse, Z)
df = torch.randn(Z, 3, 96, 96, device="cuda") * 0.01
t = timeit(lambda: ops.emm_backward(qkv, X, tt, rlse, clse, df, Z), n=20)
print("emm_backward(all) %8.1f us" % t)
import subprocess
if len(sys.argv) == 1:
    for v, r in (("32", "0"), ("32", "1"), ("16", "1")):
        subprocess.run([sys.executable, __file__, "x"], env=dict(os.environ, RP_DSMM=v, RP_DSMM_REV=r))
else:
    ta = timeit(lambda: ops.attn_bwd(qkv, o, lse, do, Z), n=30)
    te = timeit(lambda: ops.emm_backward(qkv, X, tt, rlse, clse, df, Z), n=20)
    print("RP_DSMM=%s REV=%s: attn_bwd %8.1f us   emm_backward %8.1f us" % (os.environ.get("RP_DSMM"), os.environ.get("RP_DSMM_REV"), ta, te))
