"""time rp_attn_fwd_bf16 (and the fp32-storage bf16-MFMA kernel it replaces) at Z images: python tools/attn_bf16_time.py [Z]"""
import sys
import torch
from rel_pose_amd import ops

Z = int(sys.argv[1]) if len(sys.argv) > 1 else 256
qkv = torch.randn(Z * 576, 576, device="cuda")
qb = qkv.to(torch.bfloat16)


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


fl = 4.0 * Z * 3 * 576 * 576 * 64
us = t(lambda: ops.attn_fwd_bf16(qb, Z))
print("attn_fwd_bf16      Z=%d  %.1f us  %.0f TF  %.2f TB/s" % (Z, us, fl / us * 1e-6, 4 * Z * 576 * 192 * 2 / us * 1e-6))
us = t(lambda: ops.attn_fwd_bf16(qb, Z, stats_only=True))
print("attn_stats_bf16    Z=%d  %.1f us" % (Z, us))
o, lse2 = ops.attn_fwd_bf16(qb, Z)
dob = torch.randn(Z * 576, 192, device="cuda").to(torch.bfloat16)
us = t(lambda: ops.attn_bwd_bf16(qb, o, lse2, dob, Z))
print("attn_bwd_bf16 (delta + dkdv + dq)  Z=%d  %.1f us  %.0f TF algorithmic (10 products)" % (Z, us, 2.5 * fl / us * 1e-6))
ops.set_attention_precision(True)
us = t(lambda: ops.attn_fwd(qkv, Z))
print("attn_fwd (fp32 storage, bf16 MFMA)  %.1f us  %.0f TF" % (us, fl / us * 1e-6))
of, lse = ops.attn_fwd(qkv, Z)
do = torch.randn(Z * 576, 192, device="cuda")
us = t(lambda: ops.attn_bwd(qkv, of, lse, do, Z))
print("attn_bwd (fp32 storage, bf16 MFMA, stored dS)  %.1f us" % us)
