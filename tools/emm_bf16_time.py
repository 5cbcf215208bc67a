"""time the EMM cluster on the bf16 data path vs the fp32-storage bf16-MFMA kernels: PYTHONPATH=. python tools/emm_bf16_time.py [Z]"""
import sys
import torch
from rel_pose_amd import ops

Z = int(sys.argv[1]) if len(sys.argv) > 1 else 256
qkv = torch.randn(Z * 576, 576, device="cuda")
qb = qkv.to(torch.bfloat16)
intr = torch.tensor([192.0, 192.0, 192.0, 192.0], device="cuda").repeat(Z // 2, 2, 1).contiguous()
pos = ops.posenc(intr, Z // 2, "cuda")
dF = torch.randn(Z, 3, 96, 96, device="cuda") * 0.01


def t(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


g, (xa, tt, r2, c2) = ops.emm_forward_bf16(qb, pos, Z)
print("emm_forward_bf16  (2 stats + build_x + apply + F + finalize)  %.1f us" % t(lambda: ops.emm_forward_bf16(qb, pos, Z)))
print("emm_backward_bf16 (W + apply(swap) + dX + 2 grad passes)      %.1f us" % t(lambda: ops.emm_backward_bf16(qb, xa, tt, r2, c2, dF, Z)))
ops.set_attention_precision(True)
ops.set_gemm_precision(1)


def fwd_old():
    rl, cl = ops.emm_stats(qkv, Z)
    x = ops.emm_build_x(qkv, pos, Z)
    tt_, fp = ops.emm_apply(qkv, x, rl, cl, Z)
    return ops.emm_finalize(fp, Z), x, tt_, rl, cl


_, x, tt_, rl, cl = fwd_old()
print("fp32-storage forward   %.1f us" % t(fwd_old))
print("fp32-storage backward  %.1f us" % t(lambda: ops.emm_backward(qkv, x, tt_, rl, cl, dF, Z)))
