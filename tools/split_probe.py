"""Accuracy and speed of rp_gemm's operand precisions (0 = exact fp32 MFMA, 3 = split-bf16 3 limbs, 1 = bf16 operands)
on the ViT block's GEMM shapes: forward (NT), input gradient (NN) and weight gradient (TN, split-K)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rel_pose_amd import ops

M = 128 * 576


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def err(c, ref):
    d = c.double() - ref
    return float(d.abs().max() / ref.abs().max()), float((d ** 2).mean().sqrt() / (ref ** 2).mean().sqrt())


torch.manual_seed(0)
tiles = os.environ.get("TILES", "").split() or [None]
for name, (N, K) in {"qkv": (576, 192), "proj": (192, 192), "fc1": (768, 192), "fc2": (192, 768)}.items():
    x = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda") * 0.1
    dy = torch.randn(M, N, device="cuda")
    R = 4096
    ref_f = x[:R].double() @ W.double().t() + b.double()
    ref_dx = dy[:R].double() @ W.double()
    ref_dw = dy.double().t() @ x.double()
    for tile in tiles:
        if tile:
            os.environ["RP_GEMM_TILE"] = tile
        for prec in (0, 3, 1):
            ops.set_gemm_precision(prec)
            f = ops.linear(x, W, b)
            dx = ops.linear_dx(dy, W)
            dw = ops.linear_dw(dy, x)
            ef, edx, edw = err(f[:R], ref_f), err(dx[:R], ref_dx), err(dw, ref_dw)
            tf = timeit(lambda: ops.linear(x, W, b))
            tdx = timeit(lambda: ops.linear_dx(dy, W))
            tdw = timeit(lambda: ops.linear_dw(dy, x))
            print("%-4s tile=%-4s prec=%d | fwd %6.1f us rms %.1e max %.1e | dX %6.1f us rms %.1e max %.1e | dW %6.1f us rms %.1e max %.1e"
                  % (name, tile, prec, tf, ef[1], ef[0], tdx, edx[1], edx[0], tdw, edw[1], edw[0]), flush=True)
ops.set_gemm_precision(0)
