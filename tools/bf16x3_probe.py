import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rel_pose_amd import ops, _lib
lib = _lib.load()
M = 128 * 576
def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
def x3(A, W, b=None, act=0, res=None):
    Mm, K = A.shape; N = W.shape[0]
    C = torch.empty(Mm, N, device=A.device)
    _lib.check(lib.rp_gemm_nt_bf16x3(ops._p(A), ops._p(W), ops._p(C), Mm, N, K, K, K, N, ops._p(b), ops._p(res), None, act, ops._st()), "x3")
    return C
torch.manual_seed(0)
for name, (N, K) in {"qkv": (576, 192), "proj": (192, 192), "fc1": (768, 192), "fc2": (192, 768)}.items():
    A = torch.randn(M, K, device="cuda").abs_(); W = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda") * 0.1
    ref = A[:4096].double() @ W.double().t() + b.double()
    c32 = ops.linear(A, W, b); c3 = x3(A, W, b)
    e32 = float((c32[:4096].double() - ref).abs().max() / ref.abs().max()); e3 = float((c3[:4096].double() - ref).abs().max() / ref.abs().max())
    rms32 = float(((c32[:4096].double() - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt()); rms3 = float(((c3[:4096].double() - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt())
    t32 = timeit(lambda: ops.linear(A, W, b)); t3 = timeit(lambda: x3(A, W, b))
    print("%-5s fp32-MFMA %.0f us (max %.1e rms %.1e) | bf16x3 %.0f us (max %.1e rms %.1e)  speedup %.2fx  %.0f TF-equiv  %.2f TB/s" %
          (name, t32, e32, rms32, t3, e3, rms3, t32 / t3, 2.0 * M * N * K / t3 / 1e6, 4.0 * (M * K + M * N) / t3 / 1e6), flush=True)
