import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rel_pose_amd import ops, _lib
_lib.load()
def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for Z in (14, 28, 56, 57, 64, 86, 114, 128, 170, 228, 256):
    qkv = torch.randn(Z * 576, 576, device="cuda")
    t = timeit(lambda: ops.attn_fwd(qkv, Z))
    fl = 4.0 * Z * 3 * 576 * 576 * 64
    print("Z=%3d WGs=%5d fwd %.1f us  %.1f TF  (%.2f us per image)" % (Z, 18 * Z, t, fl / t / 1e6, t / Z), flush=True)
