import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rel_pose_amd import ops, _lib
_lib.load()
Z = 128; M = Z * 576
def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
qkv = torch.randn(M, 576, device="cuda"); do = torch.randn(M, 192, device="cuda")
o, lse = ops.attn_fwd(qkv, Z)
timeit(lambda: ops.attn_fwd(qkv, Z))
for nw in sys.argv[1:]:
    os.environ["RP_ATTN_FWD"] = nw
    print("NW", nw, "fwd %.1f us  bwd %.1f us" % (timeit(lambda: ops.attn_fwd(qkv, Z)), timeit(lambda: ops.attn_bwd(qkv, o, lse, do, Z))), flush=True)
