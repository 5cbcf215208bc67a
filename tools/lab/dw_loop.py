"""dw192_f32_kernel (the headline's roofline kernel: fc1 / fc2 weight gradients, M = 73 728 x 768) in a loop, HIP-event timed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rel_pose_amd import ops, _lib
_lib.load()
M = 128 * 576
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dY = torch.randn(M, 768, device="cuda")
x = torch.randn(M, 192, device="cuda")
for _ in range(5):
    ops.linear_dw(dY, x)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.time()
s.record()
for _ in range(n):
    ops.linear_dw(dY, x)
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) / n * 1e3
print("linear_dw (dw192_f32_kernel + split-K reduce) x %d: %.1f us per call, %.1f TF incl. the reduce; wall %.1f s" % (n, us, 2.0 * M * 768 * 192 / us * 1e-6, time.time() - t0))
