#!/usr/bin/env python3
"""What would an implicit-GEMM convolution on rp_gemm's LDS-DMA main loop run at?  Times the DENSE GEMM of each CNN layer's im2col
shape ([pixels, taps*Cin] x [Cout, taps*Cin]^T; forward / backward-data form <0,0>, backward-weights form <1,1> with split-K) --
an upper bound for a gather version of the same loop -- next to the MIOpen numbers of profiles/r2_conv_probe.txt.  Tuning aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rel_pose_amd import ops, _lib
_lib.load()
def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
Z = 128
layers = [("layer1 3x3 64->64", 64, 64, 9, 56 * 56, 267, 307, 301), ("layer2 3x3 128->128", 128, 128, 9, 28 * 28, 267, 263, 260),
          ("extractor.conv1 3x3 128->192", 128, 192, 9, 28 * 28, 385, 385, 372), ("extractor.conv2 5x5 192->192", 192, 192, 25, 24 * 24, 1009, 1361, 1071),
          ("extractor.downsample 5x5 128->192", 128, 192, 25, 24 * 24, 727, 917, 702)]
for name, ci, co, taps, pix, mf, mb, mw in layers:
    M, K, N = Z * pix, taps * ci, co
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.05; dy = torch.randn(M, N, device="cuda")
    fl = 2.0 * M * K * N
    tf = timeit(lambda: ops.gemm(A, W, M, N, K))
    tw = timeit(lambda: ops.gemm(dy, A, N, K, M, a_layout=1, b_layout=1))
    print("%-36s %6.1f GF | dense fwd-form %7.1f us (%5.1f TF)  [MIOpen fwd %d / bwd %d] | dense wrw-form %7.1f us (%5.1f TF) [MIOpen %d]"
          % (name, fl * 1e-9, tf, fl / tf * 1e-6, mf, mb, tw, fl / tw * 1e-6, mw), flush=True)
    del A, W, dy
