#!/usr/bin/env python3
"""Per-layer convolution timings of the CNN front-end at 64 pairs (128 images), channels-last fp32, with the shipped MIOpen
user db: forward, MIOpen's backward-data, backward-data computed as a FORWARD convolution of dY with the flipped /
transposed filter, and backward-weights.  (tuning aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from rel_pose_amd import _env  # noqa: F401  (selects the shipped MIOpen user db)
torch.backends.cudnn.benchmark = True
dev = "cuda"
Z = int(os.environ.get("Z", "128"))
CL = torch.channels_last
DT = torch.bfloat16 if os.environ.get("DTYPE") == "bf16" else torch.float32      # DTYPE=bf16: what a bf16 front-end would cost


def timeit(fn, n=20, warm=4):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

layers = [  # name, Cin, Cout, k, stride, pad, Hin, count per step
    ("resnet.conv1 7x7/2", 3, 64, 7, 2, 3, 224, 1),
    ("layer1 3x3 64->64", 64, 64, 3, 1, 1, 56, 4),
    ("layer2.0.conv1 3x3/2 64->128", 64, 128, 3, 2, 1, 56, 1),
    ("layer2 3x3 128->128", 128, 128, 3, 1, 1, 28, 3),
    ("layer2.0.downsample 1x1/2", 64, 128, 1, 2, 0, 56, 1),
    ("extractor.conv1 3x3 128->192", 128, 192, 3, 1, 1, 28, 1),
    ("extractor.conv2 5x5 valid 192->192", 192, 192, 5, 1, 0, 28, 1),
    ("extractor.downsample 5x5 valid 128->192", 128, 192, 5, 1, 0, 28, 1),
]
tot = {"fwd": 0.0, "bwd": 0.0, "bwd_as_fwd": 0.0, "wrw": 0.0}
for name, ci, co, k, st, pad, hin, cnt in layers:
    x = torch.randn(Z, ci, hin, hin, device=dev).to(DT).contiguous(memory_format=CL)
    w = (torch.randn(co, ci, k, k, device=dev) * 0.05).to(DT).contiguous(memory_format=CL)
    y = F.conv2d(x, w, None, st, pad)
    dy = torch.randn_like(y)
    fl = 2.0 * y.numel() * ci * k * k
    t_f = timeit(lambda: F.conv2d(x, w, None, st, pad))
    bw = lambda mask: torch.ops.aten.convolution_backward(dy, x, w, None, [st, st], [pad, pad], [1, 1], False, [0, 0], 1, mask)
    t_b = timeit(lambda: bw([True, False, False])) if ci > 3 else float("nan")
    t_w = timeit(lambda: bw([False, True, False]))
    t_bf = float("nan")
    if st == 1 and ci > 3:
        wf = w.flip(2, 3).transpose(0, 1).contiguous(memory_format=CL)        # [Cin, Cout, k, k]: dX = conv(dY, flip(W)^T, pad k-1-p)
        dx_ref = bw([True, False, False])[0]
        dx = F.conv2d(dy, wf, None, 1, k - 1 - pad)
        err = float((dx.float() - dx_ref.float()).abs().max() / dx_ref.float().abs().max())
        t_bf = timeit(lambda: F.conv2d(dy, w.flip(2, 3).transpose(0, 1).contiguous(memory_format=CL), None, 1, k - 1 - pad))
        name += "  [bwd-as-fwd err %.1e]" % err
    print("%-62s x%d  %5.1f GF | fwd %7.1f us (%5.1f TF) | bwd-data %7.1f | bwd-as-fwd %7.1f | wrw %7.1f (%5.1f TF)" %
          (name, cnt, fl * 1e-9, t_f, fl / t_f * 1e-6, t_b, t_bf, t_w, fl / t_w * 1e-6), flush=True)
    tot["fwd"] += cnt * t_f
    tot["bwd"] += cnt * (0 if t_b != t_b else t_b)
    tot["bwd_as_fwd"] += cnt * (t_bf if t_bf == t_bf and t_bf < t_b else (0 if t_b != t_b else t_b))
    tot["wrw"] += cnt * t_w
print("per step: fwd %.2f ms, bwd-data %.2f ms (best of both %.2f ms), wrw %.2f ms" % (tot["fwd"] / 1e3, tot["bwd"] / 1e3, tot["bwd_as_fwd"] / 1e3, tot["wrw"] / 1e3))
