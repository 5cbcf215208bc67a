#!/usr/bin/env python3
"""rp_conv3x3_c64_bf16 (csrc/conv3x3_bf16.hip) at 256 images: values against fp64 F.conv2d on the same bf16 operands, and time against
MIOpen's bf16 forward convolution (+ the separate BatchNorm-apply + ReLU pass the fused form replaces)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from rel_pose_amd import _env  # noqa: F401
from rel_pose_amd import ops
torch.backends.cudnn.benchmark = True
N = int(os.environ.get("Z", "256"))
bf, CL = torch.bfloat16, torch.channels_last
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(N, 64, 56, 56, device="cuda", generator=g).to(bf).contiguous(memory_format=CL)
w = (torch.randn(64, 64, 3, 3, device="cuda", generator=g) * (64 * 9) ** -0.5).to(bf).contiguous(memory_format=CL)
scale = 0.5 + torch.rand(64, device="cuda", generator=g)
shift = 0.3 * torch.randn(64, device="cuda", generator=g)
xn, wn = x.permute(0, 2, 3, 1), w.permute(0, 2, 3, 1)          # NHWC views of the same memory
assert xn.is_contiguous() and wn.is_contiguous()


def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())


# ---- values (a slice of images in fp64)
K = min(N, 8)
y, st = ops.conv3x3_c64_bf16(xn, wn, want_stats=True)
ref = F.conv2d(x[:K].double(), w.double(), None, 1, 1).permute(0, 2, 3, 1)
print("plain : rel err vs fp64 %.2e" % rel(y[:K], ref))
yd = y.double()
print("stats : sum %.2e  sumsq %.2e" % (rel(st[:, 0].sum(0), yd.sum((0, 1, 2))), rel(st[:, 1].sum(0), (yd * yd).sum((0, 1, 2)))))
y2 = ops.conv3x3_c64_bf16(xn, wn, scale, shift)
xa = torch.relu(x[:K].float() * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)).to(bf)
ref2 = F.conv2d(xa.double(), w.double(), None, 1, 1).permute(0, 2, 3, 1)
print("bn+relu on load: rel err vs fp64 %.2e" % rel(y2[:K], ref2))
# last images too (tile ranges of the last workgroups)
ref3 = F.conv2d(x[-2:].double(), w.double(), None, 1, 1).permute(0, 2, 3, 1)
print("plain, last images: %.2e" % rel(y[-2:], ref3))
# ---- time
fl = 2.0 * N * 56 * 56 * 64 * 64 * 9
t_mi = timeit(lambda: F.conv2d(x, w, None, 1, 1))
t_apply = timeit(lambda: torch.relu_(x.float().mul_(scale.view(1, -1, 1, 1)).add_(shift.view(1, -1, 1, 1))).to(bf)) if os.environ.get("APPLY") else float("nan")
t0 = timeit(lambda: ops.conv3x3_c64_bf16(xn, wn))
t1 = timeit(lambda: ops.conv3x3_c64_bf16(xn, wn, want_stats=True))
t2 = timeit(lambda: ops.conv3x3_c64_bf16(xn, wn, scale, shift))
t3 = timeit(lambda: ops.conv3x3_c64_bf16(xn, wn, scale, shift, want_stats=True))
dy = torch.randn(N, 56, 56, 64, device="cuda", generator=g).to(bf)
dyc = dy.permute(0, 3, 1, 2)
dw = ops.conv3x3_c64_wgrad_bf16(xn, dy)
dw_mi = torch.ops.aten.convolution_backward(dyc, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
print("wgrad: rel diff vs MIOpen %.2e" % rel(dw, dw_mi.permute(0, 2, 3, 1).double()))
t_wmi = timeit(lambda: torch.ops.aten.convolution_backward(dyc, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]))
t_w = timeit(lambda: ops.conv3x3_c64_wgrad_bf16(xn, dy))
print("N=%d  %.1f GF" % (N, fl * 1e-9))
print("MIOpen wrw                      %7.1f us  %6.1f TF" % (t_wmi, fl / t_wmi * 1e-6))
print("own wgrad (+ partial reduce)    %7.1f us  %6.1f TF" % (t_w, fl / t_w * 1e-6))
print("MIOpen fwd                      %7.1f us  %6.1f TF" % (t_mi, fl / t_mi * 1e-6))
for name, t in (("own plain", t0), ("own + stats epilogue", t1), ("own + BN/ReLU on load", t2), ("own + BN/ReLU on load + stats", t3)):
    print("%-31s %7.1f us  %6.1f TF  %5.2f TB/s algorithmic" % (name, t, fl / t * 1e-6, 2.0 * x.numel() * 2 / t * 1e-6))
