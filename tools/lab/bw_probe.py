#!/usr/bin/env python3
"""HBM bandwidth of plain streams on this box: fill (write only), copy (1 read + 1 write), sum (read only), 3-read-1-write add."""
import torch
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n
for mb in (256, 1024):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, device="cuda"); b = torch.empty_like(a); c = torch.randn(n, device="cuda"); d = torch.randn(n, device="cuda")
    a16 = torch.empty(n, device="cuda", dtype=torch.bfloat16)
    print("%4d MB  fill %.2f TB/s | copy %.2f TB/s | sum %.2f TB/s | a=c+d %.2f TB/s | fp32->bf16 cast %.2f TB/s" % (
        mb, mb * 2**20 / t(lambda: a.fill_(1.0)) / 1e12, 2 * mb * 2**20 / t(lambda: b.copy_(c)) / 1e12,
        mb * 2**20 / t(lambda: c.sum()) / 1e12, 3 * mb * 2**20 / t(lambda: torch.add(c, d, out=a)) / 1e12,
        1.5 * mb * 2**20 / t(lambda: a16.copy_(c)) / 1e12), flush=True)
