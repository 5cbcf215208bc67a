import torch
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for mb in (340, 906):
    n = mb * 1000 * 1000 // 4
    x = torch.empty(n, device="cuda"); y = torch.empty(n, device="cuda")
    tf = t(lambda: x.fill_(1.0)); tc = t(lambda: y.copy_(x)); ts = t(lambda: x.sum())
    print("%d MB: fill %.1f us (%.2f TB/s)  copy %.1f us (%.2f TB/s moved)  read-sum %.1f us (%.2f TB/s)" % (mb, tf, mb / tf, tc, 2 * mb / tc, ts, mb / ts))
