"""VERDICT r5 item 5, the gate: rp_dw192_split3 (3-limb bf16 split of fp32 operands on the bf16 matrix pipe) against rp_dw192_f32 (exact
fp32 MFMA) at the weight-gradient shapes of the 64-pair step -- time (HIP events around the C-ABI call, the reduce excluded and included)
and error against fp64.  Gate: >= 1.5x at an error <= the exact kernel's."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rel_pose_amd import ops, _lib

lib = _lib.load()
Z = int(os.environ.get("Z", "128"))
M = Z * 576


def timeit(fn, n=30, warm=5):
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


def raw(fn_name, a, b, ws, nbytes):
    f = getattr(lib, fn_name)
    N = a.shape[1]
    return lambda: _lib.check(f(a.data_ptr(), N, b.data_ptr(), M, N, ws.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream), fn_name)


tot = {"f32": 0.0, "split3": 0.0}
# launches per step at these shapes (5 Blocks + CrossBlock: qkv 576, proj 192, fc1 768, fc2 768)
for name, N, per_step in (("proj [192]", 192, 6), ("qkv [576]", 576, 6), ("fc1/fc2 [768]", 768, 11)):
    g = torch.Generator(device="cpu").manual_seed(N)
    a = torch.randn(M, N, generator=g).cuda()
    b = torch.randn(M, 192, generator=g).cuda()
    nbytes = lib.rp_dw192_f32_workspace_bytes(M, N)
    ws = torch.empty(nbytes // 4, device="cuda")
    t0 = timeit(raw("rp_dw192_f32", a, b, ws, nbytes))
    t1 = timeit(raw("rp_dw192_split3", a, b, ws, nbytes))
    ref = a.double().t() @ b.double()
    ops.DW_SPLIT3 = False
    e0 = ops.linear_dw(a, b)
    tt0 = timeit(lambda: ops.linear_dw(a, b))
    ops.DW_SPLIT3 = True
    e1 = ops.linear_dw(a, b)
    tt1 = timeit(lambda: ops.linear_dw(a, b))
    ops.DW_SPLIT3 = False
    den = float(ref.abs().max())
    err = lambda x: (float((x.double() - ref).abs().max()) / den, float((x.double() - ref).square().mean().sqrt()) / den)
    flops = 2.0 * M * N * 192
    byts = 4.0 * M * (N + 192)
    print("%-14s M=%d  exact fp32: %6.1f us (%5.1f TF)  split3: %6.1f us (%5.1f TF-equivalent, %4.2f TB/s)  ratio %.2fx | with reduce %6.1f / %6.1f us"
          "  | err vs fp64 max/rms: exact %.2e / %.2e   split3 %.2e / %.2e"
          % (name, M, t0, flops / t0 * 1e-6, t1, flops / t1 * 1e-6, byts / t1 * 1e-6, t0 / t1, tt0, tt1, *err(e0), *err(e1)), flush=True)
    tot["f32"] += per_step * t0
    tot["split3"] += per_step * t1
print("per step (23 launches): exact fp32 %.0f us, split3 %.0f us, ratio %.2fx" % (tot["f32"], tot["split3"], tot["f32"] / tot["split3"]))
