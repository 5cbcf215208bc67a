#!/usr/bin/env python3
"""rp_ds_matmul variants (RP_DSMM=<NW><VAR>) at the 64-pair shape; checks the result against torch.bmm."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    from rel_pose_amd import ops, _lib
    _lib.load()
    Z = 128
    torch.manual_seed(0)
    qkv = torch.randn(Z * 576, 576, device="cuda")
    ds = torch.randn(Z, 3, 576, 576, device="cuda")
    dq = torch.zeros_like(qkv)
    f = lambda: ops.ds_matmul(ds, qkv.data_ptr() + 4 * 192, 576, dq.data_ptr(), 576, Z)
    for _ in range(5): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50): f()
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 50 * 1e3
    k = qkv.view(Z, 576, 576)[:, :, 192:384].reshape(Z, 576, 3, 64).permute(0, 2, 1, 3)
    # ds is TILED: [Z,3,18 iblk,18 jblk,16 r,64 lane] with (r, lane) -> i = (r&3) + 8 (r>>2) + 4 (lane>>5), j = lane & 31
    r_ = torch.arange(16, device="cuda")[:, None]; l_ = torch.arange(64, device="cuda")[None, :]
    ii = ((r_ & 3) + 8 * (r_ >> 2) + 4 * (l_ >> 5)).reshape(-1); jj = (l_ & 31).expand(16, 64).reshape(-1)
    dense = torch.empty(Z, 3, 18, 18, 32, 32, device="cuda")
    dense[:, :, :, :, ii, jj] = ds.view(Z, 3, 18, 18, 1024)
    dense = dense.permute(0, 1, 2, 4, 3, 5).reshape(Z, 3, 576, 576)
    ref = torch.matmul(dense.double(), k.double()).permute(0, 2, 1, 3).reshape(Z * 576, 192)
    err = float((dq[:, :192].double() - ref).abs().max() / ref.abs().max())
    ds16 = ds.to(torch.bfloat16); dq16 = torch.zeros_like(qkv)
    f16 = lambda: ops.ds_matmul(ds16, qkv.data_ptr() + 4 * 192, 576, dq16.data_ptr(), 576, Z)
    for _ in range(5): f16()
    torch.cuda.synchronize(); s.record()
    for _ in range(50): f16()
    e.record(); torch.cuda.synchronize()
    t16 = s.elapsed_time(e) / 50 * 1e3
    print("bf16 tiles: %8.1f us  %5.2f TB/s  err vs fp64 %.1e" % (t16, Z * 3 * 576 * 576 * 2 / t16 * 1e-6, float((dq16[:, :192].double() - ref).abs().max() / ref.abs().max())))
    print("RP_DSMM=%s  %8.1f us  %6.1f TF  %5.2f TB/s  err %.1e" % (os.environ.get("RP_DSMM"), t, 2.0 * Z * 3 * 576 * 576 * 64 / t * 1e-6, Z * 3 * 576 * 576 * 4 / t * 1e-6, err))
else:
    for v in ("32", "16"):
        subprocess.run([sys.executable, __file__, "x"], env=dict(os.environ, RP_DSMM=v))
