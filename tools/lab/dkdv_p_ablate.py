#!/usr/bin/env python3
"""lab: attn_bwd_dkdv_p_kernel<2,2> with phases removed (wrong results, same MFMA count): NOP = no P-tile loads, NOD = no dS tile store,
NOQ = no Q / dO prefetch + LDS staging.  libattn_abl_*.so are hand-built from a patched copy of csrc/attention.hip (not committed)."""
import ctypes, os, sys
import torch
P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
Z = 128
torch.manual_seed(0)
qkv = torch.randn(Z * 576, 576, device="cuda")
do = torch.randn(Z * 576, 192, device="cuda")
lse = torch.randn(Z, 3, 576, device="cuda") + 5
delta = torch.randn(Z, 3, 576, device="cuda")
pst = torch.rand(Z, 3, 18, 18, 1024, device="cuda")
mrun = torch.randn(Z, 3, 18, 576, device="cuda")
dqkv = torch.empty_like(qkv)
ds = torch.empty(Z, 3, 576, 576, device="cuda")
b, d = qkv.data_ptr(), dqkv.data_ptr()
st = P(torch.cuda.current_stream().cuda_stream)
here = os.path.dirname(os.path.abspath(__file__))
for rnd in range(2):
    for v in sys.argv[1:]:
        lib = ctypes.CDLL(os.path.join(here, "libattn_abl_%s.so" % v))
        f = lib.rp_attn_bwd_dkdv_p
        f.argtypes = [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, F, P, P, I, P]
        f.restype = I

        def run():
            rc = f(P(b), P(b + 8 * 192), P(do.data_ptr()), P(lse.data_ptr()), P(delta.data_ptr()), P(pst.data_ptr()), P(mrun.data_ptr()),
                   P(d + 4 * 192), P(d + 8 * 192), P(ds.data_ptr()), Z, 3, 576, 576, 192, 576, 576, 0.125, None, None, 0, st)
            assert rc == 0, rc
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(100):
            run()
        e.record()
        torch.cuda.synchronize()
        print("%-12s %.1f us per launch" % (v, s.elapsed_time(e) / 100 * 1e3), flush=True)
