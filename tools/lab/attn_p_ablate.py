#!/usr/bin/env python3
"""lab: stored-P attention kernels (rp_attn_fwd_savep, rp_attn_bwd_dkdv_p, rp_ds_matmul_t) from hand-built variants of csrc/attention.hip
(tools/lab/libattn_abl_<NAME>.so, not committed; wrong results, same MFMA count).  usage: attn_p_ablate.py NAME ..."""
import ctypes, os, sys
import torch
P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
Z = 128
torch.manual_seed(0)
qkv = torch.randn(Z * 576, 576, device="cuda")
do = torch.randn(Z * 576, 192, device="cuda")
o = torch.empty(Z * 576, 192, device="cuda")
lse = torch.randn(Z, 3, 576, device="cuda") + 5
delta = torch.randn(Z, 3, 576, device="cuda")
pst = torch.rand(Z, 3, 18, 18, 1024, device="cuda")
mrun = torch.randn(Z, 3, 18, 576, device="cuda")
dqkv = torch.empty_like(qkv)
ds = torch.empty(Z, 3, 576, 576, device="cuda")
b, d = qkv.data_ptr(), dqkv.data_ptr()
st = P(torch.cuda.current_stream().cuda_stream)
here = os.path.dirname(os.path.abspath(__file__))


def timeit(run, n=100):
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        run()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for rnd in range(2):
    for v in sys.argv[1:]:
        lib = ctypes.CDLL(os.path.join(here, "libattn_abl_%s.so" % v))
        f = lib.rp_attn_bwd_dkdv_p
        f.argtypes = [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, F, P, P, I, P]
        g = lib.rp_attn_fwd_savep
        g.argtypes = [P, P, P, P, P, P, P, I, I, I, I, I, I, F, P]
        m = lib.rp_ds_matmul_t
        m.argtypes = [P, P, P, I, I, I, I, I, P, I, P]
        w = lib.rp_attn_fwd
        w.argtypes = [P, P, P, P, P, I, I, I, I, I, I, I, I, F, I, I, P]

        def bwd():
            assert f(P(b), P(b + 8 * 192), P(do.data_ptr()), P(lse.data_ptr()), P(delta.data_ptr()), P(pst.data_ptr()), P(mrun.data_ptr()),
                     P(d + 4 * 192), P(d + 8 * 192), P(ds.data_ptr()), Z, 3, 576, 576, 192, 576, 576, 0.125, None, None, 0, st) == 0

        def fwd():
            assert g(P(b), P(b + 4 * 192), P(b + 8 * 192), P(o.data_ptr()), P(lse.data_ptr()), P(pst.data_ptr()), P(mrun.data_ptr()), Z, 3,
                     576, 576, 576, 192, 0.125, st) == 0

        def fwd0():
            assert w(P(b), P(b + 4 * 192), P(b + 8 * 192), P(o.data_ptr()), P(lse.data_ptr()), Z, 3, 576, 576, 576, 192, 0, 0, 0.125, 0, 0, st) == 0

        def dsm():
            assert m(P(ds.data_ptr()), P(b + 4 * 192), P(d), Z, 3, 576, 576, 0, None, 0, st) == 0

        print("%-8s fwd %.1f  fwd_savep %.1f  dkdv_p %.1f  ds_matmul_t %.1f us" % (v, timeit(fwd0), timeit(fwd), timeit(bwd), timeit(dsm)), flush=True)
