#!/usr/bin/env python3
"""Phase timing of attn_bwd_dkdv_p_kernel with shader-clock stamps (tools/lab/libattn_probe.so = csrc/attention.hip built with
-DRP_DKDV_PROBE): per-wave average cycles per query tile spent in each phase.  Tuning aid."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lab", "libattn_probe.so"))
P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
f = lib.rp_attn_bwd_dkdv_p
f.argtypes = [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, F, P, P, I, P]
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
def run():
    rc = f(P(b), P(b + 8 * 192), P(do.data_ptr()), P(lse.data_ptr()), P(delta.data_ptr()), P(pst.data_ptr()), P(mrun.data_ptr()),
           P(d + 4 * 192), P(d + 8 * 192), P(ds.data_ptr()), Z, 3, 576, 576, 192, 576, 576, 0.125, None, None, 0, st)
    assert rc == 0, rc
for _ in range(3): run()
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 16)()
lib.rp_debug_probe(out, 1)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): run()
e.record(); torch.cuda.synchronize()
lib.rp_debug_probe(out, 0)
waves = out[8]
names = ["prologue(total/wave)", "prefetch issue", "dP MFMA (32)", "P, dS VALU", "dS tile store", "dV,dK MFMA (64)", "Q/dO -> LDS", "barrier"]
print("%.1f us per launch, %d waves" % (s.elapsed_time(e) / 10 * 1e3, waves // 10))
tot = 0
for i, n in enumerate(names):
    per = out[i] / waves / (1 if i == 0 else 18)
    tot += 0 if i == 0 else per
    print("%-24s %9.0f cycles per %s" % (n, per, "wave" if i == 0 else "tile"))
print("sum per tile %9.0f cycles (MFMA work 96 x 64 = 6144; x2 waves per SIMD = 12288)" % tot)
