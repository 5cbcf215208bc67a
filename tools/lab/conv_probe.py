"""per-phase cycle counts of conv3x3_c64_kernel (built with -DRP_CONV_PROBE into tools/lab/libconvprobe.so)"""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
so = os.path.join(ROOT, "tools", "lab", "libconvprobe.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DRP_CONV_PROBE",
                       os.path.join(ROOT, "rel_pose_amd", "csrc", "conv3x3_bf16.hip"), "-o", so])
lib = ctypes.CDLL(so)
N = int(os.environ.get("Z", "256"))
bf = torch.bfloat16
x = torch.randn(N, 56, 56, 64, device="cuda").to(bf)
w = (torch.randn(64, 3, 3, 64, device="cuda") * 0.04).to(bf)
y = torch.empty_like(x)
st = torch.zeros(256, 8, 4, device="cuda", dtype=torch.float64)
P = lambda t: ctypes.c_void_p(t.data_ptr())
sc, sh = torch.ones(64, device="cuda"), torch.zeros(64, device="cuda")
for name, a, b in (("plain", None, None), ("bn", sc, sh)):
    for _ in range(3):
        lib.rp_conv3x3_c64_bf16(P(x), P(w), P(y), P(a) if a is not None else None, P(b) if b is not None else None, P(st), N, 56, 56, None)
    torch.cuda.synchronize()
    s = st[:, :4].mean((0, 1)).tolist()
    n = s[3]
    print("%s: per tile (cycles, s_memtime @100MHz? raw units): loop %.0f  epilogue %.0f  barrier %.0f  tiles %.1f" % (name, s[0] / n, s[1] / n, s[2] / n, n))
    print("   per-wave loop avg:", [round(v, 0) for v in (st[:, :4, 0].mean(0) / n).tolist()])
