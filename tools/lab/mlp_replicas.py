import torch, sys, os
sys.path.insert(0, os.getcwd())
from rel_pose_amd import ops, _lib
_lib.load()
torch.manual_seed(0)
bf = torch.bfloat16
for M0 in (576 * 8, 576 * 64):
    a = torch.randn(M0, 192, device="cuda"); hp = torch.randn(M0, 768, device="cuda")
    dy = torch.cat([a] * 4).contiguous(); hpre = torch.cat([hp] * 4).contiguous()
    w1 = torch.randn(768, 192, device="cuda") * 0.07; w2 = torch.randn(192, 768, device="cuda") * 0.04
    for prec in (0, 1):
        ops.set_gemm_precision(prec)
        for hdt in ((None,) if prec == 0 else (None, bf)):
            h_in = hpre if hdt is None else hpre.to(bf)
            dhp, dxn, part = ops.mlp_fused_bwd(dy, h_in, w1, w2, out_dtype=hdt)
            d = dhp.view(4, M0, 768); x = dxn.view(4, M0, 192)
            print("M0=%d prec=%d io=%s  dhp replicas equal: %s  dxn replicas equal: %s  max|dxn diff| %.3e" % (
                M0, prec, hdt, all(torch.equal(d[0], d[i]) for i in range(1, 4)), all(torch.equal(x[0], x[i]) for i in range(1, 4)),
                max(float((x[0] - x[i]).abs().max()) for i in range(1, 4))))
    ops.set_gemm_precision(0)
