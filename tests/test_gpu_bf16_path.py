"""GPU parity tests of the bf16 DATA PATH (BASELINE.json configs[4]; csrc/attention_bf16.hip and the bf16-in / bf16-out forms of the
row-resident kernels), kernel by kernel through the C ABI.  Reference: the same op in fp64 PyTorch on the SAME bf16-rounded inputs,
so the stated tolerances measure the kernels' own roundings (P / dS packed to bf16 per tile, bf16 outputs), not the input rounding.
Measured errors are appended to gpurun_out/test_report.txt."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def report(name, **kv):
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "test_report.txt"), "a") as f:
        f.write(name + ": " + ", ".join("%s=%.3e" % (k, v) for k, v in kv.items()) + "\n")


def rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from rel_pose_amd import _lib, ops as o
    _lib.load()
    return o


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed + int(np.prod(shape)) % 9973)
    return (torch.randn(*shape, generator=g) * scale).cuda()


def _attn_ref(qkv64, Z, k_xor=0):
    q, k, v = (qkv64[:, i * 192:(i + 1) * 192].reshape(Z, 576, 3, 64).permute(0, 2, 1, 3) for i in range(3))
    if k_xor & 1:
        k = k.reshape(Z // 2, 2, 3, 576, 64).flip(1).reshape(Z, 3, 576, 64)
    if k_xor & 2:
        v = v.reshape(Z // 2, 2, 3, 576, 64).flip(1).reshape(Z, 3, 576, 64)
    s = q @ k.transpose(-1, -2) * 64 ** -0.5
    o = (s.softmax(-1) @ v).transpose(1, 2).reshape(Z * 576, 192)
    return o, torch.logsumexp(s, -1)


@pytest.mark.parametrize("Z,nw", [(4, None), (4, "6"), (6, "2"), (44, None)])
def test_attention_forward_bf16_path(ops, Z, nw, monkeypatch):
    """rp_attn_fwd_bf16 vs fp64 softmax attention on the same bf16 q | k | v.  Both workgroup forms (2-wave: small launches; 6-wave,
    96-key stages: the product form, chosen by the launcher from 43 images up).  Tolerances: o is a bf16 output (2^-9 relative to its
    value) of probabilities rounded to bf16 per tile: 1e-2 of max|o|; lse = fp32 maximum + log of the row sum of the ROUNDED
    probabilities (the sum comes off the matrix pipe, and the same rounded values weight v, so o is normalised consistently): 2e-4."""
    if nw:
        monkeypatch.setenv("RP_ATTN_BF16_NW", nw)
    qkv = rnd(Z * 576, 576, seed=1)
    qkv[:, :384] *= 1.7          # sharper softmax
    qkv[5, :64] *= 6.0           # one spiky query row (forces a rescale late in the row)
    qb = qkv.to(torch.bfloat16)
    o, lse = ops.attn_fwd_bf16(qb, Z)
    o_ref, lse_ref = _attn_ref(qb.double(), Z)
    lse = lse * float(np.log(2.0))                                            # the kernel's normaliser is in log2 units
    e_o, e_l = rel(o, o_ref), rel(lse, lse_ref)
    report("attn_fwd_bf16[Z=%d,nw=%s]" % (Z, nw), o=e_o, lse=e_l)
    assert o.dtype == torch.bfloat16 and e_o < 1e-2 and e_l < 2e-4
    o2, lse2 = ops.attn_fwd_bf16(qb, Z)
    assert torch.equal(o, o2) and torch.equal(lse, lse2 * float(np.log(2.0)))  # run-to-run bit-identical
    _, lse_s = ops.attn_fwd_bf16(qb, Z, stats_only=True)
    assert torch.equal(lse_s, lse2)                                          # the statistics-only form is the same arithmetic
    if Z % 2 == 0:
        ox, lx = ops.attn_fwd_bf16(qb, Z, k_xor=3)                           # keys and values of the partner image (--noess)
        ox_ref, lx_ref = _attn_ref(qb.double(), Z, k_xor=3)
        assert rel(ox, ox_ref) < 1e-2 and rel(lx * float(np.log(2.0)), lx_ref) < 2e-4


@pytest.mark.parametrize("Z,nw", [(4, None), (4, "6"), (6, None)])
def test_attention_backward_bf16_path(ops, Z, nw, monkeypatch):
    """rp_attn_bwd_bf16 (+ delta) vs fp64 autograd of softmax attention on the same bf16 q | k | v / dO, with the forward kernel's own
    o and lse2.  Tolerance: P and dS are rounded to bf16 per tile and the gradients are bf16 outputs: 2e-2 of max|grad| per third
    (measured ~5e-3).  Bias-gradient partials come from the fp32 accumulators: they must match the fp64 column sums although
    the bf16 outputs they summarise are individually rounded (2e-2 of the largest block sum: sums of 32 signed terms).  kv_xor = 1 (keys / values of the partner image) on even Z."""
    if nw:
        monkeypatch.setenv("RP_ATTN_BF16_BWD_NW", nw)
    qkv = rnd(Z * 576, 576, seed=1)
    qkv[:, :384] *= 1.7
    qkv[5, :64] *= 6.0
    qb = qkv.to(torch.bfloat16)
    dob = rnd(Z * 576, 192, seed=2).to(torch.bfloat16)
    for kv_xor in (0, 1):
        if kv_xor and Z % 2:
            continue
        o, lse2 = ops.attn_fwd_bf16(qb, Z, k_xor=3 * kv_xor)
        q64 = qb.double().requires_grad_(True)
        o_ref, _ = _attn_ref(q64, Z, k_xor=3 * kv_xor)
        (o_ref * dob.double()).sum().backward()
        dqkv, part = ops.attn_bwd_bf16(qb, o, lse2, dob, Z, kv_xor=kv_xor, want_bias_partials=True)
        e = [rel(dqkv[:, i * 192:(i + 1) * 192], q64.grad[:, i * 192:(i + 1) * 192]) for i in range(3)]
        report("attn_bwd_bf16[Z=%d,nw=%s,kv_xor=%d]" % (Z, nw, kv_xor), dq=e[0], dk=e[1], dv=e[2])
        assert dqkv.dtype == torch.bfloat16 and max(e) < 2e-2
        d2 = ops.attn_bwd_bf16(qb, o, lse2, dob, Z, kv_xor=kv_xor)
        assert torch.equal(d2, dqkv)                                          # deterministic, and the partials do not change the gradients
        blocks = q64.grad.view(Z * 18, 32, 576).sum(1)
        e_p = rel(part, blocks)
        report("attn_bwd_bf16_bias_partials[Z=%d,kv_xor=%d]" % (Z, kv_xor), part=e_p)
        assert part.shape == (Z * 18, 576) and e_p < 2e-2


@pytest.mark.parametrize("M,N", [(4096, 192), (4096 + 640, 576), (9 * 4096, 768)])
@pytest.mark.parametrize("b_f32", [False, True])
def test_weight_gradient_stream_kernel(ops, M, N, b_f32):
    """rp_dw192_bf16 + the split-K reduce through ops.linear_dw (bf16 configuration): dW = dY^T X with the wide operand bf16 and the
    192-wide one bf16 or fp32 (rounded on chip), both orientations (dy wide: direct; x wide: transposed by the reduce = fc2's form).
    Products of bf16 values are exact in fp32 and the accumulation is fp32: 2e-5 of max|dW| against fp64 on the rounded operands
    (a token slab of ~600 rows per workgroup, then a fixed-order sum over the slabs)."""
    ops.set_gemm_precision(1)
    try:
        wide = rnd(M, N, seed=3).to(torch.bfloat16)
        nar = rnd(M, 192, seed=4)
        nar_in = nar if b_f32 else nar.to(torch.bfloat16)
        nar_ref = nar.to(torch.bfloat16).double()          # what the matrix pipe sees either way
        # dy = wide [M,N], x = narrow [M,192] -> [N,192]
        dw = ops.linear_dw(wide, nar_in)
        ref = wide.double().t() @ nar_ref
        e1 = rel(dw, ref)
        # dy = narrow [M,192], x = wide [M,N] -> [192,N]
        dw2 = ops.linear_dw(nar_in, wide)
        e2 = rel(dw2, ref.t())
        report("dw192_bf16[M=%d,N=%d,b_f32=%d]" % (M, N, b_f32), direct=e1, transposed=e2)
        assert dw.shape == (N, 192) and dw2.shape == (192, N) and max(e1, e2) < 2e-5
        assert torch.equal(dw, ops.linear_dw(wide, nar_in))                    # deterministic
        with ops.splitk_batch():                                              # deferred reduce: same bits, filled at exit
            d3 = ops.linear_dw(wide, nar_in)
            d4 = ops.linear_dw(nar_in, wide)
        assert torch.equal(d3, dw) and torch.equal(d4, dw2)
    finally:
        ops.set_gemm_precision(0)


def test_row_resident_linear_bf16_rows_in_and_out(ops):
    """rp_linear_rows192 at precision 1 with io_bf16 bits 0 (x holds bf16 rows), 1 (y bf16) and 3 (xn_out bf16): against fp64 on the
    values the matrix pipe consumes.  proj form: y = x_bf16 W^T + b + residual (fp32 out); qkv form: LayerNorm + Linear with bf16 y
    and bf16 xn; input-gradient form with a bf16 result."""
    ops.set_gemm_precision(1)
    try:
        M = 576 * 3 + 40
        W = rnd(192, 192, seed=5, scale=0.07)
        b = rnd(192, seed=6)
        res = rnd(M, 192, seed=7)
        xb = rnd(M, 192, seed=8).to(torch.bfloat16)
        y = ops.linear(xb, W, b, residual=res)
        ref = xb.double() @ W.to(torch.bfloat16).double().t() + b.double() + res.double()
        e_proj = rel(y, ref)
        assert y.dtype == torch.float32 and e_proj < 2e-6
        Wq = rnd(576, 192, seed=9, scale=0.07)
        bq = rnd(576, seed=10)
        g, be = 1 + 0.1 * rnd(192, seed=11), 0.1 * rnd(192, seed=12)
        x = rnd(M, 192, seed=13)
        yq, xn, mean, rstd = ops.ln_linear(x, g, be, Wq, bq, train=True, out_dtype=torch.bfloat16, xn_dtype=torch.bfloat16)
        xn_ref = torch.nn.functional.layer_norm(x.double(), (192,), g.double(), be.double(), 1e-6)
        e_xn = rel(xn, xn_ref)
        yq_ref = xn.double() @ Wq.to(torch.bfloat16).double().t() + bq.double()          # from the ROUNDED rows the kernel stored
        e_q = rel(yq, yq_ref)
        report("linear_rows_bf16_io", proj=e_proj, xn=e_xn, qkv=e_q)
        assert xn.dtype == torch.bfloat16 and yq.dtype == torch.bfloat16 and e_xn < 4e-3 and e_q < 4e-3       # one bf16 rounding each
        yq32, xn32, m32, r32 = ops.ln_linear(x, g, be, Wq, bq, train=True)
        assert torch.equal(mean, m32) and torch.equal(rstd, r32) and torch.equal(xn, xn32.to(torch.bfloat16))
        assert torch.equal(yq, yq32.to(torch.bfloat16))                       # same arithmetic, only the stores differ
        dy = rnd(M, 192, seed=14)
        do = ops.linear_dx(dy, W, out_dtype=torch.bfloat16)
        assert do.dtype == torch.bfloat16 and torch.equal(do, ops.linear_dx(dy, W).to(torch.bfloat16))
    finally:
        ops.set_gemm_precision(0)


def _emm_ref(qkv64, pos, Z):
    """fp64 F_z = X^T A X, T = A X, U = A^T X per (z,h) (tests/test_gpu_kernels.py::_emm_ref; X from the given, possibly rounded, qkv)"""
    t = qkv64.view(Z, 576, 3, 3, 64).permute(2, 0, 3, 1, 4)
    q, k, v = t[0], t[1], t[2]
    perm = [z ^ 1 for z in range(Z)]
    s = (q[perm] @ k.transpose(-1, -2)) * 0.125
    a = s.softmax(-1) * s.softmax(-2)
    pe = pos.to(torch.bfloat16).double()[[z // 2 for z in range(Z)]].unsqueeze(1).expand(Z, 3, 576, 6)      # X holds bf16 positional features
    x = torch.cat([v, pe], dim=-1)
    T = a @ x
    return x.transpose(-1, -2) @ T, T, a.transpose(-1, -2) @ x, x


def test_emm_bf16_path_forward_and_backward(ops):
    """csrc/emm_bf16.hip against fp64 on the same bf16 q | k | v (and bf16-rounded positional features): X bit-exact; T = A X and
    U = A^T X (bf16 outputs of bf16 probabilities) 1e-2 of their maxima; F = X^T T and the reshaped g 1e-2; the gradient thirds dq, dk,
    dv of <F, dF> 3e-2 of their maxima (two bf16-rounded intermediates, W / T / U, per product)."""
    Z = 4
    qkv = rnd(Z * 576, 576, seed=4)
    intr = torch.tensor([[30.0, 26.0, 12.0, 12.0], [18.0, 21.0, 12.0, 9.0]])[:, None, :].repeat(1, 2, 1).contiguous().cuda()
    pos = ops.posenc(intr, Z // 2, qkv.device)
    qb = qkv.to(torch.bfloat16)
    q64 = qb.double().requires_grad_(True)
    F_ref, T_ref, U_ref, x_ref = _emm_ref(q64, pos, Z)
    g, (xa, t, rlse2, clse2) = ops.emm_forward_bf16(qb, pos, Z)
    assert torch.equal(xa[..., :70].double(), x_ref.detach()) and float(xa[..., 70:].float().abs().max()) == 0.0
    lib = ops._lib.load()
    u = torch.empty_like(t)
    ops._lib.check(lib.rp_emm_apply_bf16(ops._p(qb), 576, ops._p(xa), ops._p(rlse2), ops._p(clse2), ops._p(u), Z, 3, 0.125, 1, ops._st()), "swap")
    g_ref = F_ref[[z ^ 1 for z in range(Z)]].reshape(Z, 210, 70).transpose(-1, -2)      # vision_transformer.py:229-230,238
    e = dict(T=rel(t[..., :70], T_ref), U=rel(u[..., :70], U_ref), g=rel(g.view(Z, 70, 224)[..., :210], g_ref))
    report("emm_bf16_fwd", **e)
    assert max(e.values()) < 1e-2
    assert float(g.view(Z, 70, 224)[..., 210:].abs().max()) == 0.0 and float(t[..., 70:].float().abs().max()) == 0.0
    dF = torch.zeros(Z, 3, 96, 96, device="cuda")
    dF[..., :70, :70] = rnd(Z, 3, 70, 70, seed=6)
    (F_ref * dF[..., :70, :70].double()).sum().backward()
    dqkv = ops.emm_backward_bf16(qb, xa, t, rlse2, clse2, dF, Z)
    eb = [rel(dqkv[:, i * 192:(i + 1) * 192], q64.grad[:, i * 192:(i + 1) * 192]) for i in range(3)]
    report("emm_bf16_bwd", dq=eb[0], dk=eb[1], dv=eb[2])
    assert dqkv.dtype == torch.bfloat16 and max(eb) < 3e-2
    assert torch.equal(dqkv, ops.emm_backward_bf16(qb, xa, t, rlse2, clse2, dF, Z))        # deterministic


@pytest.mark.parametrize("M", [576 * 2, 576 * 3 + 40])
@pytest.mark.parametrize("with_add", [True, False])
def test_qkv_input_gradient_with_layernorm_backward_bf16(ops, M, with_add):
    """rp_dx_lnbwd_bf16 through ops.linear_dx_lnbwd (bf16 dY [M,576]): dx, dgamma, dbeta (and the column sums of `add`) against fp64
    autograd of y = LayerNorm(x) W^T with the rounded operands.  fp32 accumulation of exact bf16 products, fp32 LayerNorm algebra:
    2e-5 of the maxima (a ragged last tile included)."""
    ops.set_gemm_precision(1)
    try:
        W = rnd(576, 192, seed=21, scale=0.07)
        x = rnd(M, 192, seed=22)
        g, be = 1 + 0.1 * rnd(192, seed=23), 0.1 * rnd(192, seed=24)
        dy = rnd(M, 576, seed=25).to(torch.bfloat16)
        add = rnd(M, 192, seed=26) if with_add else None
        _, mean, rstd = ops.layernorm_fwd(x, g, be)
        out = ops.linear_dx_lnbwd(dy, W, x, g, mean, rstd, add=add)
        x64 = x.double().requires_grad_(True)
        g64, b64 = g.double().requires_grad_(True), be.double().requires_grad_(True)
        xn = torch.nn.functional.layer_norm(x64, (192,), g64, b64, 1e-6)
        (xn @ W.to(torch.bfloat16).double().t() * dy.double()).sum().backward()
        dx_ref = x64.grad + (add.double() if with_add else 0)
        e = dict(dx=rel(out[0], dx_ref), dgamma=rel(out[1], g64.grad), dbeta=rel(out[2], b64.grad))
        if with_add:
            e["colsum_add"] = rel(out[3], add.double().sum(0))
        report("dx_lnbwd_bf16[M=%d,add=%d]" % (M, with_add), **e)
        assert max(e.values()) < 2e-5
        out2 = ops.linear_dx_lnbwd(dy, W, x, g, mean, rstd, add=add)
        assert all(torch.equal(a, b) for a, b in zip(out, out2))
    finally:
        ops.set_gemm_precision(0)


def _same(a, b):
    if isinstance(a, (tuple, list)):
        return all(_same(u, v) for u, v in zip(a, b) if u is not None)
    return torch.equal(a, b)


def test_bf16_path_kernels_are_bit_reproducible_at_full_size(ops):
    """Every kernel of the bf16 data path, at BASELINE.json configs[4]'s per-GPU size (256 images, M = 147 456 token rows), launched
    eight times on the same inputs: bit-identical outputs.  The kernels stage operands by LDS-DMA into rings whose slots are refilled
    right behind a workgroup barrier; the small-size determinism checks above cannot see a slot being refilled while a slow wave's
    LDS reads of it are still in flight (found in round 4: rp_dx_lnbwd_bf16 corrupted a 16-unit block of ~9 of 2304 row tiles per
    launch under full load -- its barrier wait lacked lgkmcnt(0) and the unrolled loop let the MFMAs consuming the previous chunk's
    reads be scheduled behind the barrier)."""
    ops.set_gemm_precision(1)
    ops.set_attention_precision(1)
    try:
        Z = 256
        M = Z * 576
        x, gm, bt = rnd(M, 192, seed=31), 1 + 0.1 * rnd(192, seed=32), 0.1 * rnd(192, seed=33)
        W = rnd(576, 192, seed=34, scale=0.07)
        dyb = rnd(M, 576, seed=35).to(torch.bfloat16)
        add = rnd(M, 192, seed=36)
        _, mean, rstd = ops.layernorm_fwd(x, gm, bt)
        w1, b1 = rnd(768, 192, seed=37, scale=0.07), 0.1 * rnd(768, seed=38)
        w2, b2 = rnd(192, 768, seed=39, scale=0.04), 0.1 * rnd(192, seed=40)
        qb = rnd(M, 576, seed=41).to(torch.bfloat16)
        dob = rnd(M, 192, seed=42).to(torch.bfloat16)
        intr = torch.tensor([[30.0, 26.0, 12.0, 12.0]]).repeat(Z // 2, 2, 1).contiguous().cuda()
        pos = ops.posenc(intr, Z // 2, qb.device)
        dF = torch.zeros(Z, 3, 96, 96, device="cuda")
        dF[..., :70, :70] = rnd(Z, 3, 70, 70, seed=43)
        o, lse2 = ops.attn_fwd_bf16(qb, Z)
        g, (xa, t, rl, cl) = ops.emm_forward_bf16(qb, pos, Z)
        hpre = ops.mlp_fused(x, gm, bt, w1, b1, w2, b2, train=True, out_dtype=torch.bfloat16, xn_dtype=torch.bfloat16)[5]
        xnb = x.to(torch.bfloat16)
        cases = {
            "dx_lnbwd": lambda: ops.linear_dx_lnbwd(dyb, W, x, gm, mean, rstd, add=add),
            "attn_fwd": lambda: ops.attn_fwd_bf16(qb, Z),
            "attn_bwd": lambda: ops.attn_bwd_bf16(qb, o, lse2, dob, Z),
            "emm_fwd": lambda: ops.emm_forward_bf16(qb, pos, Z)[0],
            "emm_bwd": lambda: ops.emm_backward_bf16(qb, xa, t, rl, cl, dF, Z),
            "mlp_fwd": lambda: ops.mlp_fused(x, gm, bt, w1, b1, w2, b2, train=True, out_dtype=torch.bfloat16, xn_dtype=torch.bfloat16),
            "mlp_bwd": lambda: ops.mlp_fused_bwd(add, hpre, w1, w2, out_dtype=torch.bfloat16),
            "qkv_fwd": lambda: ops.ln_linear(x, gm, bt, W, 0.1 * gm.repeat(3), train=True, out_dtype=torch.bfloat16, xn_dtype=torch.bfloat16),
            "dw_bf16": lambda: ops.linear_dw(dyb, xnb),
            "dw_f32b": lambda: ops.linear_dw(dyb, x),
        }
        bad = {}
        for name, fn in cases.items():
            ref = fn()
            torch.cuda.synchronize()
            n = 0
            for _ in range(8):
                out = fn()
                torch.cuda.synchronize()
                n += 0 if _same(ref, out) else 1
            bad[name] = float(n)
        report("bf16_path_reproducible_256_images", **bad)
        assert not any(bad.values()), bad
    finally:
        ops.set_gemm_precision(0)
        ops.set_attention_precision(0)
