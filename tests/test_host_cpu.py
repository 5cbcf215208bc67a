"""CPU-only checks: the C-ABI library loads and exports every symbol include/relpose_hip.h declares (no compute),
host-side logic (module tree / state_dict contract, tile + split-K selection, SE3 algebra, loss), and that the
product path fails loudly without a GPU."""
import ctypes
import json
import os
import re
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_args(**kw):
    a = types.SimpleNamespace(noess="", pool_size=60, fc_hidden_size=512, fusion_transformer=True, transformer_depth=6,
                              cross_features=False, use_single_softmax=False, no_pos_encoding=False,
                              l1_pos_encoding=False)
    a.__dict__.update(kw)
    return a


def test_library_exports_every_declared_symbol():
    from rel_pose_amd import _lib
    header = open(os.path.join(ROOT, "include", "relpose_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(rp_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    lib = ctypes.CDLL(_lib.lib_path())
    for sym in declared:
        assert hasattr(lib, sym), "missing export: " + sym
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    typed = _lib.load()
    assert typed.rp_abi_version() == _lib.ABI_VERSION and typed.rp_target_arch() == b"gfx950"
    assert typed.rp_gemm_workspace_bytes(576, 192, 103) == 103 * 576 * 192 * 4
    assert typed.rp_layernorm_bwd_blocks(73728) == 1152 and typed.rp_layernorm_bwd_blocks(100) == 2


def test_state_dict_contract_and_dropin_alias():
    from rel_pose_amd.model import ViTEss
    from src.model import ViTEss as Alias
    assert Alias is ViTEss
    m = ViTEss(make_args())
    with open(os.path.join(ROOT, "tests", "golden", "state_dict_keys.json")) as f:
        ref = json.load(f)
    sd = m.state_dict()
    assert set(sd) == set(ref) and all(list(sd[k].shape) == ref[k] for k in ref)
    assert sd["extractor_final_conv.downsample.1.weight"].data_ptr() == sd["extractor_final_conv.norm3.weight"].data_ptr()
    assert sum(p.numel() for p in m.parameters()) == 29751950
    frozen = list(m.resnet.layer3.parameters()) + list(m.resnet.layer4.parameters())       # train.py:60-64
    trainable = sum(p.numel() for p in m.parameters()) - sum(p.numel() for p in frozen)
    assert trainable == 19258510                                                           # 77.0 MB all-reduce
    # checkpoints saved under DDP carry a "module." prefix (train.py:191-194, demo.py:60-61)
    m2 = ViTEss(make_args())
    m2.load_state_dict({k.replace("module.", ""): v for k, v in {"module." + k: v for k, v in sd.items()}.items()})


def test_unsupported_variants_are_rejected_loudly():
    from rel_pose_amd.model import ViTEss
    # both crash inside the reference itself (vision_transformer.py:179 vs :225-227; src/model.py:63-71 vs :189)
    for kw in (dict(no_pos_encoding=True), dict(fusion_transformer=False)):
        with pytest.raises(NotImplementedError):
            ViTEss(make_args(**kw))


def test_noess_model_has_the_reference_state_dict():
    import json
    from rel_pose_amd.model import ViTEss
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "state_dict_keys_noess.json")) as f:
        ref = json.load(f)
    m = ViTEss(make_args(noess="1"))
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == ref
    assert m.H == 24768 and m.pose_regressor[0].in_features == 24768      # src/model.py:74


def test_no_cpu_fallback():
    from rel_pose_amd import ops
    from rel_pose_amd.model import ViTEss
    with pytest.raises(RuntimeError):
        ops.linear(torch.zeros(8, 192), torch.zeros(192, 192))
    if not torch.cuda.is_available():
        m = ViTEss(make_args()).eval()
        with pytest.raises(RuntimeError):
            m.forward_tokens(torch.zeros(2, 192, 24, 24), torch.zeros(1, 2, 7), None)


def test_update_intrinsics_mutates_caller_tensor():
    from rel_pose_amd.model import ViTEss
    m = ViTEss(make_args())
    intr = torch.tensor([[[517.97, 517.97, 320.0, 240.0]] * 2])
    out = m.update_intrinsics((1, 2, 3, 384, 512), intr)
    assert out is intr
    assert torch.allclose(intr[0, 0], torch.tensor([517.97 * 24 / 512, 517.97 * 24 / 384, 15.0, 15.0]))


def test_gemm_tile_and_splitk_selection():
    from rel_pose_amd import ops
    # mirrors of the tile choice in csrc/gemm.hip, per operand precision (0 = exact fp32 MFMA, the default; 3 = split-bf16)
    assert ops.GEMM_PRECISION == 0
    assert ops.gemm_instance(73728, 576, 0, 0) == (0, 0, 2, 1)
    assert ops.gemm_instance(73728, 768, 0, 1) == (0, 1, 1, 3)
    assert ops.gemm_instance(73728, 768, 0, 1, reads_mn=True) == (0, 1, 2, 1)
    assert ops.gemm_instance(576, 192, 1, 1) == (1, 1, 1, 3)
    assert ops.gemm_tile(73728, 768, 0, 1, precision=3) == (2, 1)
    assert ops.gemm_tile(576, 192, 1, 1, precision=3) == (1, 3)
    assert ops.gemm_instance(64, 512, 0, 0) == (0, 0, 1, 2)
    assert ops.gemm_tile(576, 64, 0, 1) == (2, 1)              # batched dQ = dS K: HBM-bound on dS, the ragged fifth panel is free
    with pytest.raises(ValueError):
        ops.set_gemm_precision(2)
    assert ops.pick_split_k(73728, 768, 192) == 1               # plenty of tiles
    sk = ops.pick_split_k(576, 192, 73728, 1, 1)                # weight gradient: 9 tiles, 2304 k-tiles
    assert 64 <= sk <= 128
    assert ops.pick_split_k(64, 512, 26880) > 32                # regressor layer 0 at batch 64
    assert ops.pick_split_k(64, 26880, 512) == 1                # its input gradient: 2 splits would run on 2 of the 8 XCDs only
    assert all(ops.pick_split_k(m, n, k, a, b) in (1,) + tuple(range(8, 129)) for (m, n, k, a, b) in
               ((6912, 192, 576, 0, 1), (64, 26880, 512, 0, 1), (192, 224, 8960, 1, 1), (512, 512, 64, 1, 1), (840, 192, 224, 0, 0)))


def test_se3_algebra_and_loss():
    from scipy.spatial.transform import Rotation as R
    from rel_pose_amd.se3 import SE3
    from rel_pose_amd.losses import geodesic_loss
    rng = np.random.default_rng(0)
    q = R.random(6, random_state=1)
    t = rng.normal(size=(6, 3))
    G = SE3(torch.tensor(np.concatenate([t, q.as_quat()], -1)).view(3, 2, 7))
    I = (G * G.inv()).data
    assert torch.allclose(I[..., :3], torch.zeros(3, 2, 3, dtype=I.dtype), atol=1e-12)
    assert torch.allclose(I[..., 3:].abs(), torch.tensor([0, 0, 0, 1.0], dtype=I.dtype).expand(3, 2, 4), atol=1e-12)
    # composition matches 4x4 matrices
    def mat(d):
        M = np.tile(np.eye(4), (d.shape[0], 1, 1))
        M[:, :3, :3] = R.from_quat(d[:, 3:]).as_matrix()
        M[:, :3, 3] = d[:, :3]
        return M
    a, b = G.data[:, 0].numpy(), G.data[:, 1].numpy()
    ab = (G[:, 0] * G[:, 1]).data.numpy()
    assert np.allclose(mat(ab), mat(a) @ mat(b), atol=1e-12)
    # log: rotation part = rotation vector, translation part = V^-1 t (check via scipy matrix log)
    from scipy.linalg import logm
    lg = G[:, 0].log().numpy()
    for i in range(3):
        L = np.real(logm(mat(a[i:i + 1])[0]))
        assert np.allclose(lg[i, 3:], [L[2, 1], L[0, 2], L[1, 0]], atol=1e-9)
        assert np.allclose(lg[i, :3], L[:3, 3], atol=1e-9)
    # loss is zero at the ground truth and differentiable
    Ps = SE3(G.data.float())
    est = G.data.float().clone().requires_grad_(True)
    ltr, lrot, metrics = geodesic_loss(Ps, [SE3(est)])
    assert float(ltr) < 1e-5 and float(lrot) < 1e-3 and set(metrics) == {"train_geo_loss_tr", "train_geo_loss_rot"}
    est2 = (G.data.float() + 0.1).requires_grad_(True)
    ltr, lrot, _ = geodesic_loss(Ps, [SE3(est2)])
    (ltr + lrot).backward()
    assert torch.isfinite(est2.grad).all() and float(ltr) > 0


def test_lazy_metrics_hand_plain_floats_to_every_dict_consumer():
    """reference train.py:168 does `metrics.update(geo_metrics)` and gives the dict to its Logger: CPython's dict fast paths must
    not copy the pending device tensors out of the subclass."""
    from rel_pose_amd.losses import LazyMetrics
    mk = lambda: LazyMetrics({"a": torch.tensor(1.5), "b": torch.tensor(2.0)})
    d = {}
    d.update(mk())
    assert type(d["a"]) is float and d == {"a": 1.5, "b": 2.0}
    assert type(dict(mk())["b"]) is float and type({**mk()}["a"]) is float
    assert type((mk() | {"c": 1})["a"]) is float and type(({"c": 1} | mk())["a"]) is float
    assert [type(v) for v in mk().values()] == [float, float] and sorted(mk()) == ["a", "b"]
    import pickle
    assert pickle.loads(pickle.dumps(mk())) == {"a": 1.5, "b": 2.0}


def test_splitk_batch_state_is_per_thread_and_nest_safe():
    """a re-entrant backward (a block opened inside a block) must not raise nor share the outer arena -- the inner block
    simply does not defer -- and a second device's backward thread has its own state."""
    import threading
    from rel_pose_amd import ops
    with ops.splitk_batch():
        outer = ops._sk_batch()
        assert outer is not None or not ops.SPLITK_BATCHING
        with ops.splitk_batch():
            assert ops._sk_batch() is None                   # inner: immediate reduces
            with ops.splitk_batch():
                assert ops._sk_batch() is None
        assert ops._sk_batch() is outer                      # the outer block resumes deferring
        seen = []
        t = threading.Thread(target=lambda: seen.append(ops._sk_batch()))
        t.start(); t.join()
        assert seen == [None]                                # other threads do not see this thread's block
    assert ops._sk_batch() is None and ops._TLS.depth == 0


def test_identity_like_and_indexing():
    from rel_pose_amd.se3 import SE3
    P = SE3(torch.randn(4, 2, 7))
    I = SE3.IdentityLike(P)
    assert I.data.shape == (4, 2, 7) and torch.equal(I.data[..., 6], torch.ones(4, 2)) and float(I.data[..., :6].abs().sum()) == 0
    assert I[:, :1].data.shape == (4, 1, 7) and I[0][1].data.shape == (7,)


def test_miopen_user_db_is_private_per_rank(tmp_path, monkeypatch):
    """rel_pose_amd/_env.py: the shipped MIOpen find-db is copied to a per-rank directory and selected through
    MIOPEN_USER_DB_PATH; a value set by the user wins."""
    import importlib
    import rel_pose_amd._env as env
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    import tempfile
    tempfile.tempdir = None
    monkeypatch.delenv("MIOPEN_USER_DB_PATH", raising=False)
    monkeypatch.setenv("LOCAL_RANK", "3")
    importlib.reload(env)
    p = os.environ["MIOPEN_USER_DB_PATH"]
    assert p.startswith(str(tmp_path)) and p.endswith("_3")
    assert sorted(os.listdir(p)) == sorted(f for f in os.listdir(env.MIOPEN_DB) if f.endswith(".txt")) and len(os.listdir(p)) == 3
    monkeypatch.setenv("MIOPEN_USER_DB_PATH", "/somewhere/else")
    importlib.reload(env)
    assert os.environ["MIOPEN_USER_DB_PATH"] == "/somewhere/else"
    tempfile.tempdir = None


def test_subepoch_schedule_follows_the_reference():
    """ten training sub-epochs then one validation pass; InteriorNet / StreetLearn have no validation split
    (reference train.py:204-208)"""
    import train
    seq, s = [], 0
    for _ in range(23):
        seq.append(s)
        s = train.next_subepoch(s, "matterport")
    assert seq == list(range(11)) + list(range(11)) + [0]
    for ds in ("interiornet", "streetlearn"):
        seq, s = [], 0
        for _ in range(21):
            seq.append(s)
            s = train.next_subepoch(s, ds)
        assert seq == list(range(10)) + list(range(10)) + [0] and 10 not in seq
    assert train.next_subepoch(3, "synthetic") == 0


class _NotATensor:
    """stands in for the bound method torch-1.8's OneCycleLR pickles into its state_dict"""

    def anneal(self, a, b, pct):
        return a + (b - a) * pct


def test_reference_style_checkpoint_loads(tmp_path):
    """the reference's checkpoints hold {'model','optimizer','scheduler'} and the scheduler state contains a bound method
    (anneal_func) that torch.load(weights_only=True) rejects; train.py / demo.py / test_matterport.py load them fully"""
    import torch, train
    obj = _NotATensor()
    ck = {"model": {"module.w": torch.ones(2)}, "optimizer": {"state": {}}, "scheduler": {"anneal_func": obj.anneal, "last_epoch": 7}}
    p = str(tmp_path / "ref_style.pth")
    torch.save(ck, p)
    with pytest.raises(Exception):
        torch.load(p, weights_only=True)
    got = train.load_checkpoint(p, map_location="cpu")
    assert got["scheduler"]["last_epoch"] == 7 and torch.equal(got["model"]["module.w"], torch.ones(2))
    assert got["scheduler"]["anneal_func"](0.0, 2.0, 0.5) == 1.0


def test_pretrained_trunk_loads_from_a_local_state_dict(tmp_path, monkeypatch):
    """resnet18(pretrained=True) takes torchvision-keyed weights from $RELPOSE_RESNET18_WEIGHTS and says whether it did"""
    import torch
    from rel_pose_amd.modules import resnet
    monkeypatch.delenv(resnet.WEIGHTS_ENV, raising=False)
    net = resnet.resnet18(pretrained=True)
    assert net.pretrained_loaded is False
    sd = {k: torch.full_like(v, 0.25) for k, v in net.state_dict().items()}
    sd["fc.weight"] = torch.zeros(1000, 512)          # torchvision's classifier head is ignored
    p = str(tmp_path / "resnet18.pth")
    torch.save(sd, p)
    monkeypatch.setenv(resnet.WEIGHTS_ENV, p)
    net2 = resnet.resnet18(pretrained=True)
    assert net2.pretrained_loaded and float(net2.conv1.weight.flatten()[0]) == 0.25
    torch.save({"not_a_resnet": torch.zeros(1)}, p)
    with pytest.raises(RuntimeError):
        resnet.resnet18(pretrained=True)


def test_tangent_space_gradient_matches_finite_differences_of_left_perturbation():
    """losses.GRADIENT_CONVENTION = "tangent": the gradient reaching the predicted pose is [dL/dtau, dL/dphi, 0] for
    G <- Exp(xi) * G.  Checked against central finite differences of the fp64 loss along Exp(xi) (scipy.linalg.expm on the
    4x4 twist), for both slots of random pose pairs.  (That lietorch uses this convention is recalled, not verifiable here.)"""
    import scipy.linalg
    import torch
    from rel_pose_amd import losses, se3

    def to_mat(d):
        x, y, z, w = d[3:]
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        T = np.eye(4); T[:3, :3] = R; T[:3, 3] = d[:3]
        return T

    def from_mat(T, q_like):
        from scipy.spatial.transform import Rotation
        q = Rotation.from_matrix(T[:3, :3]).as_quat()          # xyzw
        if np.dot(q, q_like) < 0:
            q = -q
        return np.concatenate([T[:3, 3], q])

    def twist(xi):
        tau, phi = xi[:3], xi[3:]
        M = np.zeros((4, 4))
        M[:3, :3] = np.array([[0, -phi[2], phi[1]], [phi[2], 0, -phi[0]], [-phi[1], phi[0], 0]])
        M[:3, 3] = tau
        return scipy.linalg.expm(M)

    g = torch.Generator().manual_seed(5)
    B = 3
    def rand_pose():
        q = torch.randn(B, 2, 4, generator=g, dtype=torch.float64)
        q = q / q.norm(dim=-1, keepdim=True)
        return torch.cat([torch.randn(B, 2, 3, generator=g, dtype=torch.float64), q], -1)
    P, G = rand_pose(), rand_pose()

    def loss_np(Gd):
        ltr, lrot = losses.geodesic_loss_tensors_torch(se3.SE3(P), [se3.SE3(torch.as_tensor(Gd))])
        return float(10.0 * ltr + 3.0 * lrot)

    Gt = G.clone().requires_grad_(True)
    ltr, lrot = losses.geodesic_loss_tensors_torch(se3.SE3(P), [se3.with_tangent_gradient(se3.SE3(Gt))])
    (10.0 * ltr + 3.0 * lrot).backward()
    got = Gt.grad.numpy()
    assert np.all(got[..., 6] == 0.0)
    eps = 1e-6
    for b in range(B):
        for s in range(2):
            for k in range(6):
                xi = np.zeros(6); xi[k] = eps
                Gp, Gm = G.numpy().copy(), G.numpy().copy()
                Gp[b, s] = from_mat(twist(xi) @ to_mat(G[b, s].numpy()), G[b, s, 3:].numpy())
                Gm[b, s] = from_mat(twist(-xi) @ to_mat(G[b, s].numpy()), G[b, s, 3:].numpy())
                fd = (loss_np(Gp) - loss_np(Gm)) / (2 * eps)
                assert abs(fd - got[b, s, k]) < 1e-5 * max(1.0, abs(fd)), (b, s, k, fd, got[b, s, k])
    # the switch: default euclidean; "tangent" routes through with_tangent_gradient
    assert losses.GRADIENT_CONVENTION == "euclidean"


def test_branch_free_gelu_constants_are_accurate():
    """common.h's gelu_fast / gelu_grad_fast (used by the fused MLP and row-resident kernels instead of libm's erff): the fp32 Horner
    evaluation with the constants AS WRITTEN IN THE HEADER stays within 5e-7 of the exact-erf GELU (<= 1 ulp of the result near
    |x| ~ 4.5) and its derivative within 5e-7, over [-9, 9].  tools/fit_gelu.py is the derivation."""
    from scipy.special import erf
    src = open(os.path.join(ROOT, "rel_pose_amd", "csrc", "common.h")).read()
    body = src[src.index("RP_DEV float gelu_fast(float x)"):src.index("RP_DEV float gelu_grad_fast")]
    c = [float(v) for v in re.findall(r"(-?\d\.\d+e[+-]\d+)f", body)]
    assert len(c) == 8                                               # c7 (initial p) then c6 .. c0 in Horner order
    x = np.linspace(-9, 9, 360001).astype(np.float32)
    t = np.minimum(np.abs(x), np.float32(5.7))
    p = np.full_like(t, np.float32(c[0]))
    for k in c[1:]:
        p = p * t + np.float32(k)
    e = np.exp2((-(t * p) - np.float32(1)).astype(np.float32)).astype(np.float32)
    cdf = np.where(x < 0, e, np.float32(1) - e).astype(np.float32)
    xd = x.astype(np.float64)
    phi = 0.5 * (1 + erf(xd / np.sqrt(2)))
    assert np.abs((x * cdf).astype(np.float64) - xd * phi).max() < 5e-7
    pdf = (np.float32(0.39894228040143267794) * np.exp2((np.float32(-0.72134752044448170368) * x * x).astype(np.float32))).astype(np.float32)
    grad = (x * pdf + cdf).astype(np.float64)
    assert np.abs(grad - (phi + xd * np.exp(-0.5 * xd * xd) / np.sqrt(2 * np.pi))).max() < 5e-7


def test_bf16_weight_copies_of_the_bf16_configuration():
    """Host side of the bf16 configuration's row-resident / fused-MLP kernels: ops.bf16_weight keeps a round-to-nearest-even bf16 copy of a
    weight until the weight is modified; ops._chunk_permuted_bf16 additionally stores the 768 hidden units of every 32-chunk in the order
    include/relpose_hip.h documents at rp_mlp_fused_bwd (position 8q+e <- unit 4q+e for e < 4, 16+4q+e-4 otherwise)."""
    from rel_pose_amd import ops
    w = torch.nn.Parameter(torch.randn(768, 192))
    a = ops.bf16_weight(w)
    assert a.dtype == torch.bfloat16 and torch.equal(a, w.detach().to(torch.bfloat16)) and ops.bf16_weight(w) is a
    with torch.no_grad():
        w.mul_(1.5)                                                       # an optimizer step: the copy is refreshed
    b = ops.bf16_weight(w)
    assert b is not a and torch.equal(b, w.detach().to(torch.bfloat16))
    perm = ops._mlp_unit_perm(torch.device("cpu"))
    assert perm.shape == (768,) and sorted(perm.tolist()) == list(range(768))                      # a permutation ...
    assert all(int(perm[i]) // 32 == i // 32 for i in range(768))                                  # ... inside every 32-chunk
    for pos in range(32):
        q, e = pos // 8, pos % 8
        assert int(perm[64 + pos]) == 64 + (4 * q + e if e < 4 else 16 + 4 * q + e - 4)
    w2 = torch.nn.Parameter(torch.randn(192, 768))
    # chunk-major storage (io_bf16 bit 4 of rp_mlp_fused_fwd / _bwd): [24 chunks][192][32], every staged tile contiguous
    cm = lambda t: t.reshape(192, 24, 32).permute(1, 0, 2)                                          # noqa: E731
    assert ops.MLP_W2_CHUNK_MAJOR
    p2 = ops._chunk_permuted_bf16(w2)
    assert p2.shape == (24, 192, 32) and p2.is_contiguous() and torch.equal(p2, cm(w2.detach()[:, perm].to(torch.bfloat16)))
    assert ops._chunk_permuted_bf16(w2) is p2
    w1 = torch.nn.Parameter(torch.randn(768, 192))
    p1 = ops._chunk_permuted_bf16(w1, transpose=True)
    assert p1.shape == (24, 192, 32) and torch.equal(p1, cm(w1.detach().t()[:, perm].to(torch.bfloat16)))
    with torch.no_grad():
        w1.add_(1.0)
    assert ops._chunk_permuted_bf16(w1, transpose=True) is not p1


def test_padded_weight_cache_follows_in_place_updates():
    """ops._padded caches the alignment pads of the CrossBlock / regressor weights between forwards; an optimizer step, load_state_dict
    or any other in-place write (Tensor._version) and a re-pointed .data must invalidate it."""
    from rel_pose_amd import ops
    w = torch.nn.Parameter(torch.arange(12.0).view(3, 4))
    a = ops._padded(w, (0, 2))
    assert a.shape == (3, 6) and torch.equal(a[:, :4], w.detach()) and float(a[:, 4:].abs().max()) == 0.0
    assert ops._padded(w, (0, 2)) is a                                   # unchanged parameter: cached
    assert ops._padded(w, (0, 0, 0, 1)).shape == (4, 4)                  # another pad of the same tensor: its own entry
    opt = torch.optim.SGD([w], lr=1.0)
    w.grad = torch.ones_like(w)
    opt.step()                                                           # in-place update
    b = ops._padded(w, (0, 2))
    assert b is not a and torch.equal(b[:, :4], w.detach())
    w.data = torch.zeros(3, 4)                                           # re-pointed storage
    assert float(ops._padded(w, (0, 2)).abs().max()) == 0.0
    with torch.no_grad():
        w.copy_(torch.full((3, 4), 2.0))                                 # load_state_dict-style copy
    assert float(ops._padded(w, (0, 2))[:, :4].min()) == 2.0
    # a writer that does not bump _version (a HIP-graph replay of the optimizer step): explicit invalidation
    c = ops._padded(w, (0, 2))
    w.data.view(-1).as_strided((12,), (1,)).detach().numpy()[:] = 5.0   # write through numpy: no version bump
    assert ops._padded(w, (0, 2)) is c                                  # (stale by construction)
    ops.invalidate_pad_cache()
    d = ops._padded(w, (0, 2))
    assert d is not c and float(d[:, :4].min()) == 5.0
    # the cache lives on the tensor: a new parameter (even one that reuses id / storage of a deleted one) starts empty
    w2 = torch.nn.Parameter(torch.ones(3, 4))
    assert not hasattr(w2, "_rp_pads") and float(ops._padded(w2, (0, 2))[:, :4].max()) == 1.0


def test_clip_grad_norm_equals_torch():
    """parallel.clip_grad_norm_ = torch.nn.utils.clip_grad_norm_ (reference train.py:158), bit for bit, clipping and not clipping."""
    from rel_pose_amd import parallel
    for scale in (10.0, 1e-3):
        torch.manual_seed(1)
        ps = [torch.nn.Parameter(torch.randn(*s)) for s in ((7, 5), (3,), (2, 3, 4), (1,))]
        qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
        for p, q in zip(ps, qs):
            p.grad = torch.randn_like(p) * scale
            q.grad = p.grad.clone()
        qs.append(torch.nn.Parameter(torch.zeros(2)))                     # a parameter without a gradient
        ta = torch.nn.utils.clip_grad_norm_(ps, 2.5)
        tb = parallel.clip_grad_norm_(qs, 2.5)
        assert torch.equal(ta, tb)
        for p, q in zip(ps, qs):
            assert torch.equal(p.grad, q.grad)


def test_essential_decode_oracle_round_trip():
    """oracle/svd3x3_oracle.decode_essential (the LAPACK pin of rp_pose_from_essential): pose -> E -> four candidates + cheirality ->
    the input rotation and the direction of t, on points in front of both cameras."""
    from oracle import svd3x3_oracle as SO
    rng = np.random.default_rng(5)
    n, P = 12, 10
    pose = np.zeros((n, 7))
    pose[:, :3] = rng.normal(size=(n, 3))
    ax = rng.normal(size=(n, 3))
    ax /= np.linalg.norm(ax, axis=1, keepdims=True)
    half = rng.uniform(-1.0, 1.0, size=(n, 1))          # rotation angle within +-2 rad: points in front of both cameras always exist
    pose[:, 3:6], pose[:, 6:] = ax * np.sin(half), np.cos(half)
    R, t = SO.rotation_from_quat(pose[:, 3:]), pose[:, :3]
    X1 = np.empty((n, P, 3))
    for i in range(n):
        got, tries = 0, 0
        while got < P:
            tries += 1
            assert tries < 500
            c = np.concatenate([rng.normal(size=(64, 2)) * 2, rng.uniform(1, 7, size=(64, 1))], 1)
            ok = c[(c @ R[i].T + t[i])[:, 2] > 0.5]
            k = min(P - got, len(ok))
            X1[i, got:got + k] = ok[:k]
            got += k
    X2 = np.einsum("nij,npj->npi", R, X1) + t[:, None]
    Ro, to, co = SO.decode_essential(SO.essential_from_pose(pose), X1[..., :2] / X1[..., 2:], X2[..., :2] / X2[..., 2:])
    assert (co == P).all()
    assert np.abs(Ro - R).max() < 1e-9 and np.abs(to - t / np.linalg.norm(t, axis=1, keepdims=True)).max() < 1e-9


def test_miopen_db_check_reports_a_foreign_miopen(tmp_path, monkeypatch):
    """rel_pose_amd._env.check_db: a db file under a name the shipped set does not contain (= another MIOpen build looked for its own
    file) is reported; the shipped names alone are not."""
    from rel_pose_amd import _env
    d = tmp_path / "udb"
    d.mkdir()
    for f in os.listdir(_env.MIOPEN_DB):
        if f.endswith(".txt"):
            (d / f).write_text("")
    monkeypatch.setenv("MIOPEN_USER_DB_PATH", str(d))
    base = [m for m in _env.check_db(warn=False) if "opened db files" in m]
    assert base == []
    (d / "gfx950100.HIP.9_9_9_deadbeef.ufdb.txt").write_text("")
    found = [m for m in _env.check_db(warn=False) if "opened db files" in m]
    assert len(found) == 1 and "9_9_9_deadbeef" in found[0]


def test_batched_parameter_cast_passes_gradients_through():
    """ops._CastParamsFn (the bf16 configuration's one-launch cast of all convolution weights): bf16 copies forward, fp32 copies of the
    bf16 gradients backward, None for outputs that received no gradient; ops.conv_params_bf16 is inert at CNN precision 0 / on CPU."""
    from rel_pose_amd import ops
    ps = [torch.nn.Parameter(torch.randn(4, 3, 3, 3)), torch.nn.Parameter(torch.randn(4)), torch.nn.Parameter(torch.randn(2, 2))]
    outs = ops._CastParamsFn.apply(*ps)
    assert all(o.dtype == torch.bfloat16 and torch.equal(o, p.detach().to(torch.bfloat16)) for o, p in zip(outs, ps))
    g0, g1 = torch.randn(4, 3, 3, 3), torch.randn(4)
    (outs[0].float() * g0).sum().backward(retain_graph=True)
    (outs[1].float() * g1).sum().backward()
    assert ps[0].grad.dtype == torch.float32 and torch.equal(ps[0].grad, g0.to(torch.bfloat16).float())
    assert torch.equal(ps[1].grad, g1.to(torch.bfloat16).float()) and ps[2].grad is None
    conv = torch.nn.Conv2d(3, 4, 3)
    with ops.conv_params_bf16(conv):
        assert getattr(conv, "_rp_bf16", None) is None          # precision 0: nothing is prepared, conv2d runs the module itself
        y = ops.conv2d(conv, torch.randn(1, 3, 8, 8))
    assert y.dtype == torch.float32 and y.shape == (1, 4, 6, 6)
