"""Pin the CPU oracle (oracle/relpose_oracle.py) against outputs of the real reference
(tests/golden/reference_outputs.npz, produced by tests/golden/make_fixtures.py)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import relpose_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


def intr24(dtype=torch.float32):
    a = torch.tensor([[32.373, 25.898, 12.0, 12.0], [18.0, 21.0, 12.0, 9.0]], dtype=dtype)
    return a[:, None, :].repeat(1, 2, 1).contiguous()


def rel(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max())


@pytest.fixture(scope="module")
def states():
    shapes = dict(O.vit_param_shapes())
    shapes.update(O.cnn_param_shapes())
    return O.make_state(shapes, torch.float32), O.make_state(shapes, torch.float64)


def test_state_dict_keys_match_reference():
    with open(os.path.join(HERE, "golden", "state_dict_keys.json")) as f:
        ref = json.load(f)
    mine = dict(O.vit_param_shapes())
    mine.update(O.cnn_param_shapes())
    assert set(mine) == set(ref)
    for k, shp in mine.items():
        assert list(shp) == ref[k], k
    assert len(ref) == 227
    nparam = sum(int(np.prod(v)) for k, v in ref.items()
                 if "num_batches" not in k and "running" not in k and not k.startswith("extractor_final_conv.downsample.1"))
    assert nparam == 29751950       # SURVEY.md 8b (extractor downsample.1 aliases norm3)


def test_vit_stack_fp32_and_fp64(golden, states):
    sd32, sd64 = states
    with torch.no_grad():
        x = O.synthetic_tokens(4) + sd32["fusion_transformer.pos_embed"]
        x = O.block(sd32, "fusion_transformer.blocks.0.", x)
        assert rel(x.reshape(-1)[::37], golden["vit_block0_sub_f32"]) < 2e-6
        f32 = O.vit_features(sd32, O.synthetic_tokens(4), intr24())
        f64 = O.vit_features(sd64, O.synthetic_tokens(4, dtype=torch.float64), intr24(torch.float64))
    assert rel(f64, golden["vit_feat_f64"]) < 1e-7          # pos-enc goes through fp32 in both
    assert rel(f32, golden["vit_feat_f32"]) < 2e-4          # fp32 op-order noise of the reference itself
    # how far the reference's own fp32 is from its fp64: calibrates the GPU tolerances
    assert rel(golden["vit_feat_f32"], golden["vit_feat_f64"]) < 1e-3


def test_pose_from_tokens(golden, states):
    sd32, sd64 = states
    Gs = torch.tensor([0, 0, 0, 0, 0, 0, 1.0]).repeat(2, 2, 1)
    with torch.no_grad():
        p32 = O.vit_ess_from_tokens(sd32, O.synthetic_tokens(4), Gs, intr24())
        p64 = O.vit_ess_from_tokens(sd64, O.synthetic_tokens(4, dtype=torch.float64), Gs.double(), intr24(torch.float64))
    assert rel(p64, golden["pose_from_tokens_f64"]) < 1e-7
    t, q, ang = O.pose_errors(p32, torch.as_tensor(golden["pose_from_tokens_f32"]))
    assert max(t, q) < 1e-4 and ang < 1e-4
    assert torch.equal(p32[:, 0], Gs[:, 0])                 # slot 0 is the identity passthrough, bit-exact


def test_positional_encodings(golden):
    i = intr24()
    assert rel(O.positional_encodings_loop(2, i.clone()), golden["posenc_intr_f32"]) == 0.0
    assert np.array_equal(O.positional_encodings_loop(2, None).numpy(), golden["posenc_none_f32"])
    assert np.array_equal(O.positional_encodings(2, None).numpy(), golden["posenc_none_f32"])
    assert rel(O.positional_encodings(2, i.clone()), golden["posenc_intr_f32"]) < 2e-7     # closed form
    mp = torch.tensor([[517.97, 517.97, 320, 240]] * 2)[None].clone()
    mp[:, :, [0, 2]] *= 24 / 512
    mp[:, :, [1, 3]] *= 24 / 384
    assert rel(O.positional_encodings(1, mp), golden["posenc_matterport_f32"]) < 2e-7


def test_gradients_fp64(golden, states):
    _, sd64 = states
    with open(os.path.join(HERE, "golden", "grad_param_names.json")) as f:
        names = json.load(f)
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd64.items()}
    tok = O.synthetic_tokens(4, dtype=torch.float64).requires_grad_(True)
    f = O.vit_features(sd, tok, intr24(torch.float64))
    cot = O.closed_form(tuple(f.shape), 991, 1.0, dtype=torch.float64)
    (f * cot).sum().backward()
    assert rel(tok.grad.reshape(-1)[::37], golden["grad_tokens_sub_f64"]) < 1e-7
    ref = golden["grad_param_summaries_f64"]
    for i, n in enumerate(names):
        g = sd[n].grad.double().reshape(-1)
        got = np.concatenate([[float(g.sum()), float(g.abs().sum()), float((g * g).sum())], g[:16].numpy(), np.zeros(max(0, 16 - g.numel()))])
        scale = max(np.abs(ref[i]).max(), 1e-30)
        assert np.abs(got - ref[i]).max() / scale < 1e-7, n


@pytest.mark.parametrize("tag,B,H,W,key", [("sq", 2, 384, 384, 7), ("rect", 1, 256, 320, 8)])
def test_full_forward(golden, states, tag, B, H, W, key):
    sd32, sd64 = states
    imgs = O.synthetic_images(B, H, W, key=key)
    intr = torch.tensor([[0.9 * W, 0.8 * W, W / 2.0, H / 2.0]]).repeat(B, 2, 1).contiguous()
    Gs = torch.tensor([0, 0, 0, 0, 0, 0, 1.0]).repeat(B, 2, 1)
    with torch.no_grad():
        pose, tokens = O.vit_ess_forward(sd32, imgs.clone(), Gs, intr)
        i64 = torch.tensor([[0.9 * W, 0.8 * W, W / 2.0, H / 2.0]]).repeat(B, 2, 1).double().contiguous()
        pose64, _ = O.vit_ess_forward(sd64, imgs.double(), Gs.double(), i64)
    assert np.array_equal(intr.numpy(), golden["full_%s_intr_after" % tag])        # in-place mutation, bit-exact
    assert rel(tokens.reshape(-1)[::101], golden["full_%s_tokens_sub_f32" % tag]) < 1e-4
    assert rel(pose64, golden["full_%s_pose_f64" % tag]) < 1e-7
    t, q, ang = O.pose_errors(pose, torch.as_tensor(golden["full_%s_pose_f32" % tag]))
    assert max(t, q) < 2e-4


def test_full_forward_train_mode(golden, states):
    sd32, _ = states
    imgs = O.synthetic_images(2, 384, 384, key=7)
    intr = torch.tensor([[0.9 * 384, 0.8 * 384, 192.0, 192.0]]).repeat(2, 2, 1).contiguous()
    Gs = torch.tensor([0, 0, 0, 0, 0, 0, 1.0]).repeat(2, 2, 1)
    with torch.no_grad():
        pose, _ = O.vit_ess_forward(sd32, imgs, Gs, intr, train=True)
    t, q, ang = O.pose_errors(pose, torch.as_tensor(golden["full_sq_pose_trainmode_f32"]))
    assert max(t, q) < 5e-4


def test_index_ops_bit_exact(golden):
    for n_in in (256, 320, 384, 480, 512, 640):
        assert np.array_equal(O.nearest_src_index(224, n_in), golden["nearest224_from_%d" % n_in])
    x = torch.arange(2 * 192 * 576, dtype=torch.float32).reshape(2, 192, 24, 24)
    tk = O.tokens_from_cnn(x)
    assert np.array_equal(tk[1, ::97, ::31].numpy().astype(np.int64), golden["token_layout_probe"])
    # token n = row*24+col of the CNN map, channel-last
    assert float(tk[1, 5 * 24 + 7, 33]) == float(x[1, 33, 5, 7])


VARIANTS = {"l1": dict(l1_pos_encoding=True), "single": dict(use_single_softmax=True), "cross": dict(cross_features=True),
            "all3": dict(l1_pos_encoding=True, use_single_softmax=True, cross_features=True)}


@pytest.mark.parametrize("tag", list(VARIANTS))
def test_ablation_variants_fp64(golden, states, tag):
    """the three ablation flags that run in the reference (no_pos_encoding crashes there)."""
    _, sd64 = states
    tok = O.synthetic_tokens(4, dtype=torch.float64).requires_grad_(True)
    f = O.vit_features(sd64, tok, intr24(torch.float64), **VARIANTS[tag])
    assert rel(f.reshape(-1)[::5], golden["variant_%s_feat_sub_f64" % tag]) < 1e-7
    cot = O.closed_form(tuple(f.shape), 991, 1.0, dtype=torch.float64)
    (f * cot).sum().backward()
    assert rel(tok.grad.reshape(-1)[::37], golden["variant_%s_grad_tokens_sub_f64" % tag]) < 1e-7


# ---- --noess ablation (vision_transformer.py:239-262,297-304; src/model.py:73-82,183-188) ----------------------------
@pytest.fixture(scope="module")
def golden_noess():
    return np.load(os.path.join(HERE, "golden", "reference_outputs_noess.npz"))


@pytest.fixture(scope="module")
def states_noess():
    shapes = dict(O.vit_param_shapes(noess=True))
    shapes.update(O.cnn_param_shapes())
    return O.make_state(shapes, torch.float32), O.make_state(shapes, torch.float64)


def test_noess_state_dict_keys_match_reference():
    with open(os.path.join(HERE, "golden", "state_dict_keys_noess.json")) as f:
        ref = json.load(f)
    mine = dict(O.vit_param_shapes(noess=True))
    mine.update(O.cnn_param_shapes())
    assert set(mine) == set(ref)
    for k, shp in mine.items():
        assert list(shp) == ref[k], k


def _noess_grad_sums(sd):
    def summ(g):
        g = g.double().reshape(-1)
        return np.concatenate([[float(g.sum()), float(g.abs().sum()), float((g * g).sum())], g[:16].numpy()])
    keys = ["fusion_transformer.blocks.5.cross_attn.qkv.weight", "fusion_transformer.blocks.5.cross_attn.proj.weight",
            "pool_attn.0.weight", "pool_attn.4.weight", "pose_regressor.0.weight"]
    return np.stack([summ(sd[k].grad) for k in keys])


@pytest.mark.parametrize("train", [False, True])
def test_noess_fp64_forward_and_gradients(golden_noess, states_noess, train):
    _, sd64 = states_noess
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd64.items()}
    tag = "train" if train else "eval"
    tok = O.synthetic_tokens(4, dtype=torch.float64).requires_grad_(True)
    Gs = torch.tensor([0, 0, 0, 0, 0, 0, 1.0], dtype=torch.float64).repeat(2, 2, 1)
    if not train:
        with torch.no_grad():
            f = O.vit_features(sd, tok, intr24(torch.float64), noess=True)
        assert rel(f.reshape(-1)[::23], golden_noess["noess_feat_sub_f64"]) < 1e-9
    pose = O.vit_ess_from_tokens(sd, tok, Gs, intr24(torch.float64), train=train, noess=True)
    assert rel(pose, golden_noess["noess_pose_from_tokens_%s_f64" % tag]) < 1e-9
    cot = O.closed_form((2, 7), 993, 1.0, dtype=torch.float64)
    (pose[:, 1] * cot).sum().backward()
    assert rel(tok.grad.reshape(-1)[::37], golden_noess["noess_grad_tokens_sub_%s_f64" % tag]) < 1e-7
    assert rel(_noess_grad_sums(sd), golden_noess["noess_grad_sums_%s_f64" % tag]) < 1e-7


def test_noess_fp32_and_full_model(golden_noess, states_noess):
    sd32, sd64 = states_noess
    Gs = torch.tensor([0, 0, 0, 0, 0, 0, 1.0]).repeat(2, 2, 1)
    with torch.no_grad():
        f = O.vit_features(sd32, O.synthetic_tokens(4), intr24(), noess=True)
        assert rel(f.reshape(-1)[::23], golden_noess["noess_feat_sub_f32"]) < 2e-4
        p = O.vit_ess_from_tokens(sd32, O.synthetic_tokens(4), Gs, intr24(), noess=True)
        assert rel(p, golden_noess["noess_pose_from_tokens_f32"]) < 2e-4
        imgs = O.synthetic_images(2, 384, 384, key=7)
        intr = torch.tensor([[0.9 * 384, 0.8 * 384, 192.0, 192.0]]).repeat(2, 2, 1).contiguous()
        pose64, _ = O.vit_ess_forward(sd64, imgs.double(), Gs.double(), intr.clone().double(), noess=True)
        assert rel(pose64, golden_noess["noess_full_sq_pose_f64"]) < 1e-7
        pose32, _ = O.vit_ess_forward(sd32, imgs, Gs, intr.clone(), noess=True)
        assert rel(pose32, golden_noess["noess_full_sq_pose_f32"]) < 1e-3


# ---- BASELINE configs[0]: demo.py plumbing, pinned to the reference's own demo.py run on its demo images ----
DEMO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _demo_inputs():
    import demo
    return demo, [os.path.join(DEMO, "demo", "matterport_%d.png" % i) for i in (1, 2)]


def test_demo_png_decode_and_tensor_assembly_bit_exact():
    """The stdlib PNG reader (cv2 is absent) decodes the reference's 640x480 RGBA demo images to exactly what cv2.imread
    gives (BGR, alpha dropped), and the [1,2,3,384,512] tensor demo.py assembles from them (demo.py:65-76, nearest resize)
    equals the one the reference's demo.py built, sample for sample."""
    from PIL import Image
    demo, paths = _demo_inputs()
    for pth in paths:
        im = Image.open(pth)
        assert im.mode == "RGBA" and im.size == (640, 480)
        want = np.asarray(im.convert("RGB"))[:, :, ::-1]
        assert np.array_equal(demo.read_png_bgr(pth), want)
    images = demo.load_pair(paths[0], paths[1], matterport=True)
    ref = np.load(os.path.join(DEMO, "reference_demo.npz"))
    assert tuple(images.shape) == (1, 2, 3, 384, 512)
    assert np.array_equal(images[0, :, :, ::29, ::31].numpy(), ref["demo_matterport_images_sub"])


def test_demo_pose_matches_reference_demo_run():
    """Oracle forward on the demo pair with the closed-form checkpoint + demo.py's post-processing (x5 depth scale, quaternion
    reorder [4,5,3,6]) reproduces the [7] vector printed by the reference's demo.py on the same inputs and weights."""
    demo, paths = _demo_inputs()
    ref = np.load(os.path.join(DEMO, "reference_demo.npz"))
    shapes = dict(O.vit_param_shapes())
    shapes.update(O.cnn_param_shapes())
    sd32 = O.make_state(shapes, torch.float32)
    images = demo.load_pair(paths[0], paths[1], matterport=True)
    intr = torch.tensor([[[517.97, 517.97, 320, 240]] * 2], dtype=torch.float32)
    Gs = torch.tensor([[[0, 0, 0, 0, 0, 0, 1.0]] * 2])
    with torch.no_grad():
        pose, _ = O.vit_ess_forward(sd32, images, Gs, intr)
    raw = pose[0, 1].numpy()
    assert np.abs(raw - ref["demo_matterport_raw7_f32"]).max() < 2e-4 * np.abs(ref["demo_matterport_raw7_f32"]).max()
    got = demo.postprocess(raw, True)
    assert np.abs(got - ref["demo_matterport_pred7_f32"]).max() < 2e-4 * np.abs(ref["demo_matterport_pred7_f32"]).max()
    r = ref["demo_matterport_raw7_f32"]
    assert np.array_equal(demo.postprocess(r, True), ref["demo_matterport_pred7_f32"])      # the index shuffle itself: exact
