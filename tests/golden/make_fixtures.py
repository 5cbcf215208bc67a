#!/usr/bin/env python3
"""Generate golden vectors by importing the REAL reference (build container only).

    python -B tests/golden/make_fixtures.py            # writes tests/golden/*.npz, *.json

The reference (/root/reference, read-only, Python) cannot travel to the GPU box, so its outputs on
closed-form (RNG-free) inputs and weights are committed here as data.  Harness-side shims, reference
files untouched:
  * torch.Tensor.cuda -> identity  (vision_transformer.py:209,211 and model.py:164 hard-code .cuda())
  * sys.modules['torchvision.models'].resnet18 -> this repo's torchvision-free trunk (same keys)
  * sys.modules['lietorch'].SE3 -> thin wrapper exposing .data / __getitem__ (all the model path uses)
Inputs/weights come from oracle.relpose_oracle.closed_form (integer hash), so tests regenerate them
bit-identically and only OUTPUTS are stored.
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("RELPOSE_REFERENCE", "/root/reference")

import numpy as np
import torch

torch.Tensor.cuda = lambda self, *a, **k: self          # shim 1

from oracle import relpose_oracle as O                   # noqa: E402  (closed-form generators only)
from rel_pose_amd.modules import resnet as _rn           # noqa: E402
import importlib.util as _ilu                            # noqa: E402
_spec = _ilu.spec_from_file_location("_eval_cases", os.path.join(ROOT, "tests", "_eval_cases.py"))
EC = _ilu.module_from_spec(_spec)                        # closed-form fake datasets / metric inputs shared with the tests
_spec.loader.exec_module(EC)

tv = types.ModuleType("torchvision")
tvm = types.ModuleType("torchvision.models")
tvm.resnet18 = _rn.resnet18                              # shim 2
tv.models = tvm
sys.modules["torchvision"] = tv
sys.modules["torchvision.models"] = tvm


class SE3:                                               # shim 3
    def __init__(self, data):
        self.data = data

    def __getitem__(self, idx):
        return SE3(self.data[idx])

    @staticmethod
    def IdentityLike(other):                             # test_streetlearn_interiornet.py:220
        d = torch.zeros_like(other.data)
        d[..., 6] = 1
        return SE3(d)


lt = types.ModuleType("lietorch")
lt.SE3 = SE3
sys.modules["lietorch"] = lt

# this repo ships drop-in aliases under the same package name `src` (a regular package, which would shadow the reference's
# namespace package): take the repo root off sys.path while the reference is imported, and check what was imported
sys.path = [p_ for p_ in sys.path if os.path.abspath(p_ or ".") not in (ROOT, HERE)]
sys.path.insert(0, REF)
for _m in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
    del sys.modules[_m]
from src.model import ViTEss                              # noqa: E402  (reference)
from src.modules import vision_transformer as RVT         # noqa: E402  (reference)
assert os.path.abspath(RVT.__file__).startswith(os.path.abspath(REF)), RVT.__file__
assert os.path.abspath(sys.modules["src.model"].__file__).startswith(os.path.abspath(REF))

torch.set_num_threads(8)
torch.manual_seed(0)


def ref_args(**kw):
    a = types.SimpleNamespace(noess="", pool_size=60, fc_hidden_size=512, fusion_transformer=True,
                              transformer_depth=6, cross_features=False, use_single_softmax=False,
                              no_pos_encoding=False, l1_pos_encoding=False)
    a.__dict__.update(kw)
    # reference does `'noess' in args` (argparse.Namespace supports `in`)
    return _NS(a.__dict__)


class _NS(types.SimpleNamespace):
    def __init__(self, d):
        super().__init__(**d)

    def __contains__(self, k):
        return k in self.__dict__


def subsample(t, step=37):
    return t.detach().reshape(-1)[::step].contiguous().numpy()


def summarize(g, n=16):
    g = g.detach().double().reshape(-1)
    return np.concatenate([[float(g.sum()), float(g.abs().sum()), float((g * g).sum())], g[:n].numpy(),
                           np.zeros(max(0, n - g.numel()))])


def intrinsics_24(dtype=torch.float32):
    # already on the 24x24 grid; equal across the two images of a pair (vision_transformer.py:117)
    a = torch.tensor([[32.373, 25.898, 12.0, 12.0], [18.0, 21.0, 12.0, 9.0]], dtype=dtype)
    return a[:, None, :].repeat(1, 2, 1).contiguous()


def main():
    out = {}
    model = ViTEss(ref_args()).eval()
    ref_keys = {k: list(v.shape) for k, v in model.state_dict().items()}
    with open(os.path.join(HERE, "state_dict_keys.json"), "w") as f:
        json.dump(ref_keys, f, indent=0, sort_keys=True)

    shapes = dict(O.vit_param_shapes())
    shapes.update(O.cnn_param_shapes())
    sd32 = O.make_state(shapes, torch.float32)
    # reference also carries resnet.layer3/4 (unused): leave at their constructed values
    missing, unexpected = model.load_state_dict(sd32, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith("resnet.layer3") or k.startswith("resnet.layer4") for k in missing), missing
    ft = model.fusion_transformer

    # ---- A: ViT stack on tokens, fp32 and fp64 -------------------------------------------------
    B = 2
    tok32 = O.synthetic_tokens(2 * B)
    intr = intrinsics_24()

    def run_stack(m, tok, intr_):
        x = tok + m.pos_embed
        inter = []
        for l in range(6):
            x = m.blocks[l](x, intrinsics=intr_)
            inter.append(x)
        return m.norm(x), inter

    with torch.no_grad():
        f32, inter32 = run_stack(ft, tok32, intr.clone())
    out["vit_feat_f32"] = f32.numpy()
    out["vit_block0_sub_f32"] = subsample(inter32[0])
    out["vit_block4_sub_f32"] = subsample(inter32[4])
    out["vit_block5_f32"] = inter32[5].numpy()

    model64 = ViTEss(ref_args()).eval()
    sd64 = O.make_state(shapes, torch.float64)
    model64.load_state_dict(sd32, strict=False)
    model64 = model64.double()
    model64.load_state_dict({k: v for k, v in sd64.items()}, strict=False)
    ft64 = model64.fusion_transformer
    tok64 = O.synthetic_tokens(2 * B, dtype=torch.float64).requires_grad_(True)
    f64, inter64 = run_stack(ft64, tok64, intr.clone().double())
    out["vit_feat_f64"] = f64.detach().numpy()
    out["vit_block0_sub_f64"] = subsample(inter64[0])
    out["vit_block4_sub_f64"] = subsample(inter64[4])

    # ---- B: gradients of <feat, cot> through the reference (fp64) -----------------------------
    cot = O.closed_form(tuple(f64.shape), 991, 1.0, dtype=torch.float64)
    for p_ in ft64.parameters():
        p_.grad = None
    (f64 * cot).sum().backward()
    out["grad_tokens_sub_f64"] = subsample(tok64.grad)
    out["grad_tokens_sum_f64"] = summarize(tok64.grad)
    gnames = []
    gsum = []
    for n_, p_ in ft64.named_parameters():
        if p_.grad is None:
            continue
        gnames.append("fusion_transformer." + n_)
        gsum.append(summarize(p_.grad))
    out["grad_param_summaries_f64"] = np.stack(gsum)
    with open(os.path.join(HERE, "grad_param_names.json"), "w") as f:
        json.dump(gnames, f)

    # ---- a14: the ablation flags that actually run in the reference (no_pos_encoding crashes there: proj_fundamental is
    # always Linear(210,192), vision_transformer.py:179 vs :225-227).  fp64 features + token gradients per variant. ----
    for tag, kw in {"l1": dict(l1_pos_encoding=True), "single": dict(use_single_softmax=True),
                    "cross": dict(cross_features=True),
                    "all3": dict(l1_pos_encoding=True, use_single_softmax=True, cross_features=True)}.items():
        mv = ViTEss(ref_args(**kw)).eval()
        mv.load_state_dict(sd32, strict=False)
        mv = mv.double()
        mv.load_state_dict({k: v for k, v in sd64.items()}, strict=False)
        tv = O.synthetic_tokens(2 * B, dtype=torch.float64).requires_grad_(True)
        fv, _ = run_stack(mv.fusion_transformer, tv, intr.clone().double())
        out["variant_%s_feat_sub_f64" % tag] = subsample(fv, 5)
        (fv * cot).sum().backward()
        out["variant_%s_grad_tokens_sub_f64" % tag] = subsample(tv.grad)
        out["variant_%s_grad_qkv_sum_f64" % tag] = summarize(mv.fusion_transformer.blocks[5].cross_attn.qkv.weight.grad)

    # ---- regressor + normalise on the stack output (fp32, fp64) --------------------------------
    Gs = torch.tensor([0, 0, 0, 0, 0, 0, 1.0]).repeat(B, 2, 1)
    with torch.no_grad():
        pp32 = model.pose_regressor(f32.reshape(B, -1))
        out["pose_from_tokens_f32"] = model.normalize_preds(SE3(Gs), pp32, False)[0].data.numpy()
        pp64 = model64.pose_regressor(f64.detach().reshape(B, -1))
        out["pose_from_tokens_f64"] = model64.normalize_preds(SE3(Gs.double()), pp64, False)[0].data.numpy()

    # ---- C: positional encodings ---------------------------------------------------------------
    out["posenc_intr_f32"] = RVT.get_positional_encodings(B, 576, intrinsics=intr.clone()).numpy()
    out["posenc_none_f32"] = RVT.get_positional_encodings(B, 576, intrinsics=None).numpy()
    mp = torch.tensor([[517.97, 517.97, 320, 240]] * 2)[None].repeat(1, 1, 1)
    mp[:, :, [0, 2]] *= 24 / 512
    mp[:, :, [1, 3]] *= 24 / 384
    out["posenc_matterport_f32"] = RVT.get_positional_encodings(1, 576, intrinsics=mp).numpy()

    # ---- D: full ViTEss forward (stand-in trunk), eval + train mode ---------------------------
    for tag, (Bf, H, W) in {"sq": (2, 384, 384), "rect": (1, 256, 320)}.items():
        imgs = O.synthetic_images(Bf, H, W, key=7 if tag == "sq" else 8)
        intr_px = torch.tensor([[0.9 * W, 0.8 * W, W / 2.0, H / 2.0]]).repeat(Bf, 2, 1).contiguous()
        Gsf = torch.tensor([0, 0, 0, 0, 0, 0, 1.0]).repeat(Bf, 2, 1)
        with torch.no_grad():
            i_in = intr_px.clone()
            feats, i_out = model.extract_features(imgs.clone(), i_in.clone())
            out["full_%s_tokens_sub_f32" % tag] = subsample(feats, 101)
            pose = model(imgs.clone(), SE3(Gsf), intrinsics=i_in)[0].data
        out["full_%s_pose_f32" % tag] = pose.numpy()
        out["full_%s_intr_after" % tag] = i_in.numpy()          # mutated in place by forward
        with torch.no_grad():
            pose64 = model64(imgs.double(), SE3(Gsf.double()), intrinsics=intr_px.clone().double())[0].data
        out["full_%s_pose_f64" % tag] = pose64.numpy()
    model.train()
    with torch.no_grad():
        imgs = O.synthetic_images(2, 384, 384, key=7)
        intr_px = torch.tensor([[0.9 * 384, 0.8 * 384, 192.0, 192.0]]).repeat(2, 2, 1).contiguous()
        pose_tr = model(imgs, SE3(torch.tensor([0, 0, 0, 0, 0, 0, 1.0]).repeat(2, 2, 1)), intrinsics=intr_px)[0].data
    out["full_sq_pose_trainmode_f32"] = pose_tr.numpy()
    model.eval()

    # ---- F: index ops (bit-exact rows) ---------------------------------------------------------
    for n_in in (256, 320, 384, 480, 512, 640):
        ramp = torch.arange(n_in, dtype=torch.float32)[None, None, None, :].repeat(1, 1, 2, 1)
        res = torch.nn.functional.interpolate(ramp, size=(2, 224))
        out["nearest224_from_%d" % n_in] = res[0, 0, 0].numpy().astype(np.int32)
    x = torch.arange(2 * 192 * 576, dtype=torch.float32).reshape(2, 192, 24, 24)
    tk = x.reshape(2, -1, 576)[:, :192].permute(0, 2, 1)
    out["token_layout_probe"] = tk[1, ::97, ::31].contiguous().numpy().astype(np.int64)

    np.savez_compressed(os.path.join(HERE, "reference_outputs.npz"), **out)
    sz = os.path.getsize(os.path.join(HERE, "reference_outputs.npz"))
    print("wrote reference_outputs.npz: %d arrays, %.1f KB" % (len(out), sz / 1024))


def noess_fixtures():
    """SURVEY 8a row a14, --noess: plain cross attention in blocks[5] + pool_attn conv head.  Separate file so the
    default-configuration vectors above stay byte-stable."""
    out = {}
    B = 2
    model = ViTEss(ref_args(noess="1")).eval()
    with open(os.path.join(HERE, "state_dict_keys_noess.json"), "w") as f:
        json.dump({k: list(v.shape) for k, v in model.state_dict().items()}, f, indent=0, sort_keys=True)
    shapes = dict(O.vit_param_shapes(noess=True))
    shapes.update(O.cnn_param_shapes())
    sd32 = O.make_state(shapes, torch.float32)
    sd64 = O.make_state(shapes, torch.float64)
    missing, unexpected = model.load_state_dict(sd32, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith("resnet.layer3") or k.startswith("resnet.layer4") for k in missing), missing
    model64 = ViTEss(ref_args(noess="1")).eval()
    model64.load_state_dict(sd32, strict=False)
    model64 = model64.double()
    model64.load_state_dict(sd64, strict=False)
    intr = intrinsics_24()
    Gs = torch.tensor([0, 0, 0, 0, 0, 0, 1.0]).repeat(B, 2, 1)

    def from_tokens(m, tok, train):
        m.train(train)
        ft = m.fusion_transformer
        x = tok + ft.pos_embed
        for l in range(6):
            x = ft.blocks[l](x, intrinsics=intr.clone().to(tok.dtype))
        feats = ft.norm(x)
        f = feats.reshape([B, 24, 24, -1]).permute([0, 3, 1, 2])            # src/model.py:185
        pp = m.pose_regressor(m.pool_attn(f).reshape([B, -1]))
        pose = m.normalize_preds(SE3(Gs.to(tok.dtype)), pp, False)[0].data
        m.eval()
        return feats, pose

    # full model first: the train-mode passes below update pool_attn's BatchNorm running statistics
    imgs = O.synthetic_images(B, 384, 384, key=7)
    intr_px = torch.tensor([[0.9 * 384, 0.8 * 384, 192.0, 192.0]]).repeat(B, 2, 1).contiguous()
    with torch.no_grad():
        out["noess_full_sq_pose_f32"] = model(imgs.clone(), SE3(Gs), intrinsics=intr_px.clone())[0].data.numpy()
        out["noess_full_sq_pose_f64"] = model64(imgs.double(), SE3(Gs.double()), intrinsics=intr_px.clone().double())[0].data.numpy()
    with torch.no_grad():
        f32, p32 = from_tokens(model, O.synthetic_tokens(2 * B), False)
    out["noess_feat_sub_f32"] = subsample(f32, 23)
    out["noess_pose_from_tokens_f32"] = p32.numpy()
    for train in (False, True):       # eval before train for the same reason
        tag = "train" if train else "eval"
        tok64 = O.synthetic_tokens(2 * B, dtype=torch.float64).requires_grad_(True)
        for p_ in model64.parameters():
            p_.grad = None
        f64, p64 = from_tokens(model64, tok64, train)
        out["noess_pose_from_tokens_%s_f64" % tag] = p64.detach().numpy()
        if not train:
            out["noess_feat_sub_f64"] = subsample(f64, 23)
        cot = O.closed_form((B, 7), 993, 1.0, dtype=torch.float64)
        (p64[:, 1] * cot).sum().backward()
        out["noess_grad_tokens_sub_%s_f64" % tag] = subsample(tok64.grad)
        ca = model64.fusion_transformer.blocks[5].cross_attn
        out["noess_grad_sums_%s_f64" % tag] = np.stack([summarize(ca.qkv.weight.grad), summarize(ca.proj.weight.grad),
                                                         summarize(model64.pool_attn[0].weight.grad),
                                                         summarize(model64.pool_attn[4].weight.grad),
                                                         summarize(model64.pose_regressor[0].weight.grad)])
    np.savez_compressed(os.path.join(HERE, "reference_outputs_noess.npz"), **out)
    print("wrote reference_outputs_noess.npz: %d arrays, %.1f KB" %
          (len(out), os.path.getsize(os.path.join(HERE, "reference_outputs_noess.npz")) / 1024))


def demo_fixture():
    """SURVEY 8c, BASELINE configs[0]: the reference's OWN demo.py body (intrinsics selection :52-57, tensor assembly :65-81,
    post-processing :86-92) executed on its demo/matterport_{1,2}.png (640x480 RGBA) with a closed-form checkpoint -> the [7]
    vector it prints.  demo.py is a script (everything sits under `if __name__ == '__main__'`), needs cv2 / CUDA and cannot
    be imported: its source is read from the read-only reference at generation time and exec'd under harness shims
    (cv2.imread -> PIL decode in cv2's BGR / alpha-dropped convention, nn.Module.cuda -> identity); nothing of it is stored.
    The two PNGs are copied next to the fixture as DATA so that the GPU test can feed this repo's demo.py the same bytes."""
    import ast
    import shutil
    import tempfile
    from PIL import Image
    torch.nn.Module.cuda = lambda self, *a, **k: self
    cv2 = types.ModuleType("cv2")
    cv2.imread = lambda path: np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])
    sys.modules["cv2"] = cv2
    os.makedirs(os.path.join(HERE, "demo"), exist_ok=True)
    for n in ("matterport_1.png", "matterport_2.png"):
        shutil.copyfile(os.path.join(REF, "demo", n), os.path.join(HERE, "demo", n))
    shapes = dict(O.vit_param_shapes())
    shapes.update(O.cnn_param_shapes())
    sd32 = O.make_state(shapes, torch.float32)
    full = ViTEss(ref_args()).state_dict()                 # layer3/4 (unused by the forward) keep their constructed values
    full.update(sd32)
    tmp = tempfile.mkdtemp()
    ck = os.path.join(tmp, "matterport_closed_form.pth")    # "matterport" in the name selects that branch (demo.py:52,71,88)
    torch.save({"model": {"module." + k: v for k, v in full.items()}}, ck)
    src = open(os.path.join(REF, "demo.py")).read()
    tree = ast.parse(src)
    body = [n for n in tree.body if isinstance(n, ast.If)][-1].body          # statements under `if __name__ == '__main__':`
    mod = ast.Module(body=[n for n in tree.body if not isinstance(n, ast.If)] + body, type_ignores=[])
    argv, sys.argv = sys.argv, ["demo.py", "--img1", os.path.join(REF, "demo", "matterport_1.png"), "--img2",
                                os.path.join(REF, "demo", "matterport_2.png"), "--ckpt", ck]
    ns = {"__name__": "reference_demo"}
    try:
        torch.multiprocessing.set_start_method = lambda *a, **k: None
        exec(compile(mod, os.path.join(REF, "demo.py"), "exec"), ns)
    finally:
        sys.argv = argv
    out = {"demo_matterport_pred7_f32": np.asarray(ns["preds"], dtype=np.float32),
           "demo_matterport_raw7_f32": np.asarray(ns["pr_copy"], dtype=np.float32),
           "demo_matterport_images_sub": ns["images"][0, :, :, ::29, ::31].numpy().astype(np.float32)}
    np.savez_compressed(os.path.join(HERE, "reference_demo.npz"), **out)
    print("wrote reference_demo.npz:", out["demo_matterport_pred7_f32"])


def _reference_script(name, argv, workdir):
    """Execute one of the reference's evaluation SCRIPTS (everything under `if __name__ == '__main__'`) from the read-only
    reference under the harness shims, in `workdir`, with the model class wrapped so that its raw outputs are recorded.
    Returns the namespace after the run (predictions, camera_metrics, eval_camera ...).  Nothing of the script is stored."""
    import ast
    from PIL import Image
    torch.nn.Module.cuda = lambda self, *a, **k: self
    cv2 = types.ModuleType("cv2")
    cv2.imread = lambda path: np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])
    sys.modules["cv2"] = cv2
    torch.multiprocessing.set_start_method = lambda *a, **k: None
    tree = ast.parse(open(os.path.join(REF, name)).read())
    top = [n for n in tree.body if not isinstance(n, ast.If)]
    body = [n for n in tree.body if isinstance(n, ast.If)][-1].body
    ns = {"__name__": "reference_eval"}
    exec(compile(ast.Module(body=top, type_ignores=[]), os.path.join(REF, name), "exec"), ns)
    raw = []
    RefModel = ns["ViTEss"]

    class Recording(RefModel):
        def forward(self, *a, **k):
            out = super().forward(*a, **k)
            raw.append(out[0].data[0, 1].detach().clone().numpy())
            return out

    ns["ViTEss"] = Recording
    old_argv, old_cwd = sys.argv, os.getcwd()
    sys.argv = [name] + argv
    os.chdir(workdir)
    try:
        exec(compile(ast.Module(body=body, type_ignores=[]), os.path.join(REF, name), "exec"), ns)
    finally:
        sys.argv = old_argv
        os.chdir(old_cwd)
    ns["_raw_outputs"] = np.stack(raw)
    return ns


def _metric_arrays(prefix, metrics, out):
    out[prefix + "_metric_names"] = np.array(list(metrics.keys()))
    out[prefix + "_metric_values"] = np.array([float(v) for v in metrics.values()], dtype=np.float64)


def metrics_fixture():
    """the reference's OWN evaluation code -- test_matterport.py:27-68,117-164 and
    test_streetlearn_interiornet.py:26-128,194-244 -- executed here (a) as whole scripts on closed-form fake datasets
    (tests/_eval_cases.py) with the closed-form checkpoint, the reference model running on the CPU, and (b) function by
    function on hand-made prediction sets with the edge cases.  Only what the reference PRODUCED is committed
    (reference_metrics.npz): raw model outputs, the converted predictions / ground truths, metric values, the CSV / results files."""
    import tempfile
    out = {}
    shapes = dict(O.vit_param_shapes())
    shapes.update(O.cnn_param_shapes())
    sd32 = O.make_state(shapes, torch.float32)
    full = ViTEss(ref_args()).state_dict()
    full.update(sd32)
    tmp = tempfile.mkdtemp()
    ck = os.path.join(tmp, "closed_form.pth")
    torch.save({"model": {"module." + k: v for k, v in full.items()}}, ck)

    # ---- (a1) test_matterport.py as a script --------------------------------------------------------------------------
    root = os.path.join(tmp, "matterport_fake")
    EC.write_matterport(root)
    ns = _reference_script("test_matterport.py", ["--datapath", root, "--exp", "e0", "--ckpt", ck, "--fusion_transformer"], tmp)
    P = ns["predictions"]["camera"]
    out["mp_script_raw_outputs_f32"] = ns["_raw_outputs"].astype(np.float32)
    out["mp_script_pred_tran"] = np.vstack(P["preds"]["tran"])
    out["mp_script_pred_rot"] = np.vstack(P["preds"]["rot"])
    out["mp_script_gt_tran"] = np.vstack(P["gts"]["tran"]).astype(np.float64)
    out["mp_script_gt_rot"] = np.vstack(P["gts"]["rot"]).astype(np.float64)
    _metric_arrays("mp_script", ns["camera_metrics"], out)
    d = os.path.join(tmp, "output", "e0", "matterport_test")
    for f in ("results.txt", "gt_translation_magnitude_vs_error.csv", "gt_rotation_magnitude_vs_error.csv"):
        out["mp_script_file_" + f] = np.array(open(os.path.join(d, f)).read())
    mp_eval = ns["eval_camera"]

    # ---- (b1) eval_camera of test_matterport.py on the hand-made cases -----------------------------------------------
    for name, c in EC.matterport_metric_cases().items():
        ns["args"].exp = "k_" + name
        os.makedirs(os.path.join(tmp, "output", ns["args"].exp, ns["output_folder"]), exist_ok=True)
        pred = {"camera": {"preds": {"tran": list(c["pred_tran"]), "rot": list(c["pred_rot"])},
                           "gts": {"tran": list(c["gt_tran"]), "rot": list(c["gt_rot"])}}}
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            m = mp_eval(pred)
        finally:
            os.chdir(cwd)
        _metric_arrays("mp_case_" + name, m, out)
        d = os.path.join(tmp, "output", ns["args"].exp, ns["output_folder"])
        for f in ("gt_translation_magnitude_vs_error.csv", "gt_rotation_magnitude_vs_error.csv"):
            out["mp_case_%s_file_%s" % (name, f)] = np.array(open(os.path.join(d, f)).read())

    # ---- (a2) test_streetlearn_interiornet.py as a script (interiornet, rotation-only pairs) -------------------------
    proot = os.path.join(tmp, "pano_fake")
    split = EC.write_panorama(proot, "interiornet")
    ns = _reference_script("test_streetlearn_interiornet.py",
                           ["--datapath", proot, "--exp", "e1", "--ckpt", ck, "--dataset", "interiornet", "--fusion_transformer"], tmp)
    P = ns["predictions"]["camera"]
    out["pano_script_raw_outputs_f32"] = ns["_raw_outputs"].astype(np.float32)
    out["pano_script_pred_rot"] = np.vstack(P["preds"]["rot"])
    out["pano_script_gt_rot"] = np.vstack(P["gts"]["rot"]).astype(np.float64)
    _metric_arrays("pano_script", ns["camera_metrics"], out)
    d = os.path.join(tmp, "output", "e1", "interiornet_test")
    for f in ("results.txt", "all_rotation_err_degrees.csv", "all_gt_rot_degrees.csv"):
        out["pano_script_file_" + f] = np.array(open(os.path.join(d, f)).read())
    # the ground-truth construction alone, on more viewpoints than the 5 pairs of the fake split (compute_gt_rmat, :124-128)
    vp = O.hash_uniform(4 * 32, 123).reshape(32, 4) * np.array([1.5, 3.1, 1.5, 3.1])
    from scipy.spatial.transform import Rotation as _R
    gq = []
    for x1, y1, x2, y2 in vp:
        m = ns["compute_gt_rmat"](torch.tensor([[x1]]), torch.tensor([[y1]]), torch.tensor([[x2]]), torch.tensor([[y2]]), 1)
        gq.append(_R.from_matrix(m).as_quat()[0])
    out["pano_gt_quat_for_viewpoints"] = np.stack(gq)

    # ---- (b2) eval_camera of test_streetlearn_interiornet.py on the hand-made cases -----------------------------------
    for name, c in EC.panorama_metric_cases().items():
        d = os.path.join(tmp, "k_pano_" + name)
        os.makedirs(d, exist_ok=True)
        pred = {"camera": {"preds": {"rot": list(c["pred_rot"])}, "gts": {"rot": list(c["gt_rot"])}}}
        m = ns["eval_camera"](pred, d)
        _metric_arrays("pano_case_" + name, m, out)
        for f in ("all_rotation_err_degrees.csv", "all_gt_rot_degrees.csv"):
            out["pano_case_%s_file_%s" % (name, f)] = np.array(open(os.path.join(d, f)).read())

    np.savez_compressed(os.path.join(HERE, "reference_metrics.npz"), **out)
    print("wrote reference_metrics.npz: %d arrays, %.1f KB" % (len(out), os.path.getsize(os.path.join(HERE, "reference_metrics.npz")) / 1024))
    for k in ("mp_script", "pano_script"):
        print(k, dict(zip(out[k + "_metric_names"].tolist(), out[k + "_metric_values"].tolist())))


def _install_reader_shims():
    """Harness for the reference's data readers: cv2.imread on PIL (BGR, alpha dropped, None for an unreadable
    file like cv2) and a `torchvision.transforms` stand-in with torchvision's PIL-backend semantics for the five transforms the
    reference composes (src/data_readers/augmentation.py:12-16).  ColorJitter / RandomGrayscale do not draw: JITTER["mode"] is either
    None (both are the identity) or one fixed parameter set of tests/_eval_cases.FIXED_JITTER, applied the way torchvision's
    functional_pil does (ImageEnhance blends, uint8 HSV hue wrap-around, convert("L") greyscale)."""
    from PIL import Image, ImageEnhance
    cv2 = types.ModuleType("cv2")

    def imread(path):
        try:
            with Image.open(path) as im:
                return np.ascontiguousarray(np.asarray(im.convert("RGB"))[:, :, ::-1])
        except Exception:
            return None
    cv2.imread = imread
    sys.modules["cv2"] = cv2
    JITTER = {"mode": None, "ctor": {}}
    tvt = types.ModuleType("torchvision.transforms")

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class ToPILImage:
        def __call__(self, pic):              # to_pil_image of a float CHW tensor: mul(255).byte(), HWC, mode RGB
            assert pic.dim() == 3 and pic.shape[0] == 3 and pic.is_floating_point()
            return Image.fromarray(pic.mul(255).byte().permute(1, 2, 0).contiguous().numpy(), "RGB")

    class ToTensor:
        def __call__(self, img):              # pil_to_tensor + float32 div(255)
            a = torch.from_numpy(np.array(img, np.uint8, copy=True)).permute(2, 0, 1).contiguous()
            return a.to(torch.float32).div(255)

    class ColorJitter:
        def __init__(self, brightness=0, contrast=0, saturation=0, hue=0):
            JITTER["ctor"].update(brightness=brightness, contrast=contrast, saturation=saturation, hue=hue)

        def __call__(self, img):
            m = JITTER["mode"]
            if m is None:
                return img
            for op in m["order"]:
                if op == 0:
                    img = ImageEnhance.Brightness(img).enhance(m["b"])
                elif op == 1:
                    img = ImageEnhance.Contrast(img).enhance(m["c"])
                elif op == 2:
                    img = ImageEnhance.Color(img).enhance(m["s"])
                else:
                    h, s_, v = img.convert("HSV").split()
                    np_h = np.array(h, dtype=np.uint8)
                    np_h = (np_h.astype(np.int32) + int(m["h"] * 255)).astype(np.uint8)       # uint8 wrap-around
                    img = Image.merge("HSV", (Image.fromarray(np_h, "L"), s_, v)).convert("RGB")
            return img

    class RandomGrayscale:
        def __init__(self, p=0.1):
            JITTER["ctor"].update(p_gray=p)

        def __call__(self, img):
            m = JITTER["mode"]
            if m is None or not m["gray"]:
                return img
            g = np.array(img.convert("L"), dtype=np.uint8)
            return Image.fromarray(np.dstack([g, g, g]), "RGB")

    tvt.Compose, tvt.ToPILImage, tvt.ToTensor, tvt.ColorJitter, tvt.RandomGrayscale = Compose, ToPILImage, ToTensor, ColorJitter, RandomGrayscale
    sys.modules["torchvision"].transforms = tvt
    sys.modules["torchvision.transforms"] = tvt
    return JITTER


def readers_fixture():
    """the reference's OWN readers -- src/data_readers/{base,matterport,interiornet,streetlearn,factory,
    augmentation}.py -- run here on the closed-form fake training datasets of tests/_eval_cases.py.  Committed: what they PRODUCED
    (reference_readers.npz): lengths, file lists, scene_info poses / intrinsics, sub-epoch slices, samples (sha256 of the image tensor
    bytes + a subsample; whole images as uint8 for the fixed-jitter cases), the skip-forward behaviour on unreadable samples."""
    import hashlib
    import tempfile
    JITTER = _install_reader_shims()
    from src.data_readers.factory import dataset_factory           # reference
    from src.data_readers.matterport import Matterport            # reference
    from src.data_readers.interiornet import InteriorNet          # reference
    from src.data_readers.streetlearn import StreetLearn          # reference
    import src.data_readers.base as RB
    assert os.path.abspath(RB.__file__).startswith(os.path.abspath(REF)), RB.__file__
    out = {}
    tmp = tempfile.mkdtemp()

    def rec_scene(prefix, db, root):
        si = db.scene_info
        out[prefix + "_len"] = np.array(len(db))
        out[prefix + "_files"] = np.array("\n".join(os.path.relpath(f, root) for pair in si["images"] for f in pair))
        out[prefix + "_scene_poses"] = np.stack(si["poses"]).astype(np.float64) if len(db) else np.zeros((0, 2, 7))
        out[prefix + "_scene_intrinsics"] = np.stack(si["intrinsics"]).astype(np.float64) if len(db) else np.zeros((0, 2, 4))

    def rec_sample(prefix, sample, whole=False):
        im, po, K = sample
        assert im.dtype == torch.float32 and po.dtype == torch.float32 and K.dtype == torch.float32
        a = im.contiguous().numpy()
        out[prefix + "_images_shape"] = np.array(a.shape)
        out[prefix + "_images_sha256"] = np.array(hashlib.sha256(a.tobytes()).hexdigest())
        out[prefix + "_images_sub"] = a.reshape(-1)[::53].copy()
        if whole:
            assert np.array_equal(a, np.round(a)) and a.min() >= 0 and a.max() <= 255
            out[prefix + "_images_u8"] = a.astype(np.uint8)
        out[prefix + "_poses"] = po.numpy().copy()
        out[prefix + "_intrinsics"] = K.numpy().copy()

    # ---- Matterport: train (sub-epochs 0..9 read the same file) and val (sub-epoch 10) ---------------------------------------
    mroot = os.path.join(tmp, "matterport_fake")
    EC.write_matterport_train(mroot)
    for sub in (0, 4, 10):
        db = Matterport(datapath=mroot, subepoch=sub, reshape_size=[96, 128])
        rec_scene("mp_sub%d" % sub, db, mroot)
        for i in range(len(db)):
            rec_sample("mp_sub%d_i%d" % (sub, i), db[i])
    db = Matterport(datapath=mroot, subepoch=0)                         # the default reshape_size (384 x 512)
    rec_sample("mp_default_size_i2", db[2])
    cat = dataset_factory(["matterport"], datapath=mroot, subepoch=10, reshape_size=[48, 64])
    out["mp_factory_len"] = np.array(len(cat))
    rec_sample("mp_factory_i1", cat[1])
    for name, prm in EC.FIXED_JITTER.items():
        JITTER["mode"] = prm
        rec_sample("mp_%s_i3" % name, Matterport(datapath=mroot, subepoch=0, reshape_size=[96, 128])[3], whole=True)
    JITTER["mode"] = None

    # ---- the panorama datasets: rotation-only and translation ("T") sets, sub-epoch slices, mini dataset, skip-forward -------
    proot = os.path.join(tmp, "pano_fake")
    for (ds, typ) in sorted(EC.PANORAMA_TRAIN):
        EC.write_panorama_train(proot, ds, typ)
    for (ds, typ) in sorted(EC.PANORAMA_TRAIN):
        cls = InteriorNet if ds == "interiornet" else StreetLearn
        tag = "%s%s" % (ds, typ)
        for sub in (0, 1, 3, 9):
            db = cls(datapath=proot, subepoch=sub, streetlearn_interiornet_type=typ, reshape_size=[64, 80])
            rec_scene("%s_sub%d" % (tag, sub), db, proot)
        mini = cls(datapath=proot, subepoch=7, streetlearn_interiornet_type=typ, use_mini_dataset=True, reshape_size=[64, 80])
        rec_scene("%s_mini" % tag, mini, proot)
        # sub-epoch 0 = pairs 0..3 all readable; sub-epoch 1 = pairs 4..7 with 5 and 6 unreadable: index 1 and 2 skip forward to pair 7
        db0 = cls(datapath=proot, subepoch=0, streetlearn_interiornet_type=typ, reshape_size=[64, 80])
        db1 = cls(datapath=proot, subepoch=1, streetlearn_interiornet_type=typ, reshape_size=[64, 80])
        for i in range(4):
            rec_sample("%s_sub0_i%d" % (tag, i), db0[i])
            rec_sample("%s_sub1_i%d" % (tag, i), db1[i])
        # mini dataset: pair 13 (both files missing) skips to 14
        rec_sample("%s_mini_i13" % tag, mini[13])
        rec_sample("%s_mini_i12" % tag, mini[12])
        if typ == "":
            rec_sample("%s_default_size_i1" % tag, cls(datapath=proot, subepoch=0, streetlearn_interiornet_type=typ)[1])
        for name, prm in EC.FIXED_JITTER.items():
            JITTER["mode"] = prm
            rec_sample("%s_%s_i2" % (tag, name), cls(datapath=proot, subepoch=0, streetlearn_interiornet_type=typ, reshape_size=[64, 80])[2], whole=True)
        JITTER["mode"] = None
    cat = dataset_factory(["interiornet", "streetlearn"], datapath=proot, subepoch=0, streetlearn_interiornet_type="", reshape_size=[64, 80])
    out["pano_factory_len"] = np.array(len(cat))
    rec_sample("pano_factory_i5", cat[5])                               # second dataset of the concatenation, its pair 1
    out["jitter_ctor_names"] = np.array(sorted(JITTER["ctor"]))
    out["jitter_ctor_values"] = np.array([float(JITTER["ctor"][k]) for k in sorted(JITTER["ctor"])])
    path = os.path.join(HERE, "reference_readers.npz")
    np.savez_compressed(path, **out)
    print("wrote reference_readers.npz: %d arrays, %.1f KB" % (len(out), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    if "--readers-only" in sys.argv:
        readers_fixture()
        sys.exit(0)
    if "--metrics-only" in sys.argv:
        metrics_fixture()
        sys.exit(0)
    if "--demo-only" in sys.argv:
        demo_fixture()
        sys.exit(0)
    if "--noess-only" not in sys.argv:
        main()
    noess_fixtures()
    demo_fixture()
    metrics_fixture()
    readers_fixture()
