"""Dataset readers, augmentation and evaluation metrics on tiny fake datasets written in the
reference's on-disk layout (reference src/data_readers/*.py, test_matterport.py, test_streetlearn_interiornet.py)."""
import json
import os

import numpy as np
import pytest
import torch
from PIL import Image, ImageEnhance
from scipy.spatial.transform import Rotation

from rel_pose_amd import evaluation as E
from rel_pose_amd.data_readers import augmentation as A
from rel_pose_amd.data_readers import viewpoint as V


def _write_img(path, h, w, seed):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    rng = np.random.default_rng(seed)
    Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(path)


@pytest.fixture()
def matterport_root(tmp_path):
    root = tmp_path / "matterport_fake"
    data = []
    for i in range(3):
        names = []
        for k in range(2):
            rel = "rgb/house%d/img_%d_%d.png" % (i, i, k)
            _write_img(str(root / rel), 48, 64, 10 * i + k)
            names.append("/a/b/c/d/e/" + rel)              # the reader drops the first 6 path components
        q = Rotation.from_euler("xyz", [10 * i, -20, 5 + i], degrees=True).as_quat()       # xyzw
        wxyz = [float(q[3]), float(q[0]), float(q[1]), float(q[2])]
        if i == 1:
            wxyz = [-v for v in wxyz]                        # a negative-w ground truth
        data.append({"0": {"file_name": names[0]}, "1": {"file_name": names[1]},
                     "rel_pose": {"position": [1.0 + i, -2.0, 0.5], "rotation": wxyz}})
    for split in ("train", "val", "test"):
        os.makedirs(root / "mp3d_planercnn_json", exist_ok=True)
        with open(root / "mp3d_planercnn_json" / ("cached_set_%s.json" % split), "w") as f:
            json.dump({"data": data if split != "val" else data[:1]}, f)
    return str(root), data


def test_matterport_reader_conventions(matterport_root):
    from rel_pose_amd.data_readers.factory import dataset_factory
    from rel_pose_amd.data_readers.matterport import Matterport
    root, data = matterport_root
    db = Matterport(datapath=root, subepoch=0, reshape_size=[96, 128])
    assert len(db) == 3 and len(Matterport(datapath=root, subepoch=10)) == 1          # sub-epoch 10 = validation split
    images, poses, intr = db[1]
    assert images.shape == (2, 3, 96, 128) and images.dtype == torch.float32 and 0 <= float(images.min()) and float(images.max()) <= 255
    assert torch.equal(poses[0], torch.tensor([0, 0, 0, 0, 0, 0, 1.0]))
    t = np.array(data[1]["rel_pose"]["position"]) / 5.0                                # DEPTH_SCALE
    w, x, y, z = data[1]["rel_pose"]["rotation"]
    # the reference swaps slots 3 and 6 of (t, w, x, y, z) (matterport.py:46-48): the stored quaternion is (z, x, y, w) --
    # a quirk the evaluation (test_matterport.py:150-151) and demo.py:86-92 undo with the same swap; sign made w-positive
    q = -np.array([z, x, y, w])
    assert np.allclose(poses[1].numpy(), np.concatenate([t, q]), atol=1e-6) and poses[1, 6] > 0
    # intrinsics of the 480x640 camera rescaled by reshape/actual image size (augmentation.py:28-34)
    assert torch.allclose(intr[0], torch.tensor([517.97 * 128 / 64, 517.97 * 96 / 48, 320 * 128 / 64, 240 * 96 / 48]))
    loader = torch.utils.data.DataLoader(dataset_factory(["matterport"], datapath=root, subepoch=0, reshape_size=[96, 128]), batch_size=3)
    bi, bp, bk = next(iter(loader))
    assert bi.shape == (3, 2, 3, 96, 128) and bp.shape == (3, 2, 7) and bk.shape == (3, 2, 4)
    import src.data_readers.factory as alias                                            # drop-in module path of the reference
    assert alias.dataset_factory is dataset_factory


def test_raw_reader_mode_and_batched_parameter_draws(matterport_root):
    """raw=True (decode-only workers for the GPU augmentation path): uint8 BGR as decoded, unscaled intrinsics, same poses;
    draw_batch rows are valid rp_augment_pairs parameter rows inside ColorJitter's ranges (augmentation.py:12-16)."""
    from rel_pose_amd.data_readers.augmentation import RGBDAugmentor
    from rel_pose_amd.data_readers.base import imread_bgr
    from rel_pose_amd.data_readers.matterport import Matterport
    root, data = matterport_root
    db = Matterport(datapath=root, subepoch=0, reshape_size=[96, 128], raw=True)
    ref = Matterport(datapath=root, subepoch=0, reshape_size=[96, 128])
    images, poses, intr = db[2]
    assert images.dtype == torch.uint8 and images.shape == (2, 48, 64, 3)
    assert np.array_equal(images[1].numpy(), imread_bgr(ref.scene_info["images"][2][1]))
    assert torch.equal(poses, ref[2][1]) and torch.allclose(intr[0], torch.tensor([517.97, 517.97, 320.0, 240.0]))
    bi, bp, bk = next(iter(torch.utils.data.DataLoader(db, batch_size=3)))
    assert bi.shape == (3, 2, 48, 64, 3) and bi.dtype == torch.uint8
    aug = RGBDAugmentor(reshape_size=[96, 128], generator=torch.Generator().manual_seed(0))
    prm = aug.draw_batch(4000)
    assert prm.shape == (4000, 9) and torch.equal(prm[:, :4].sort(dim=1).values, torch.arange(4.0).expand(4000, 4))
    assert len({tuple(r) for r in prm[:, :4].int().tolist()}) == 24                         # every order occurs
    for col, lo, hi in ((4, .75, 1.25), (5, .75, 1.25), (6, .75, 1.25), (7, -.4 / 3.14, .4 / 3.14)):
        assert lo <= float(prm[:, col].min()) and float(prm[:, col].max()) <= hi and float(prm[:, col].std()) > 0.25 * (hi - lo)
    assert 0.07 < float(prm[:, 8].mean()) < 0.13                                            # RandomGrayscale(p=0.1)
    d = RGBDAugmentor.params_to_dict(prm[0])
    assert sorted(d["order"]) == [0, 1, 2, 3] and isinstance(d["gray"], bool)


def test_panorama_readers_and_viewpoint_rotation(tmp_path):
    from rel_pose_amd.data_readers.interiornet import InteriorNet
    from rel_pose_amd.data_readers.streetlearn import StreetLearn
    root = tmp_path / "pano"
    rng = np.random.default_rng(3)
    for meta, folder, cls, typ in (("interiornet", "interiornet", InteriorNet, ""), ("streetlearnT", "streetlearn_2016", StreetLearn, "T")):
        split = {}
        for i in range(40):
            p1, p2 = "s%d/a.png" % i, "s%d/b.png" % i
            if i < 12:
                _write_img(str(root / "data" / folder / p1), 32, 32, i)
                _write_img(str(root / "data" / folder / p2), 32, 32, 100 + i)
            split[i] = {"img1": {"path": p1, "x": float(rng.uniform(-0.5, 0.5)), "y": float(rng.uniform(-3, 3))},
                        "img2": {"path": p2, "x": float(rng.uniform(-0.5, 0.5)), "y": float(rng.uniform(-3, 3))}}
        os.makedirs(root / "metadata" / meta, exist_ok=True)
        name = "train_pair_translation_overlap.npy" if typ else "train_pair_rotation_overlap.npy"
        np.save(root / "metadata" / meta / name, split, allow_pickle=True)
        db = cls(datapath=str(root), subepoch=2, streetlearn_interiornet_type=typ, reshape_size=[64, 64])
        assert len(db) == 4                                          # a tenth of the 40 pairs: keys 8..11
        images, poses, intr = db[0]
        assert images.shape == (2, 3, 64, 64) and torch.allclose(intr, torch.full((2, 4), 128.0 * 2))
        a, b = split[8]["img1"], split[8]["img2"]
        assert np.allclose(poses[1, 3:].numpy(), V.relative_quaternion(a["x"], a["y"], b["x"], b["y"]), atol=1e-6)
        assert torch.equal(poses[1, :3], torch.zeros(3))             # rotation-only ground truth
        mini = cls(datapath=str(root), subepoch=7, streetlearn_interiornet_type=typ, use_mini_dataset=True)
        assert len(mini) == 40                                       # first 32000 pairs regardless of the sub-epoch
    # unreadable samples are skipped forward (base.py:72-97): pair 12 has no files -> error only past the end
    with pytest.raises(OSError):
        mini[39]
    # the viewpoint rotation: orthonormal, identity for equal views, pure yaw difference = that angle about the y axis
    R = V.relative_rotation(0.3, 1.0, -0.2, 2.5)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-6) and np.isclose(np.linalg.det(R), 1.0, atol=1e-6)
    assert np.allclose(V.relative_rotation(0.1, 0.7, 0.1, 0.7), np.eye(3), atol=1e-6)
    ang = np.degrees(np.arccos((np.trace(V.relative_rotation(0.0, 0.2, 0.0, 0.9)) - 1) / 2))
    assert np.isclose(ang, np.degrees(0.7), atol=1e-3)


def _pil_hue(img, hue_factor):
    """torchvision functional_pil.adjust_hue (uint8 H channel + uint8(hue_factor * 255), wrapping)"""
    h, s_, v = img.convert("HSV").split()
    nh = (np.array(h, dtype=np.uint8).astype(np.int32) + int(hue_factor * 255)).astype(np.uint8)
    return Image.merge("HSV", (Image.fromarray(nh, "L"), s_, v)).convert("RGB")


def test_colour_ops_are_pils_arithmetic_bit_for_bit():
    """brightness / contrast / saturation = PIL's ImageEnhance blends, hue = the uint8 HSV round trip, greyscale = convert("L") -- what
    torchvision's PIL backend calls on the reference's ToPILImage output (reference augmentation.py:12-16) -- EXACTLY, on random
    images and for the mode conversions on ALL 2^24 colours."""
    rng = np.random.default_rng(0)
    rgb8 = rng.integers(0, 256, (96, 128, 3), dtype=np.uint8)
    rgb8[0, :, :] = rgb8[0, :, :1]                                  # a row of greys (min == max: hue / saturation 0)
    pil = Image.fromarray(rgb8)
    x = torch.from_numpy(rgb8).permute(2, 0, 1)
    hwc = lambda t: t.permute(1, 2, 0).numpy()                       # noqa: E731
    for fn, enh in ((A.adjust_brightness, ImageEnhance.Brightness), (A.adjust_saturation, ImageEnhance.Color), (A.adjust_contrast, ImageEnhance.Contrast)):
        for f in (0.75, 0.8, 0.9999, 1.0, 1.0001, 1.2, 1.25, 0.0, 2.0):
            assert np.array_equal(hwc(fn(x, f)), np.asarray(enh(pil).enhance(f))), (fn.__name__, f)
    for hf in (0.0, 0.06, -0.1, 0.4 / 3.14, -0.4 / 3.14, 0.5, -0.5, 1.0 / 255, -1.0 / 255):
        assert np.array_equal(hwc(A.adjust_hue(x, hf)), np.asarray(_pil_hue(pil, hf))), hf
    assert np.array_equal(hwc(A.to_grayscale(x))[..., 1], np.asarray(pil.convert("L")))
    # the three mode conversions, exhaustively (16.7 M colours each)
    ax = np.arange(256, dtype=np.uint8)
    cube = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(4096, 4096, 3)
    t = torch.from_numpy(cube).permute(2, 0, 1)
    cpil = Image.fromarray(cube)
    assert np.array_equal(A._luma_u8(t)[0].numpy(), np.asarray(cpil.convert("L")))
    assert np.array_equal(hwc(A._rgb2hsv_u8(t)), np.asarray(cpil.convert("HSV")))
    assert np.array_equal(hwc(A._hsv2rgb_u8(t)), np.asarray(Image.fromarray(cube, "HSV").convert("RGB")))


def test_augmentor_pair_semantics_and_batch():
    aug = A.RGBDAugmentor([40, 48], generator=torch.Generator().manual_seed(5))
    img = torch.floor(torch.rand(1, 3, 20, 24, generator=torch.Generator().manual_seed(1)) * 255)
    pair = img.repeat(2, 1, 1, 1)                                    # two identical images
    out = aug.color_transform(pair)
    assert torch.equal(out[0], out[1])                               # one parameter draw per sample, like the reference
    assert out.shape == pair.shape and float(out.min()) >= 0 and float(out.max()) <= 255
    ident = dict(order=[0, 1, 2], b=1.0, c=1.0, s=1.0, h=0.0, gray=False)         # (PIL's uint8 HSV round trip is lossy even at shift 0)
    assert torch.equal(A.RGBDAugmentor.apply(pair, ident), pair) and torch.equal(A.RGBDAugmentor.apply(pair, A.RGBDAugmentor.IDENTITY), pair)
    gray = A.RGBDAugmentor.apply(pair, dict(ident, gray=True))
    assert torch.equal(gray[:, 0], gray[:, 1]) and torch.equal(gray[:, 1], gray[:, 2])
    # draws stay inside torchvision's ColorJitter ranges
    for _ in range(50):
        p = aug.draw()
        assert 0.75 <= p["b"] <= 1.25 and 0.75 <= p["c"] <= 1.25 and 0.75 <= p["s"] <= 1.25 and abs(p["h"]) <= 0.4 / 3.14
        assert sorted(p["order"]) == [0, 1, 2, 3]
    intr = torch.tensor([[10.0, 20.0, 12.0, 10.0]] * 2)
    images, poses, k = aug(pair.clone(), torch.zeros(2, 7), intr)
    assert images.shape == (2, 3, 40, 48) and torch.allclose(k[0], torch.tensor([20.0, 40.0, 24.0, 20.0]))
    batch = torch.floor(torch.rand(3, 2, 3, 20, 24) * 255)
    kb = torch.tensor([[[10.0, 20.0, 12.0, 10.0]] * 2] * 3)
    ob, kb2 = aug.augment_batch(batch, kb)
    assert ob.shape == (3, 2, 3, 40, 48) and torch.allclose(kb2[2, 1], torch.tensor([20.0, 40.0, 24.0, 20.0]))


def test_matterport_metrics_known_answers(tmp_path):
    gt_t = [[1.0, 0, 0], [0, 2.0, 0], [0, 0, 3.0], [1.0, 1.0, 1.0]]
    gt_q = [Rotation.from_euler("z", a, degrees=True).as_quat()[[3, 0, 1, 2]] for a in (0, 40, 90, 10)]     # wxyz
    errs_deg = (5.0, 20.0, 45.0, 0.0)
    pred_q = [(Rotation.from_euler("z", a, degrees=True) * Rotation.from_euler("x", e, degrees=True)).as_quat()[[3, 0, 1, 2]]
              for a, e in zip((0, 40, 90, 10), errs_deg)]
    pred_q[1] = -pred_q[1]                                            # sign of a quaternion must not matter
    pred_t = [[1.5, 0, 0], [0, 2.0, 0], [0, 0, 5.0], [1.0, 1.0, 1.0]]
    m = E.camera_metrics_matterport(pred_t, pred_q, gt_t, gt_q, str(tmp_path / "out"))
    assert np.isclose(m["R mean err"], np.mean(errs_deg), atol=1e-4) and np.isclose(m["R median err"], 12.5, atol=1e-4)
    assert np.isclose(m["top1 R err < 30"], 75.0) and np.isclose(m["top1 T err < 1.0"], 75.0)
    assert np.isclose(m["T mean err"], (0.5 + 0 + 2 + 0) / 4) and np.isclose(m["T median err"], 0.25)
    rows = np.loadtxt(tmp_path / "out" / "gt_rotation_magnitude_vs_error.csv", delimiter=",")
    assert rows.shape == (4, 2) and np.allclose(rows[:, 0], [0, 40, 90, 10], atol=1e-3)
    # model output -> evaluation convention (test_matterport.py:147-153): w back in front, metres again
    t, q = E.matterport_prediction(np.array([0.2, -0.4, 0.1, 0.1, 0.2, 0.3, 0.9]))
    assert np.allclose(t, [1.0, -2.0, 0.5]) and np.allclose(q, [0.9, 0.2, 0.3, 0.1])
    assert np.allclose(E.matterport_gt_rotation([-0.5, 0.5, 0.5, 0.5]), [0.5, -0.5, -0.5, -0.5])


def test_panorama_rotation_metrics_known_answers(tmp_path):
    gt = [Rotation.from_euler("y", a, degrees=True) for a in (10, 30, 60, 80, 120)]
    err = (2.0, 15.0, 8.0, 30.0, 1.0)
    pred = [Rotation.from_euler("x", e, degrees=True) * g for g, e in zip(gt, err)]
    m = E.rotation_metrics_panorama([p.as_quat() for p in pred], [g.as_quat() for g in gt], str(tmp_path / "o"))
    assert np.isclose(m["rotation_geodesic_error_overlap_large/mean"], 8.5, atol=1e-4)       # gt angle < 45: errors 2, 15
    assert np.isclose(m["rotation_geodesic_error_overlap_large/10deg"], 0.5)
    assert np.isclose(m["rotation_geodesic_error_overlap_small/median"], 19.0, atol=1e-4)   # 45 <= gt < 90: errors 8, 30
    assert np.loadtxt(tmp_path / "o" / "all_rotation_err_degrees.csv").shape == (4,)         # the 120-degree pair is dropped


# ---------------------------------------------------------------------------------------------------------------------
# SURVEY 8f row 4 pinned against the REFERENCE's own evaluation code: tests/golden/reference_metrics.npz holds what
# test_matterport.py / test_streetlearn_interiornet.py of the reference produced (executed by tests/golden/make_fixtures.py
# in the build container) on the closed-form inputs of tests/_eval_cases.py.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ref_metrics():
    here = os.path.dirname(os.path.abspath(__file__))
    return np.load(os.path.join(here, "golden", "reference_metrics.npz"))


def _same_metrics(got, ref, prefix, tol=0.0):
    names = ref[prefix + "_metric_names"].tolist()
    vals = ref[prefix + "_metric_values"]
    assert list(got.keys()) == names, (list(got.keys()), names)                       # same keys in the same order
    for n, v in zip(names, vals):
        assert abs(float(got[n]) - v) <= tol * max(1.0, abs(v)), (prefix, n, float(got[n]), v)


def test_matterport_metrics_equal_the_references_eval_camera(ref_metrics, tmp_path):
    from tests import _eval_cases as EC
    for name, c in EC.matterport_metric_cases().items():
        out = str(tmp_path / name)
        m = E.camera_metrics_matterport(list(c["pred_tran"]), list(c["pred_rot"]), list(c["gt_tran"]), list(c["gt_rot"]), out)
        _same_metrics(m, ref_metrics, "mp_case_" + name)                               # same numpy arithmetic: bit-identical
        for f in ("gt_translation_magnitude_vs_error.csv", "gt_rotation_magnitude_vs_error.csv"):
            assert open(os.path.join(out, f)).read() == str(ref_metrics["mp_case_%s_file_%s" % (name, f)]), (name, f)


def test_matterport_script_conversions_equal_the_references(ref_metrics, tmp_path):
    """the per-sample conversions of test_matterport.py:138-156 (w-positive ground truth; w back in front and metres again for
    the prediction) and the metrics / files of the whole run, from the raw model outputs the reference's model produced"""
    from tests import _eval_cases as EC
    data = EC.matterport_entries()
    raw = ref_metrics["mp_script_raw_outputs_f32"]
    pt, pr, gt_t, gt_r = [], [], [], []
    for e, r in zip(data, raw):
        t, q = E.matterport_prediction(r)
        pt.append(t)
        pr.append(q)
        gt_t.append(e["rel_pose"]["position"])
        gt_r.append(E.matterport_gt_rotation(e["rel_pose"]["rotation"]))
    assert np.array_equal(np.vstack(pt), ref_metrics["mp_script_pred_tran"])
    assert np.array_equal(np.vstack(pr), ref_metrics["mp_script_pred_rot"])
    assert np.array_equal(np.vstack(gt_t), ref_metrics["mp_script_gt_tran"])
    assert np.array_equal(np.vstack(gt_r), ref_metrics["mp_script_gt_rot"])
    assert (ref_metrics["mp_script_gt_rot"][:, 0] >= 0).all()
    out = str(tmp_path / "o")
    m = E.camera_metrics_matterport(pt, pr, gt_t, gt_r, out)
    _same_metrics(m, ref_metrics, "mp_script")
    for f in ("gt_translation_magnitude_vs_error.csv", "gt_rotation_magnitude_vs_error.csv"):
        assert open(os.path.join(out, f)).read() == str(ref_metrics["mp_script_file_" + f])
    text = "".join("%s %s\n" % (k, v) for k, v in m.items())                          # what the script prints into results.txt
    assert text == str(ref_metrics["mp_script_file_results.txt"])


def test_panorama_metrics_equal_the_references_eval_camera(ref_metrics, tmp_path):
    from tests import _eval_cases as EC
    for name, c in EC.panorama_metric_cases().items():
        out = str(tmp_path / name)
        m = E.rotation_metrics_panorama(list(c["pred_rot"]), list(c["gt_rot"]), out)
        _same_metrics(m, ref_metrics, "pano_case_" + name)
        for f in ("all_rotation_err_degrees.csv", "all_gt_rot_degrees.csv"):
            assert open(os.path.join(out, f)).read() == str(ref_metrics["pano_case_%s_file_%s" % (name, f)]), (name, f)
    assert len(ref_metrics["pano_case_none_metric_names"]) == 0                        # both buckets empty: the reference returns {}


def test_panorama_script_ground_truth_and_metrics_equal_the_references(ref_metrics, tmp_path):
    """ground-truth quaternions as test_streetlearn_interiornet.py:53-69,124-128,204-211 builds them (float32 viewpoint matrices ->
    scipy), on the fake split and on 32 more viewpoint pairs; metrics and files of the whole run from the reference's raw outputs"""
    from oracle import relpose_oracle as O
    from tests import _eval_cases as EC
    split = EC.panorama_entries()
    gt = [V.relative_quaternion(e["img1"]["x"], e["img1"]["y"], e["img2"]["x"], e["img2"]["y"]) for _, e in sorted(split.items())]
    assert np.array_equal(np.vstack(gt), ref_metrics["pano_script_gt_rot"])
    vp = O.hash_uniform(4 * 32, 123).reshape(32, 4) * np.array([1.5, 3.1, 1.5, 3.1])
    assert np.array_equal(np.stack([V.relative_quaternion(*v) for v in vp]), ref_metrics["pano_gt_quat_for_viewpoints"])
    raw = ref_metrics["pano_script_raw_outputs_f32"]
    pred = [r[3:] for r in raw]
    assert np.array_equal(np.vstack(pred), ref_metrics["pano_script_pred_rot"])
    out = str(tmp_path / "o")
    m = E.rotation_metrics_panorama(pred, gt, out)
    _same_metrics(m, ref_metrics, "pano_script")
    for f in ("all_rotation_err_degrees.csv", "all_gt_rot_degrees.csv"):
        assert open(os.path.join(out, f)).read() == str(ref_metrics["pano_script_file_" + f])
    assert "".join("%s %s\n" % (k, v) for k, v in m.items()) == str(ref_metrics["pano_script_file_results.txt"])


# ---------------------------------------------------------------------------------------------------------------------
# SURVEY 8f row 3 pinned against the reference's OWN readers: tests/golden/reference_readers.npz holds what
# /root/reference/src/data_readers/{base,matterport,interiornet,streetlearn,factory,augmentation}.py produced on the
# closed-form fake training datasets of tests/_eval_cases.py (tests/golden/make_fixtures.py --readers-only, build
# container only; the torchvision.transforms stand-in of that script has ColorJitter / RandomGrayscale as the identity or
# as ONE fixed parameter set applied with PIL the way torchvision's functional_pil does).
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ref_readers():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_readers.npz"))


@pytest.fixture(scope="module")
def reader_roots(tmp_path_factory):
    from tests import _eval_cases as EC
    tmp = tmp_path_factory.mktemp("readers")
    mroot, proot = str(tmp / "matterport_fake"), str(tmp / "pano_fake")
    EC.write_matterport_train(mroot)
    for ds, typ in sorted(EC.PANORAMA_TRAIN):
        EC.write_panorama_train(proot, ds, typ)
    return mroot, proot


def _same_scene(ref, prefix, db, root):
    si = db.scene_info
    assert len(db) == int(ref[prefix + "_len"]), prefix
    files = "\n".join(os.path.relpath(f, root) for pair in si["images"] for f in pair)
    assert files == str(ref[prefix + "_files"]), prefix
    if len(db):
        assert np.array_equal(np.stack(si["poses"]).astype(np.float64), ref[prefix + "_scene_poses"]), prefix
        assert np.array_equal(np.stack(si["intrinsics"]).astype(np.float64), ref[prefix + "_scene_intrinsics"]), prefix


def _same_sample(ref, prefix, sample):
    """bit for bit: shape, the sha256 of the fp32 image bytes, poses, rescaled intrinsics"""
    import hashlib
    im, po, K = sample
    a = im.contiguous().numpy()
    assert a.dtype == np.float32 and list(a.shape) == ref[prefix + "_images_shape"].tolist(), prefix
    assert np.array_equal(a.reshape(-1)[::53], ref[prefix + "_images_sub"]), prefix
    assert hashlib.sha256(a.tobytes()).hexdigest() == str(ref[prefix + "_images_sha256"]), prefix
    assert po.dtype == torch.float32 and np.array_equal(po.numpy(), ref[prefix + "_poses"]), prefix
    assert K.dtype == torch.float32 and np.array_equal(K.numpy(), ref[prefix + "_intrinsics"]), prefix


def test_matterport_reader_reproduces_the_references_reader_bit_for_bit(ref_readers, reader_roots):
    """reference src/data_readers/matterport.py:21-62 + base.py:45-69 + augmentation.py:28-37 with the jitter off: scene-info
    construction (file-name rewrite, DEPTH_SCALE, the 3<->6 slot swap, w >= 0), train / val file selection by sub-epoch
    (base.py:30), float32 casts, intrinsics rescale, nearest resize -- every sample identical to the last bit."""
    from rel_pose_amd.data_readers.factory import dataset_factory
    from rel_pose_amd.data_readers.matterport import Matterport
    mroot, _ = reader_roots
    for sub in (0, 4, 10):
        db = Matterport(datapath=mroot, subepoch=sub, reshape_size=[96, 128], jitter=False)
        _same_scene(ref_readers, "mp_sub%d" % sub, db, mroot)
        for i in range(len(db)):
            _same_sample(ref_readers, "mp_sub%d_i%d" % (sub, i), db[i])
    _same_sample(ref_readers, "mp_default_size_i2", Matterport(datapath=mroot, subepoch=0, jitter=False)[2])     # 384 x 512 default
    cat = dataset_factory(["matterport"], datapath=mroot, subepoch=10, reshape_size=[48, 64], jitter=False)
    assert len(cat) == int(ref_readers["mp_factory_len"])
    _same_sample(ref_readers, "mp_factory_i1", cat[1])


def test_panorama_readers_reproduce_the_references_readers_bit_for_bit(ref_readers, reader_roots):
    """reference interiornet.py:53-107 / streetlearn.py:53-108 + base.py:70-97: metadata file and image folder per
    (dataset, type), the ten sub-epoch slices (43 pairs: 4 per slice, 3 never read), the mini dataset, viewpoint ->
    quaternion ground truth, and the skip-forward on unreadable samples (pair 5: missing file, pair 6: undecodable file ->
    indices 1 and 2 of sub-epoch 1 both return pair 7; pair 13 with both files missing -> pair 14)."""
    from rel_pose_amd.data_readers.factory import dataset_factory
    from rel_pose_amd.data_readers.interiornet import InteriorNet
    from rel_pose_amd.data_readers.streetlearn import StreetLearn
    from tests import _eval_cases as EC
    _, proot = reader_roots
    for ds, typ in sorted(EC.PANORAMA_TRAIN):
        cls = InteriorNet if ds == "interiornet" else StreetLearn
        tag = ds + typ
        kw = dict(datapath=proot, streetlearn_interiornet_type=typ, reshape_size=[64, 80], jitter=False)
        for sub in (0, 1, 3, 9):
            _same_scene(ref_readers, "%s_sub%d" % (tag, sub), cls(subepoch=sub, **kw), proot)
        mini = cls(subepoch=7, use_mini_dataset=True, **kw)
        _same_scene(ref_readers, tag + "_mini", mini, proot)
        db0, db1 = cls(subepoch=0, **kw), cls(subepoch=1, **kw)
        for i in range(4):
            _same_sample(ref_readers, "%s_sub0_i%d" % (tag, i), db0[i])
            _same_sample(ref_readers, "%s_sub1_i%d" % (tag, i), db1[i])
        assert torch.equal(db1[1][0], db1[3][0]) and torch.equal(db1[2][1], db1[3][1])          # both skipped forward to pair 7
        _same_sample(ref_readers, tag + "_mini_i13", mini[13])
        _same_sample(ref_readers, tag + "_mini_i12", mini[12])
        if typ == "":
            _same_sample(ref_readers, tag + "_default_size_i1", cls(datapath=proot, subepoch=0, streetlearn_interiornet_type=typ, jitter=False)[1])
    cat = dataset_factory(["interiornet", "streetlearn"], datapath=proot, subepoch=0, streetlearn_interiornet_type="",
                          reshape_size=[64, 80], jitter=False)
    assert len(cat) == int(ref_readers["pano_factory_len"])
    _same_sample(ref_readers, "pano_factory_i5", cat[5])


def test_colour_jitter_ranges_are_the_references(ref_readers):
    """the arguments the reference hands to ColorJitter / RandomGrayscale (augmentation.py:14-15), as recorded by the stand-in"""
    ctor = dict(zip(ref_readers["jitter_ctor_names"].tolist(), ref_readers["jitter_ctor_values"].tolist()))
    aug = A.RGBDAugmentor([8, 8])
    assert ctor == dict(brightness=aug.brightness, contrast=aug.contrast, saturation=aug.saturation, hue=aug.hue, p_gray=aug.p_gray)


@pytest.mark.parametrize("name", ["j0", "j1", "j2"])
def test_fixed_colour_jitter_reproduces_the_references_reader_bit_for_bit(ref_readers, reader_roots, name):
    """A FIXED ColorJitter / RandomGrayscale draw (tests/_eval_cases.FIXED_JITTER: three op orders, both signs of the hue shift, grey
    on and off) through the reference's reader (ToPILImage, the PIL ops in that order on the glued pair, ToTensor, `255 *`, nearest
    resize) against this repo's reader with the same draw: identical image bytes, poses and intrinsics, for every dataset class."""
    from rel_pose_amd.data_readers.interiornet import InteriorNet
    from rel_pose_amd.data_readers.matterport import Matterport
    from rel_pose_amd.data_readers.streetlearn import StreetLearn
    from tests import _eval_cases as EC
    mroot, proot = reader_roots
    prm = EC.FIXED_JITTER[name]
    cases = [("mp_%s_i3" % name, Matterport(datapath=mroot, subepoch=0, reshape_size=[96, 128]), 3)]
    for ds, typ in sorted(EC.PANORAMA_TRAIN):
        cls = InteriorNet if ds == "interiornet" else StreetLearn
        cases.append(("%s%s_%s_i2" % (ds, typ, name), cls(datapath=proot, subepoch=0, streetlearn_interiornet_type=typ, reshape_size=[64, 80]), 2))
    for prefix, db, idx in cases:
        db.aug.draw = lambda prm=prm: dict(prm)
        sample = db[idx]
        _same_sample(ref_readers, prefix, sample)
        assert np.array_equal(sample[0].numpy(), ref_readers[prefix + "_images_u8"].astype(np.float32))
