"""Closed-form (RNG-free) inputs for the evaluation-metric parity fixtures.

Shared by tests/golden/make_fixtures.py (which feeds them to the REAL reference's test_matterport.py /
test_streetlearn_interiornet.py in the build container and commits only what the reference produced) and by the tests
(which feed the same inputs to rel_pose_amd/evaluation.py and this repo's scripts).  Nothing here is reference code:
fake datasets in the reference's on-disk layout + hand-made prediction / ground-truth sets that hit the edge cases.
"""
import json
import os

import numpy as np
from scipy.spatial.transform import Rotation

from oracle import relpose_oracle as O


def _image_u8(h, w, key):
    u = O.hash_uniform(h * w * 3, key) * 0.5 + 0.5
    return np.floor(u * 256.0).clip(0, 255).astype(np.uint8).reshape(h, w, 3)


def _save_png(path, arr):
    from PIL import Image
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(arr).save(path)


# ---------------------------------------------------------------------------------------------------------------------
# fake datasets (reference layout: test_matterport.py:97-121, test_streetlearn_interiornet.py:158-201)
# ---------------------------------------------------------------------------------------------------------------------
def matterport_entries(n=5):
    data = []
    for i in range(n):
        names = ["/a/b/c/d/e/rgb/house%d/img_%d_%d.png" % (i, i, k) for k in range(2)]     # first 6 components are dropped
        q = Rotation.from_euler("xyz", [7.0 * i - 9.0, 11.0 - 6.0 * i, 3.0 + 2.0 * i], degrees=True).as_quat()    # xyzw
        wxyz = [float(q[3]), float(q[0]), float(q[1]), float(q[2])]
        if i % 2 == 1:
            wxyz = [-v for v in wxyz]                   # negative-w ground truth (test_matterport.py:140-145)
        data.append({"0": {"file_name": names[0]}, "1": {"file_name": names[1]},
                     "rel_pose": {"position": [0.4 * i - 0.7, 0.3 - 0.2 * i, 0.15 * i], "rotation": wxyz}})
    return data


def write_matterport(root, n=5, hw=(120, 160)):
    data = matterport_entries(n)
    for i, e in enumerate(data):
        for k in ("0", "1"):
            rel = "/".join(e[k]["file_name"].split("/")[6:])
            _save_png(os.path.join(root, rel), _image_u8(hw[0], hw[1], 1000 + 10 * i + int(k)))
    os.makedirs(os.path.join(root, "mp3d_planercnn_json"), exist_ok=True)
    for split in ("train", "val", "test"):
        with open(os.path.join(root, "mp3d_planercnn_json", "cached_set_%s.json" % split), "w") as f:
            json.dump({"data": data}, f)
    return data


def panorama_entries(n=6):
    split = {}
    for i in range(n):
        u = O.hash_uniform(4, 77 + i)
        x1, y1 = 0.5 * u[0], 3.0 * u[1]
        # even pairs: a small relative rotation (ground-truth angle < 45 degrees), odd pairs: anything
        x2, y2 = (x1 + 0.1 * u[2], y1 + 0.4 * u[3]) if i % 2 == 0 else (0.5 * u[2], 3.0 * u[3])
        split[i] = {"img1": {"path": "s%d/a.png" % i, "x": float(x1), "y": float(y1)},
                    "img2": {"path": "s%d/b.png" % i, "x": float(x2), "y": float(y2)}}
    return split


def write_panorama(root, dataset="interiornet", n=6, hw=(128, 128)):
    split = panorama_entries(n)
    for i, e in split.items():
        for j, k in enumerate(("img1", "img2")):
            _save_png(os.path.join(root, "data", dataset, e[k]["path"]), _image_u8(hw[0], hw[1], 2000 + 10 * i + j))
    os.makedirs(os.path.join(root, "metadata", dataset), exist_ok=True)
    np.save(os.path.join(root, "metadata", dataset, "test_pair_rotation.npy"), split, allow_pickle=True)
    return split


# ---------------------------------------------------------------------------------------------------------------------
# fake TRAINING datasets for the reader fixtures (reference layout: src/data_readers/matterport.py:31-36,
# interiornet.py:60-85, streetlearn.py:60-86)
# ---------------------------------------------------------------------------------------------------------------------
PANORAMA_TRAIN = {                 # (dataset, streetlearn_interiornet_type) -> (metadata folder, file name, image folder under data/)
    ("interiornet", ""): ("interiornet", "train_pair_rotation_overlap.npy", "interiornet"),
    ("interiornet", "T"): ("interiornetT", "train_pair_translation_overlap.npy", "interiornet"),
    ("streetlearn", ""): ("streetlearn", "train_pair_rotation_overlap.npy", "streetlearn"),
    ("streetlearn", "T"): ("streetlearnT", "train_pair_translation_overlap.npy", "streetlearn_2016"),
}
PANORAMA_TRAIN_PAIRS = 43          # 43 // 10 = 4 pairs per sub-epoch, 3 left over that no sub-epoch ever reads
PANORAMA_UNREADABLE = (5, 6, 13)   # pair 5: first image missing, pair 6: second image is not an image, pair 13: both missing


def panorama_train_entries(dataset, typ, n=PANORAMA_TRAIN_PAIRS):
    key = 300 + 17 * sorted(PANORAMA_TRAIN).index((dataset, typ))
    split = {}
    for i in range(n):
        u = O.hash_uniform(4, key + i)
        split[i] = {"img1": {"path": "scene%02d/%s_a.png" % (i // 4, i), "x": float(1.4 * u[0]), "y": float(3.1 * u[1])},
                    "img2": {"path": "scene%02d/%s_b.png" % (i // 4, i), "x": float(1.4 * u[2]), "y": float(3.1 * u[3])}}
    return split


def write_panorama_train(root, dataset, typ, n=PANORAMA_TRAIN_PAIRS, hw=(32, 40), images_upto=16):
    """metadata for n pairs, images for pairs 0 .. images_upto-1 except the PANORAMA_UNREADABLE ones"""
    meta, fname, folder = PANORAMA_TRAIN[(dataset, typ)]
    split = panorama_train_entries(dataset, typ, n)
    # image content is a function of the image FOLDER (interiornet's two sets share data/interiornet: whichever set is written, the
    # files are the same)
    key = 5000 + 1000 * sorted({v[2] for v in PANORAMA_TRAIN.values()}).index(folder)
    for i in range(min(n, images_upto)):
        for j, k in enumerate(("img1", "img2")):
            path = os.path.join(root, "data", folder, split[i][k]["path"])
            if i in (5, 13) and (j == 0 or i == 13):
                continue                                               # missing file
            if i == 6 and j == 1:
                os.makedirs(os.path.dirname(path), exist_ok=True)
                with open(path, "wb") as f:
                    f.write(b"not a png")                              # undecodable file
                continue
            _save_png(path, _image_u8(hw[0], hw[1], key + 10 * i + j))
    os.makedirs(os.path.join(root, "metadata", meta), exist_ok=True)
    np.save(os.path.join(root, "metadata", meta, fname), split, allow_pickle=True)
    return split


def write_matterport_train(root, n_train=7, n_val=3, hw=(48, 64)):
    """like write_matterport, with DIFFERENT train / val splits (sub-epoch 10 reads the val file, base.py:30)"""
    data = matterport_entries(n_train + n_val)
    for i, e in enumerate(data):
        for k in ("0", "1"):
            rel = "/".join(e[k]["file_name"].split("/")[6:])
            _save_png(os.path.join(root, rel), _image_u8(hw[0], hw[1], 1000 + 10 * i + int(k)))
    os.makedirs(os.path.join(root, "mp3d_planercnn_json"), exist_ok=True)
    for split, part in (("train", data[:n_train]), ("val", data[n_train:])):
        with open(os.path.join(root, "mp3d_planercnn_json", "cached_set_%s.json" % split), "w") as f:
            json.dump({"data": part}, f)
    return data


# a fixed ColorJitter / RandomGrayscale draw per case for the "fixed jitter" reader fixtures: order of the four ops
# (0 brightness, 1 contrast, 2 saturation, 3 hue), their factors, greyscale yes/no
FIXED_JITTER = {
    "j0": dict(order=[0, 1, 2, 3], b=1.2, c=0.8, s=1.15, h=0.06, gray=False),
    "j1": dict(order=[3, 2, 0, 1], b=0.8, c=1.2, s=0.8, h=-0.1, gray=False),
    "j2": dict(order=[1, 0, 3, 2], b=1.1, c=1.1, s=1.25, h=0.0, gray=True),
}


# ---------------------------------------------------------------------------------------------------------------------
# hand-made prediction sets for the metric functions themselves
# ---------------------------------------------------------------------------------------------------------------------
def _wxyz(r):
    q = r.as_quat()
    return np.array([q[3], q[0], q[1], q[2]])


def matterport_metric_cases():
    """name -> dict(pred_tran, pred_rot, gt_tran, gt_rot) for test_matterport.py:27-68 (quaternions wxyz, gt with w >= 0).
    Edge cases: a prediction whose quaternion is the negated one, rotation errors exactly at the 30-degree threshold (as close
    as the arithmetic gets), translation errors exactly 1.0, identical prediction (error 0 -> |dot| clips at 1), an even and
    an odd number of samples (median), a single sample."""
    cases = {}
    gt_ang = (0.0, 40.0, 90.0, 10.0, 135.0, 179.0)
    err = (5.0, 30.0, 45.0, 0.0, 29.999999, 30.000001)
    gt_r = [Rotation.from_euler("zyx", [a, 0.3 * a, -0.2 * a], degrees=True) for a in gt_ang]
    pr_r = [g * Rotation.from_euler("x", e, degrees=True) for g, e in zip(gt_r, err)]
    gt_q = np.stack([_wxyz(g) * (1.0 if _wxyz(g)[0] >= 0 else -1.0) for g in gt_r])
    pr_q = np.stack([_wxyz(p) for p in pr_r])
    pr_q[1] = -pr_q[1]
    gt_t = np.array([[1.0, 0, 0], [0, 2.0, 0], [0, 0, 3.0], [1.0, 1.0, 1.0], [-0.5, 0.25, 4.0], [0.0, 0.0, 0.0]])
    pr_t = gt_t + np.array([[0.5, 0, 0], [0, 0, 0], [0, 0, 2.0], [1.0, 0, 0], [0.6, 0.8, 0.0], [0.0, 0.0, 0.99999999]])
    cases["mixed6"] = dict(pred_tran=pr_t, pred_rot=pr_q, gt_tran=gt_t, gt_rot=gt_q)
    cases["odd5"] = {k: v[:5] for k, v in cases["mixed6"].items()}
    cases["single"] = {k: v[3:4] for k, v in cases["mixed6"].items()}
    # float32 predictions as the scripts produce them (model output .cpu().numpy()), float64 ground truth from the json
    cases["f32_preds"] = dict(pred_tran=pr_t.astype(np.float32), pred_rot=pr_q.astype(np.float32), gt_tran=gt_t, gt_rot=gt_q)
    # unnormalised prediction quaternions (|q| = 1 +- 1e-3: what normalize_preds leaves after fp32 rounding, exaggerated)
    s = 1.0 + 1e-3 * O.hash_uniform(6, 5)[:, None]
    cases["unnormalised"] = dict(pred_tran=pr_t, pred_rot=pr_q * s, gt_tran=gt_t, gt_rot=gt_q)
    # a ground-truth w one ulp above 1 (an identity rotation after a float round trip): the reference's magnitude column is nan
    gq2 = gt_q.copy()
    gq2[0] = [1.0000000000000002, 0.0, 0.0, 0.0]
    cases["gt_w_above_one"] = dict(pred_tran=pr_t, pred_rot=pr_q, gt_tran=gt_t, gt_rot=gq2)
    return cases


def panorama_metric_cases():
    """name -> dict(pred_rot, gt_rot) for test_streetlearn_interiornet.py:72-122 (quaternions xyzw).  Edge cases: ground-truth
    angles on both sides of 45 and 90 degrees, errors on both sides of the 10-degree count, an empty bucket (the reference skips
    it), negated quaternions, identical prediction."""
    cases = {}
    gt_ang = (10.0, 30.0, 44.999999, 45.000001, 60.0, 80.0, 89.999999, 90.000001, 120.0)
    err = (2.0, 15.0, 10.0, 9.999999, 10.000001, 30.0, 0.0, 1.0, 1.0)
    gt = [Rotation.from_euler("y", a, degrees=True) for a in gt_ang]
    pred = [Rotation.from_euler("x", e, degrees=True) * g for g, e in zip(gt, err)]
    gq = np.stack([g.as_quat() for g in gt])
    pq = np.stack([p.as_quat() for p in pred])
    pq[4] = -pq[4]
    gq[5] = -gq[5]
    cases["mixed9"] = dict(pred_rot=pq, gt_rot=gq)
    cases["large_only"] = dict(pred_rot=pq[:3], gt_rot=gq[:3])            # overlap_small bucket empty
    cases["small_only"] = dict(pred_rot=pq[3:7], gt_rot=gq[3:7])          # overlap_large bucket empty
    cases["none"] = dict(pred_rot=pq[7:], gt_rot=gq[7:])                  # both empty -> {}
    cases["f32_preds"] = dict(pred_rot=pq.astype(np.float32), gt_rot=gq)
    return cases
