#!/usr/bin/env python3
"""Evaluate a checkpoint on the Matterport3D test split -- counterpart of reference test_matterport.py:70-164, same flags and
output files (output/<exp>/matterport_test/{results.txt, gt_*_magnitude_vs_error.csv}).  Needs the dataset and a checkpoint
(neither exists in the build container); the metric code lives in rel_pose_amd/evaluation.py and is unit-tested."""
import argparse
import json
import os
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from rel_pose_amd import evaluation as E
from rel_pose_amd.data_readers.base import imread_bgr
from rel_pose_amd.model import ViTEss
from rel_pose_amd.se3 import SE3


def model_parser(parser):
    for flag in ("no_pos_encoding", "noess", "cross_features", "use_single_softmax", "l1_pos_encoding", "fusion_transformer"):
        parser.add_argument("--" + flag, action="store_true")
    parser.add_argument("--fc_hidden_size", type=int, default=512)
    parser.add_argument("--pool_size", type=int, default=60)
    parser.add_argument("--transformer_depth", type=int, default=6)
    return parser


def load_model(args):
    args.noess = "1" if args.noess else ""
    model = ViTEss(args)
    if args.ckpt:
        sd = torch.load(args.ckpt, map_location="cpu", weights_only=False)["model"]
        model.load_state_dict(OrderedDict((k.replace("module.", ""), v) for k, v in sd.items()))
    return model.cuda().eval()


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--datapath")
    parser.add_argument("--weights")
    parser.add_argument("--image_size", default=[384, 512])
    parser.add_argument("--exp", default="eval")
    parser.add_argument("--ckpt")
    parser.add_argument("--gamma", type=float, default=0.9)
    parser.add_argument("--limit", type=int, default=0, help="evaluate only the first N pairs (0 = all)")
    args = model_parser(parser).parse_args(argv)
    with open(os.path.join(args.datapath, "mp3d_planercnn_json/cached_set_test.json")) as f:
        dset = json.load(f)["data"]
    if args.limit:
        dset = dset[:args.limit]
    out_dir = os.path.join("output", args.exp, "matterport_test")
    print("performing evaluation on matterport_test set using model %s" % args.ckpt)
    model = load_model(args)
    pt, pr, gt_t, gt_r, raw = [], [], [], [], []
    Gs = SE3(torch.tensor([[0, 0, 0, 0, 0, 0, 1.0]] * 2).unsqueeze(0).cuda())
    for entry in dset:
        names = [os.path.join(args.datapath, "/".join(str(entry[k]["file_name"]).split("/")[6:])) for k in ("0", "1")]
        images = torch.from_numpy(np.stack([imread_bgr(n) for n in names]).astype(np.float32)).permute(0, 3, 1, 2)
        images = F.interpolate(images, size=[384, 512]).unsqueeze(0).cuda()
        intrinsics = torch.tensor([[[517.97, 517.97, 320, 240]] * 2], dtype=torch.float32).cuda()
        with torch.no_grad():
            est = model(images, Gs, intrinsics=intrinsics)
        raw.append(est[0][0][1].data.cpu().numpy())
        t, q = E.matterport_prediction(raw[-1])
        pt.append(t)
        pr.append(q)
        gt_t.append(entry["rel_pose"]["position"])
        gt_r.append(E.matterport_gt_rotation(entry["rel_pose"]["rotation"]))
    metrics = E.camera_metrics_matterport(pt, pr, gt_t, gt_r, out_dir)
    with open(os.path.join(out_dir, "results.txt"), "w") as f:
        for k, v in metrics.items():
            print(k, v)
            print(k, v, file=f)
    return metrics, {"raw": raw, "pred_tran": pt, "pred_rot": pr, "gt_tran": gt_t, "gt_rot": gt_r, "out_dir": out_dir}


if __name__ == "__main__":
    main()
