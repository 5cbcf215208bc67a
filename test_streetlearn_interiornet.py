#!/usr/bin/env python3
"""Evaluate a checkpoint on the InteriorNet / StreetLearn test pairs -- counterpart of reference
test_streetlearn_interiornet.py:130-244, same flags and output files (first 1000 pairs in key order)."""
import argparse
import os

import numpy as np
import torch

from rel_pose_amd import evaluation as E
from rel_pose_amd.data_readers.base import imread_bgr
from rel_pose_amd.data_readers.viewpoint import relative_quaternion
from rel_pose_amd.se3 import SE3
from test_matterport import load_model, model_parser


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--datapath")
    parser.add_argument("--weights")
    parser.add_argument("--image_size", default=[384, 512])
    parser.add_argument("--exp", default="eval")
    parser.add_argument("--ckpt")
    parser.add_argument("--dataset", default="interiornet", choices=("interiornet", "streetlearn"))
    parser.add_argument("--gamma", type=float, default=0.9)
    parser.add_argument("--streetlearn_interiornet_type", default="", choices=("", "nooverlap", "T", "nooverlapT"))
    args = model_parser(parser).parse_args(argv)
    with_t = args.streetlearn_interiornet_type == "T"
    meta = "metadata/%s%s/test_pair_%s.npy" % (args.dataset, "T" if with_t else "", "translation" if with_t else "rotation")
    out_name = "%s%s_test" % (args.dataset, "T" if with_t else "")
    folder = "streetlearn_2016" if (args.dataset == "streetlearn" and with_t) else args.dataset
    dset = np.array(np.load(os.path.join(args.datapath, meta), allow_pickle=True), ndmin=1)[0]
    out_dir = os.path.join("output", args.exp, out_name)
    print("performing evaluation on %s set using model %s" % (out_name, args.ckpt))
    model = load_model(args)
    Gs = SE3(torch.tensor([[0, 0, 0, 0, 0, 0, 1.0]] * 2).unsqueeze(0).cuda())
    pred_q, gt_q, raw = [], [], []
    for i, item in sorted(dset.items())[:1000]:
        a, b = item["img1"], item["img2"]
        images = np.stack([imread_bgr(os.path.join(args.datapath, "data", folder, a["path"])),
                           imread_bgr(os.path.join(args.datapath, "data", folder, b["path"]))]).astype(np.float32)
        images = torch.from_numpy(images).permute(0, 3, 1, 2).unsqueeze(0).cuda()
        intrinsics = torch.full((1, 2, 4), 128.0).cuda()
        with torch.no_grad():
            est = model(images, Gs, intrinsics=intrinsics)
        raw.append(est[0][0][1].data.cpu().numpy())
        pred_q.append(raw[-1][3:])
        gt_q.append(relative_quaternion(a["x"], a["y"], b["x"], b["y"]))
    metrics = E.rotation_metrics_panorama(pred_q, gt_q, out_dir)
    with open(os.path.join(out_dir, "results.txt"), "w") as f:
        for k, v in metrics.items():
            print(k, v)
            print(k, v, file=f)
    return metrics, {"raw": raw, "pred_rot": pred_q, "gt_rot": gt_q, "out_dir": out_dir}


if __name__ == "__main__":
    main()
