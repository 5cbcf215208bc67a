#!/usr/bin/env python3
"""demo.py -- predict the relative pose of one image pair (counterpart of reference demo.py:24-101, same flags).

    python demo.py --img1 a.png --img2 b.png --ckpt pretrained_models/matterport.pth

Checkpoints are the reference's own files ({'model': state_dict} with an optional 'module.' prefix); the pretrained
ones are not obtainable offline, so without --ckpt the model keeps its random initialisation (plumbing check only).
cv2 is absent in this image: PNGs are read with a tiny stdlib reader (8-bit RGB/RGBA, non-interlaced), channels
reordered to BGR like cv2.imread, alpha dropped.
"""
import argparse
import struct
import zlib
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from rel_pose_amd.model import ViTEss
from rel_pose_amd.se3 import SE3


def read_png_bgr(path):
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n", "not a PNG"
    pos, idat, hdr = 8, b"", None
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat += body
        pos += 12 + n
    w, h, depth, ctype, _, _, interlace = hdr
    assert depth == 8 and ctype in (2, 6) and interlace == 0, "only 8-bit RGB/RGBA non-interlaced PNGs"
    ch = 3 if ctype == 2 else 4
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + w * ch)
    out = np.zeros((h, w * ch), dtype=np.uint8)
    prev = np.zeros(w * ch, dtype=np.int32)
    for y in range(h):
        ft, line = raw[y, 0], raw[y, 1:].astype(np.int32)
        cur = np.zeros(w * ch, dtype=np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        else:                                   # Sub / Average / Paeth need the running left neighbour
            for i in range(w * ch):
                a = cur[i - ch] if i >= ch else 0
                b = prev[i]
                c = prev[i - ch] if i >= ch else 0
                if ft == 1:
                    p = a
                elif ft == 3:
                    p = (a + b) >> 1
                else:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (line[i] + p) & 255
        out[y] = cur
        prev = cur
    img = out.reshape(h, w, ch)[:, :, :3]
    return img[:, :, ::-1].copy()               # RGB -> BGR (cv2 order)


def load_pair(img1, img2, matterport):
    """[1,2,3,H,W] float32 BGR 0..255 on the CPU: reference demo.py:65-76 (cv2.imread order, nearest resize to 384x512 for the
    Matterport checkpoints, whose training data was resized that way)."""
    images = np.stack([read_png_bgr(img1), read_png_bgr(img2)]).astype(np.float32)
    images = torch.from_numpy(images).permute(0, 3, 1, 2)
    if matterport:
        images = F.interpolate(images, size=[384, 512])                                         # demo.py:72-73
    return images.unsqueeze(0)


def postprocess(raw7, matterport):
    """reference demo.py:86-92: undo the training-time depth scale and reorder the quaternion (yzxw -> xyzw) on Matterport."""
    preds = np.array(raw7, dtype=np.float32, copy=True)
    if matterport:
        preds[:3] = preds[:3] * 5                                                               # DEPTH_SCALE
        preds[3:] = np.array([raw7[4], raw7[5], raw7[3], raw7[6]])
    return preds


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--datapath"); ap.add_argument("--weights")
    ap.add_argument("--image_size", default=[384, 512])
    ap.add_argument("--img1"); ap.add_argument("--img2"); ap.add_argument("--ckpt", default="")
    for flag in ("no_pos_encoding", "noess", "cross_features", "use_single_softmax", "l1_pos_encoding"):
        ap.add_argument("--" + flag, action="store_true")
    ap.add_argument("--fc_hidden_size", type=int, default=512)
    ap.add_argument("--pool_size", type=int, default=60)
    ap.add_argument("--transformer_depth", type=int, default=6)
    args = ap.parse_args(argv)
    args.fusion_transformer = True
    args.noess = "1" if args.noess else ""
    print("predicting pose on %s and %s using model %s" % (args.img1, args.img2, args.ckpt or "<random init>"))
    matterport = "matterport" in args.ckpt or not args.ckpt
    intr = [[517.97, 517.97, 320, 240]] * 2 if matterport else [[128, 128, 128, 128]] * 2      # demo.py:52-55
    intrinsics = torch.tensor([intr], dtype=torch.float32).cuda()

    model = ViTEss(args)
    if args.ckpt:
        sd = OrderedDict((k.replace("module.", ""), v) for k, v in torch.load(args.ckpt, map_location="cpu", weights_only=False)["model"].items())
        model.load_state_dict(sd)
    model = model.cuda().eval()

    images = load_pair(args.img1, args.img2, matterport).cuda()
    Gs = SE3(torch.tensor([[[0, 0, 0, 0, 0, 0, 1.0]] * 2]).cuda())
    with torch.no_grad():
        est = model(images, Gs, intrinsics=intrinsics)
    preds = postprocess(est[0][0][1].data.cpu().numpy(), matterport)
    np.set_printoptions(suppress=True, precision=5)
    if matterport:
        print("predicted R&t, as quaternion, in format x,y,z,qx,qy,qz,qw:")
        print(preds)
    else:
        print("predicted R, as quaternion in format qx,qy,qz,qw")
        print(preds[3:])
    return preds


if __name__ == "__main__":
    main()
