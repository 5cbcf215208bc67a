"""Shared reader of the two panorama-crop datasets (reference src/data_readers/interiornet.py:53-107, streetlearn.py:53-108):
`metadata/<set>[T]/train_pair_{rotation,translation}_overlap.npy` holds a pickled dict {i: {'img1': {path,x,y}, 'img2': ...}};
a tenth of it per sub-epoch (or the first 32000 pairs with use_mini_dataset); rotation-only ground truth, 256x256 crops."""
import os
import os.path as osp

import numpy as np

from .base import RGBDDataset
from .viewpoint import relative_quaternion


class PanoramaPairs(RGBDDataset):
    META = None            # metadata folder stem, e.g. "interiornet"
    DATA = None            # image folder under data/ for the rotation set
    DATA_T = None          # ... and for the translation ("T") set

    def _build_dataset(self, subepoch):
        scene_info = {"images": [], "poses": [], "intrinsics": []}
        base_pose = np.array([0, 0, 0, 0, 0, 0, 1])
        with_t = self.streetlearn_interiornet_type not in ("", None)
        path = ("metadata/%sT/train_pair_translation_overlap.npy" if with_t else "metadata/%s/train_pair_rotation_overlap.npy") % self.META
        folder = self.DATA_T if with_t else self.DATA
        split = np.array(np.load(osp.join(self.root, path), allow_pickle=True), ndmin=1)[0]
        size = len(split.keys()) // 10
        start, end = (0, 32000) if self.use_mini_dataset else (size * subepoch, size * (subepoch + 1))
        for i in split.keys():
            if i < start or i >= end:
                continue
            a, b = split[i]["img1"], split[i]["img2"]
            rotation = relative_quaternion(a["x"], a["y"], b["x"], b["y"])
            scene_info["images"].append([os.path.join(self.root, "data", folder, a["path"]),
                                         os.path.join(self.root, "data", folder, b["path"])])
            scene_info["poses"].append(np.vstack([base_pose, np.concatenate([np.zeros(3), rotation])]))   # translation is 0
            scene_info["intrinsics"].append(np.array([[128, 128, 128, 128], [128, 128, 128, 128]]))
        return scene_info
