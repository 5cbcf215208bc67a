"""Matterport3D pairs (reference src/data_readers/matterport.py:13-66): `mp3d_planercnn_json/cached_set_{train,val}.json`,
poses rescaled by DEPTH_SCALE and stored (t, q xyzw) with w >= 0, slot 0 = identity."""
import json
import os
import os.path as osp

import numpy as np

from .base import RGBDDataset


class Matterport(RGBDDataset):
    DEPTH_SCALE = 5.0          # depths are scaled to balance the rotation and translation losses

    def __init__(self, mode="training", **kwargs):
        self.mode = mode
        super().__init__(name="Matterport", **kwargs)

    def _build_dataset(self, valid=False):
        scene_info = {"images": [], "poses": [], "intrinsics": []}
        base_pose = np.array([0, 0, 0, 0, 0, 0, 1])
        with open(osp.join(self.root, "mp3d_planercnn_json", "cached_set_val.json" if valid else "cached_set_train.json")) as f:
            split = json.load(f)
        for entry in split["data"]:
            images = [os.path.join(self.root, "/".join(str(entry[k]["file_name"]).split("/")[6:])) for k in ("0", "1")]
            rel = np.array(entry["rel_pose"]["position"] + entry["rel_pose"]["rotation"], dtype=np.float64)   # t, q wxyz
            rel[:3] /= Matterport.DEPTH_SCALE
            # "w first -> w last" by swapping slots 3 and 6 exactly as the reference does: (w,x,y,z) becomes (z,x,y,w), not
            # (x,y,z,w) -- checkpoints are trained in that convention and test_matterport.py / demo.py undo it by the same swap
            rel[3], rel[6] = rel[6], rel[3]
            if rel[6] < 0:
                rel[3:] *= -1
            scene_info["images"].append(images)
            scene_info["poses"].append(np.vstack([base_pose, rel]))
            scene_info["intrinsics"].append(np.array([[517.97, 517.97, 320, 240], [517.97, 517.97, 320, 240]]))   # 480x640
        return scene_info
