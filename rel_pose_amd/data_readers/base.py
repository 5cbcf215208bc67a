"""RGBDDataset -- counterpart of reference src/data_readers/base.py:17-100 (same constructor, same sample tuple)."""
import numpy as np
import torch
import torch.utils.data as data

from .augmentation import RGBDAugmentor


def imread_bgr(path):
    """cv2.imread equivalent ([H,W,3] uint8, BGR) on PIL -- cv2 is not installed in this image"""
    from PIL import Image
    with Image.open(path) as im:
        return np.ascontiguousarray(np.asarray(im.convert("RGB"))[:, :, ::-1])


class RGBDDataset(data.Dataset):
    def __init__(self, name, datapath, reshape_size=[384, 512], subepoch=None, is_training=True, gpu=0,
                 streetlearn_interiornet_type=None, use_mini_dataset=False, raw=False, jitter=True):
        # raw=True (not in the reference): samples are (uint8 [2,H,W,3] BGR as decoded, poses, UNSCALED intrinsics); the colour
        # jitter + resize + intrinsics rescale then run once per batch on the GPU (RGBDAugmentor.augment_batch_hip).  The
        # reference's per-sample CPU augmentation feeds ~26 pairs/s per core (profiles/r2_loader_bench.txt).
        self.raw = raw
        self.root = datapath
        self.name = name
        self.streetlearn_interiornet_type = streetlearn_interiornet_type
        # jitter=False (not in the reference): colour jitter off -- what the parity fixtures of the reference's own readers are compared in
        self.aug = RGBDAugmentor(reshape_size=reshape_size, datapath=datapath, jitter=jitter)
        self.matterport = "matterport" in datapath
        if self.matterport:
            self.scene_info = self._build_dataset(subepoch == 10)          # sub-epoch 10 is the validation pass
        elif "StreetLearn" in self.name or "InteriorNet" in self.name:
            self.use_mini_dataset = use_mini_dataset
            self.scene_info = self._build_dataset(subepoch)
        else:
            raise ValueError("unknown dataset %r at %r (the reference drops into pdb here, base.py:37-39)" % (name, datapath))

    @staticmethod
    def image_read(image_file):
        return imread_bgr(image_file)

    def _load(self, index):
        files = self.scene_info["images"][index]
        if self.raw:
            images = torch.from_numpy(np.stack([self.__class__.image_read(f) for f in files]))
            poses = torch.from_numpy(np.stack(self.scene_info["poses"][index]).astype(np.float32))
            intrinsics = torch.from_numpy(np.stack(self.scene_info["intrinsics"][index]).astype(np.float32))
            return images, poses, intrinsics
        images = np.stack([self.__class__.image_read(f) for f in files]).astype(np.float32)
        images = torch.from_numpy(images).permute(0, 3, 1, 2)
        poses = torch.from_numpy(np.stack(self.scene_info["poses"][index]).astype(np.float32))
        intrinsics = torch.from_numpy(np.stack(self.scene_info["intrinsics"][index]).astype(np.float32))
        return self.aug(images, poses, intrinsics)

    def __getitem__(self, index):
        if self.matterport:
            return self._load(index)
        local_index = index                     # the panorama datasets skip unreadable samples (base.py:72-97)
        while True:
            try:
                return self._load(local_index)
            except (OSError, ValueError):
                local_index += 1
                if local_index >= len(self):
                    raise

    def __len__(self):
        return len(self.scene_info["images"])
