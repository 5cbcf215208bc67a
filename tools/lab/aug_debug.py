import sys, tempfile, torch, numpy as np
sys.path.insert(0, '.')
import torch.nn.functional as F
from rel_pose_amd import ops
from rel_pose_amd.data_readers.augmentation import RGBDAugmentor
from rel_pose_amd.data_readers.matterport import Matterport
from tests import _eval_cases as EC
tmp = tempfile.mkdtemp(); mroot = tmp + "/matterport_fake"; EC.write_matterport_train(mroot)
mp = Matterport(datapath=mroot, subepoch=0, raw=True)
images, poses, intr = mp[3]
x = images.permute(0, 3, 1, 2).float()
def run(row, size=(96, 128)):
    row = torch.tensor(row, dtype=torch.float32)
    out = ops.augment_pairs(images[None].cuda(), row[None].cuda(), size[0], size[1])[0].cpu()
    ref = F.interpolate(RGBDAugmentor.apply(x, RGBDAugmentor.params_to_dict(row)), size=list(size))
    d = (out - ref).abs()
    print(row.tolist(), "mismatch px:", int((d > 0).sum()), "max", float(d.max()), "per-channel", [int((d[:, c] > 0).sum()) for c in range(3)])
run([3, 2, 0, 1, 0.8, 1.2, 0.8, -0.1, 0])
run([3, -1, -1, -1, 0.8, 1.2, 0.8, -0.1, 0])
run([3, 2, -1, -1, 0.8, 1.2, 0.8, -0.1, 0])
run([3, 2, 0, -1, 0.8, 1.2, 0.8, -0.1, 0])
run([-1, -1, -1, 1, 0.8, 1.2, 0.8, -0.1, 0])
run([0, -1, -1, 1, 0.8, 1.2, 0.8, -0.1, 0])
run([2, -1, -1, 1, 0.8, 1.2, 0.8, -0.1, 0])
run([3, -1, -1, 1, 0.8, 1.2, 0.8, -0.1, 0])
run([3, -1, -1, 1, 0.8, 1.2, 0.8, 0.05, 0])
run([1, 0, 3, 2, 1.1, 1.1, 1.25, 0.0, 1])
