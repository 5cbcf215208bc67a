#!/usr/bin/env python3
"""train.py -- MI355X counterpart of the reference's trainer (reference train.py:28-212), same CLI flags.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI), batch-of-pairs data parallel through
DistributedDataParallel: the only exchange per step is the gradient all-reduce.  The model is the drop-in ViTEss of
rel_pose_amd (HIP hot path).  `--dataset matterport|interiornet|streetlearn --datapath ...` reads the reference's
dataset layouts through rel_pose_amd/data_readers (ten sub-epochs then a validation pass, reference train.py:108-131);
the default is the reference's `matterport`.  No dataset exists in the build environment: `--dataset synthetic` streams
seeded random pairs of the real tensor shapes instead, so the whole loop -- forward, geodesic loss, backward, clip, Adam,
OneCycle schedule, checkpoint save / auto-resume with the reference's file layout and keys -- runs anywhere.

    python train.py --name run0 --gpus 1 --batch 64 --steps 100 --fusion_transformer          # single GPU
    python train.py ... --precision bf16          # (not in the reference) BASELINE configs[4]: bf16 data path + bf16 CNN front-end
    python train.py --name run0 --gpus 8 ...      # spawns 8 ranks itself, like the reference (train.py:286-291, port 12356)
    python -m torch.distributed.run --nproc-per-node 8 train.py --name run0 --gpus 8 ...       # or under a launcher
"""
import argparse
import os
import time
from collections import OrderedDict
from datetime import datetime

import torch
import torch.distributed as dist

from rel_pose_amd import parallel
from rel_pose_amd.losses import geodesic_loss
from rel_pose_amd.model import ViTEss
from rel_pose_amd.se3 import SE3


class SyntheticPairs(torch.utils.data.Dataset):
    """(images [2,3,H,W] BGR 0..255, poses [2,7], intrinsics [2,4]) like RGBDDataset.__getitem__
    (reference src/data_readers/base.py:45-97; pose convention of matterport.py:44-54)."""

    def __init__(self, n, hw=(384, 512), seed=0):
        self.n, self.hw, self.seed = n, hw, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        H, W = self.hw
        images = torch.floor(torch.rand(2, 3, H, W, generator=g) * 255.0)
        q = torch.randn(4, generator=g)
        q = q / q.norm()
        if q[3] < 0:
            q = -q
        poses = torch.zeros(2, 7)
        poses[:, 6] = 1.0
        poses[1] = torch.cat([torch.rand(3, generator=g) * 2 - 1, q])
        intr = torch.tensor([[517.97, 517.97, 320.0, 240.0]] * 2) * torch.tensor([W / 640.0, H / 480.0, W / 640.0, H / 480.0])
        return images, poses, intr


def find_resume(name):
    """Auto-resume rule of the reference (train.py:255-275)."""
    d = "output/%s/checkpoints" % name
    if not os.path.isdir(d):
        return None
    ck = [f for f in os.listdir(d) if f.endswith(".pth")]
    if not ck:
        return None
    if "most_recent_ckpt.pth" in ck:
        return os.path.join(d, "most_recent_ckpt.pth")
    return os.path.join(d, "%06d.pth" % max(int(f[:-4]) for f in ck))


def load_checkpoint(path, map_location=None):
    """Checkpoints are trusted local files written by this train.py or by the reference's (torch 1.8, whose OneCycleLR
    state holds a bound method that the weights-only unpickler rejects) -> full unpickling, reference train.py:89."""
    return torch.load(path, map_location=map_location, weights_only=False)


def next_subepoch(subepoch, dataset):
    """Ten training sub-epochs, then one validation pass -- except InteriorNet / StreetLearn, which have no validation
    split (reference train.py:204-208)."""
    if dataset == "synthetic":
        return 0
    subepoch += 1
    if subepoch == 11 or (subepoch == 10 and dataset in ("interiornet", "streetlearn")):
        subepoch = 0
    return subepoch


def run(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ddp = world > 1 and not args.no_ddp
    if ddp:
        parallel.setup(rank, world, backend="nccl")
    torch.manual_seed(0)
    torch.backends.cudnn.benchmark = True      # MIOpen solver search for CNN shapes missing from rel_pose_amd/miopen_db (once)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from rel_pose_amd import ops
    prec = {"fp32": 0, "split3": 3, "bf16": 1}[getattr(args, "precision", "fp32")]
    ops.set_gemm_precision(prec)
    ops.set_attention_precision(1 if prec == 1 else 0)
    ops.set_cnn_precision(1 if prec == 1 else 0)
    model = ViTEss(args).to(dev).train()
    if rank == 0 and not model.resnet_pretrained and not (args.ckpt or find_resume(args.name)):
        print("WARNING: the ResNet-18 trunk starts from random weights -- the reference fine-tunes conv1/layer1/layer2 from "
              "torchvision's ImageNet weights (src/model.py:31).  Point RELPOSE_RESNET18_WEIGHTS (or --resnet_weights) at a "
              "local torchvision resnet18 state_dict to reproduce its training recipe.", flush=True)
    for p in list(model.resnet.layer3.parameters()) + list(model.resnet.layer4.parameters()):
        p.requires_grad = False
    net = parallel.wrap(model, [local]) if ddp else model
    opt = torch.optim.Adam(net.parameters(), lr=args.lr, weight_decay=args.weight_decay, fused=True)    # same update, one kernel
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, args.lr, args.steps, pct_start=min(0.99, args.warmup / args.steps),
                                                div_factor=25, cycle_momentum=False)
    resume = args.ckpt or find_resume(args.name)
    if resume:
        ck = load_checkpoint(resume, map_location=dev)
        sd = ck["model"]
        if not ddp:
            sd = OrderedDict((k.replace("module.", ""), v) for k, v in sd.items())
        net.load_state_dict(sd, strict=not args.ckpt)
        opt.load_state_dict(ck["optimizer"])
        resumed_step = 0
        if "scheduler" in ck and not args.ckpt:
            # keep THIS run's schedule (--steps may have grown) but continue from the saved position.  (The reference
            # restores the scheduler wholesale and restarts its step counter at 0, train.py:97-104,111.)
            st = sched.state_dict()
            for k in ("last_epoch", "_step_count"):
                st[k] = ck["scheduler"][k]
            sched.load_state_dict(st)
            resumed_step = int(sched.last_epoch)
        if rank == 0:
            print("resumed from", resume, "at step", resumed_step)

    def make_loader(subepoch):
        """(loader, sampler, is_training): the reference rebuilds its dataset every sub-epoch; the 11th is validation"""
        is_training = subepoch != 10
        if args.dataset == "synthetic":
            db = SyntheticPairs(args.batch * world * 50, tuple(args.image_size))
        else:
            from rel_pose_amd.data_readers.factory import dataset_factory
            db = dataset_factory([args.dataset], datapath=args.datapath, subepoch=subepoch, is_training=is_training, gpu=local,
                                 streetlearn_interiornet_type=args.streetlearn_interiornet_type,
                                 use_mini_dataset=args.use_mini_dataset, reshape_size=list(args.image_size),
                                 raw=device_augment)
        smp = (torch.utils.data.distributed.DistributedSampler(db, num_replicas=world, rank=rank, shuffle=is_training)
               if ddp else None)
        # every rank runs its own worker pool: cap it so that the ranks of this node together do not oversubscribe the host cores
        # (ranks on THIS node: under a multi-node launch the global world size would halve every rank's share)
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", min(world, torch.cuda.device_count() or 1))) if ddp else 1
        nw = parallel.loader_workers(args.num_workers, local_world)
        if nw < args.num_workers and rank == 0 and subepoch == 0:
            print("note: --num_workers %d capped to %d per rank (%d ranks share this host's cores; ~14 decode cores feed one GPU, "
                  "README.md)" % (args.num_workers, nw, local_world))
        ld = torch.utils.data.DataLoader(db, batch_size=args.batch, sampler=smp, shuffle=(smp is None and is_training),
                                         num_workers=nw, pin_memory=True, drop_last=is_training)
        return ld, smp, is_training

    # Input pipeline: with --device_augment (default on a GPU) the workers only decode; colour jitter, resize
    # and the intrinsics rescale run once per batch in rp_augment_pairs.  --no_device_augment = the reference's per-sample
    # CPU augmentation inside the workers (26 pairs/s per core here, profiles/r2_loader_bench.txt).
    device_augment = args.device_augment and dev.type == "cuda" and args.dataset != "synthetic"
    augmentor = None
    if device_augment:
        from rel_pose_amd.data_readers.augmentation import RGBDAugmentor
        augmentor = RGBDAugmentor(reshape_size=list(args.image_size))

    os.makedirs("output/%s/checkpoints" % args.name, exist_ok=True)
    step, t0 = (resumed_step if resume else 0), time.time()
    subepoch = 0
    while step < args.steps:
        loader, sampler, is_training = make_loader(subepoch)
        if sampler is not None:
            sampler.set_epoch(step)
        net.train(is_training)
        val = []
        for images, poses, intr in loader:
            images, poses, intr = images.to(dev, non_blocking=True), poses.to(dev), intr.to(dev)
            if augmentor is not None:
                images, intr = augmentor.augment_batch_hip(images, intr)
            Ps = SE3(poses)
            Gs = SE3.IdentityLike(Ps)
            if not is_training:                       # validation pass (reference train.py:147-150)
                with torch.no_grad():
                    est = net(images, Gs, intrinsics=intr)
                    val.append(geodesic_loss(Ps, est, train_val="val")[2])
                continue
            opt.zero_grad(set_to_none=True)
            est = net(images, Gs, intrinsics=intr)
            ltr, lrot, metrics = geodesic_loss(Ps, est)
            (args.w_tr * ltr + args.w_rot * lrot).backward()
            parallel.clip_grad_norm_(net.parameters(), args.clip)
            opt.step()
            sched.step()
            step += 1
            if rank == 0 and step % 20 == 0:
                print("step %6d  %s" % (step, metrics), flush=True)
            if rank == 0 and (step % 10000 == 0 or step >= args.steps):
                torch.save({"model": net.state_dict(), "optimizer": opt.state_dict(), "scheduler": sched.state_dict()},
                           "output/%s/checkpoints/%06d.pth" % (args.name, step))
            if step >= args.steps:
                break
        if val and rank == 0:
            print("validation  %s" % {k: sum(v[k] for v in val) / len(val) for k in val[0]}, flush=True)
        subepoch = next_subepoch(subepoch, args.dataset)
    if rank == 0:
        print("finished training!")
    if ddp:
        parallel.cleanup()


def parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--w_tr", type=float, default=10.0)
    ap.add_argument("--w_rot", type=float, default=10.0)
    ap.add_argument("--warmup", type=int, default=10000)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=120000)
    ap.add_argument("--lr", type=float, default=5e-4)
    ap.add_argument("--clip", type=float, default=2.5)
    ap.add_argument("--weight_decay", type=float, default=1e-5)
    ap.add_argument("--num_workers", type=int, default=4)
    ap.add_argument("--device_augment", dest="device_augment", action="store_true", default=True,
                    help="decode-only workers + fused colour jitter / resize on the GPU (default)")
    ap.add_argument("--no_device_augment", dest="device_augment", action="store_false",
                    help="the reference's per-sample CPU augmentation inside the DataLoader workers")
    ap.add_argument("--no_ddp", action="store_true", default=False)
    ap.add_argument("--gpus", type=int, default=4)              # reference train.py:229 (clamped to the visible GPUs, with a note)
    ap.add_argument("--ckpt", help="checkpoint to restore")
    ap.add_argument("--name", default="bla")
    ap.add_argument("--datapath")
    ap.add_argument("--image_size", default=[384, 512], nargs=2, type=int)
    ap.add_argument("--exp")
    ap.add_argument("--use_mini_dataset", action="store_true")
    ap.add_argument("--streetlearn_interiornet_type", default="", choices=("", "T"))
    ap.add_argument("--dataset", default="matterport", choices=("matterport", "interiornet", "streetlearn", "synthetic"))   # reference
    # train.py:238; "synthetic" (random 8-bit pairs of --image_size, no files) is this repo's addition for smoke runs / benchmarks
    for flag in ("no_pos_encoding", "noess", "cross_features", "use_single_softmax", "l1_pos_encoding", "fusion_transformer"):
        ap.add_argument("--" + flag, action="store_true")
    ap.add_argument("--fc_hidden_size", type=int, default=512)
    ap.add_argument("--pool_size", type=int, default=60)
    ap.add_argument("--transformer_depth", type=int, default=6)
    ap.add_argument("--resnet_weights", help="local torchvision resnet18 state_dict (.pth) for the pretrained trunk")
    ap.add_argument("--precision", default="fp32", choices=("fp32", "split3", "bf16"),
                    help="not in the reference: operand precision of the HIP hot path -- fp32 = exact fp32 MFMA (the parity path, default); "
                         "bf16 = BASELINE configs[4] (bf16 data path + bf16 CNN front-end, tanh-form GELU); split3 = fp32-grade on the bf16 pipe")
    return ap


if __name__ == "__main__":
    a = parser().parse_args()
    a.noess = "1" if a.noess else ""
    os.makedirs("output/%s" % a.name, exist_ok=True)
    with open("output/%s/args_%s.txt" % (a.name, datetime.now().strftime("%Y-%m-%d_%H-%M")), "w") as f:
        for k, v in vars(a).items():
            f.write("%s  %s\n" % (k, v))
    if a.resnet_weights:
        os.environ["RELPOSE_RESNET18_WEIGHTS"] = a.resnet_weights
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if "WORLD_SIZE" not in os.environ and not a.no_ddp and a.gpus > max(ndev, 1):
        # the reference would fail in mp.spawn here; a default of 4 on a smaller (or GPU-less) box is more useful clamped (deviation, stated)
        print("note: --gpus %d but %d GPU(s) visible: running %d rank(s)" % (a.gpus, ndev, max(ndev, 1)))
        a.gpus = max(ndev, 1)
    if "WORLD_SIZE" in os.environ or a.no_ddp or a.gpus <= 1:
        run(a)                      # under a launcher (one rank per process already), or a single GPU
    else:
        parallel.spawn(run, a.gpus, (a,))     # the reference's own process model: --gpus N spawns N ranks
