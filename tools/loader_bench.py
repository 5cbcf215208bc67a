#!/usr/bin/env python3
"""Input-pipeline throughput: can the readers feed the GPUs at the rate bench.py consumes pairs?

Builds a fake Matterport tree of 640x480 PNG pairs (random content = worst case for the PNG decoder) and measures
  (a) the reference-style path: DataLoader workers that decode, colour-jitter and resize on the CPU (one sample at a time),
  (b) the device-side augmentor: RGBDAugmentor.augment_batch on a resident [64,2,3,480,640] batch (what train.py can use
      instead of the per-sample CPU jitter),
and prints pairs/s next to the consumption rate of one MI355X and of an 8-GPU node."""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image

CONSUME = 1767.0          # pairs/s per GPU, profiles/r2_bench.json


def build(root, n):
    rng = np.random.default_rng(0)
    data = []
    for i in range(n):
        names = []
        for k in range(2):
            rel = "rgb/house%d/img_%d_%d.png" % (i % 7, i, k)
            os.makedirs(os.path.dirname(os.path.join(root, rel)), exist_ok=True)
            # smooth gradients + noise: compresses like a photograph rather than like white noise
            base = np.linspace(0, 255, 640, dtype=np.float32)[None, :, None] * np.ones((480, 1, 3), np.float32)
            img = np.clip(base * rng.random() + rng.normal(0, 12, (480, 640, 3)), 0, 255).astype(np.uint8)
            Image.fromarray(img).save(os.path.join(root, rel))
            names.append("/a/b/c/d/e/" + rel)
        data.append({"0": {"file_name": names[0]}, "1": {"file_name": names[1]},
                     "rel_pose": {"position": [1.0, -2.0, 0.5], "rotation": [1.0, 0.0, 0.0, 0.0]}})
    os.makedirs(os.path.join(root, "mp3d_planercnn_json"), exist_ok=True)
    for split in ("train", "val"):
        with open(os.path.join(root, "mp3d_planercnn_json", "cached_set_%s.json" % split), "w") as f:
            json.dump({"data": data}, f)


def drain(ld, fn=None):
    """pairs/s between the arrival of the first batch and the last (worker start-up excluded) and including start-up"""
    t_start, t_first, cnt, first_cnt = time.time(), None, 0, 0
    for im, po, it in ld:
        if fn is not None:
            fn(im, it)
        cnt += im.shape[0]
        if t_first is None:
            t_first, first_cnt = time.time(), cnt
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    t_end = time.time()
    return (cnt - first_cnt) / max(t_end - t_first, 1e-9), cnt / (t_end - t_start)


def main():
    from rel_pose_amd.data_readers.factory import dataset_factory
    from rel_pose_amd.data_readers.augmentation import RGBDAugmentor
    n = 512
    root = os.path.join(tempfile.mkdtemp(), "matterport_fake")
    t0 = time.time()
    build(root, n)
    print("built %d fake 640x480 pairs in %.1f s" % (n, time.time() - t0), flush=True)
    db = dataset_factory(["matterport"], datapath=root, subepoch=0, is_training=True, gpu=0, reshape_size=[384, 512])
    torch.set_num_threads(1)
    from rel_pose_amd.data_readers.base import imread_bgr
    inner = db.datasets[0] if hasattr(db, "datasets") else db
    files = [f for pair in inner.scene_info["images"][:32] for f in pair]
    t0 = time.time()
    for f in files:
        imread_bgr(f)
    dec = (time.time() - t0) / len(files) * 2
    t0 = time.time()
    for i in range(32):
        db[i]
    one = (time.time() - t0) / 32
    print("one core: PNG decode of a pair %.1f ms; decode + colour jitter + resize (the reference's per-sample CPU path) %.1f ms "
          "= %.1f pairs/s per core" % (dec * 1e3, one * 1e3, 1.0 / one), flush=True)
    try:
        quota = open("/sys/fs/cgroup/cpu.max").read().strip()
    except OSError:
        quota = "n/a"
    print("host: os.cpu_count %d, sched_getaffinity %d, cgroup cpu.max %s" % (os.cpu_count(), len(os.sched_getaffinity(0)), quota), flush=True)
    big = torch.utils.data.ConcatDataset([db] * 16)             # 8192 samples = 128 batches of 64
    for workers in (16, 32, 64):
        ld = torch.utils.data.DataLoader(big, batch_size=64, num_workers=workers, shuffle=False, drop_last=True, pin_memory=False)
        rate, incl = drain(ld)
        print("DataLoader, %3d workers (CPU decode + jitter + resize per sample): %8.1f pairs/s steady (%.1f incl. worker start-up) "
              "= %.2f x one GPU, %.2f x an 8-GPU node" % (workers, rate, incl, rate / CONSUME, rate / (8 * CONSUME)), flush=True)
    if torch.cuda.is_available():
        aug = RGBDAugmentor(reshape_size=[384, 512], datapath=root)
        imgs = torch.floor(torch.rand(64, 2, 3, 480, 640, device="cuda") * 255.0)
        intr = torch.tensor([[517.97, 517.97, 320.0, 240.0]], device="cuda").repeat(64, 2, 1)
        for _ in range(2):
            aug.augment_batch(imgs, intr.clone())
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(5):
            aug.augment_batch(imgs, intr.clone())
        torch.cuda.synchronize()
        dt = (time.time() - t0) / 5
        print("device-side augment_batch (64 pairs resident on the GPU): %.1f ms per batch = %.0f pairs/s = %.1f x the GPU's own "
              "consumption rate (%.1f %% of a 36.2 ms step)" % (dt * 1e3, 64 / dt, 64 / dt / CONSUME, 100 * dt / 0.0362), flush=True)
        # (c) the GPU-rate path: decode-only workers (raw=True) -> uint8 batches -> upload -> rp_augment_pairs
        u8 = torch.randint(0, 256, (64, 2, 480, 640, 3), dtype=torch.uint8, device="cuda")
        for _ in range(3):
            aug.augment_batch_hip(u8, intr.clone())
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(50):
            aug.augment_batch_hip(u8, intr.clone())
        torch.cuda.synchronize()
        dt = (time.time() - t0) / 50
        nbytes = 64 * 2 * (480 * 640 * 3 * 0.64 + 384 * 512 * 3 * 4)      # nearest resize touches 0.64 of the source rows/cols
        print("rp_augment_pairs (fused HIP jitter + resize, 64 uint8 pairs resident): %.3f ms per batch = %.0f pairs/s "
              "(%.2f %% of a 36.2 ms step; ~%.0f GB/s of source + output traffic)" % (dt * 1e3, 64 / dt, 100 * dt / 0.0362, nbytes / dt / 1e9),
              flush=True)
        raw = dataset_factory(["matterport"], datapath=root, subepoch=0, is_training=True, gpu=0, reshape_size=[384, 512], raw=True)
        big = torch.utils.data.ConcatDataset([raw] * 32)           # 16384 samples = 256 batches

        def consume(im, it):
            aug.augment_batch_hip(im.cuda(non_blocking=True), it.cuda(non_blocking=True))

        for workers in (16, 32, 64, 128):
            ld = torch.utils.data.DataLoader(big, batch_size=64, num_workers=workers, shuffle=False, drop_last=True, pin_memory=True,
                                             prefetch_factor=2)
            rate, incl = drain(ld, consume)
            print("DataLoader, %3d decode-only workers -> upload -> rp_augment_pairs: %8.1f pairs/s steady (%.1f incl. start-up) "
                  "= %.2f x one GPU, %.2f x an 8-GPU node" % (workers, rate, incl, rate / CONSUME, rate / (8 * CONSUME)), flush=True)
        # in-process thread pool: PIL's PNG decoder releases the GIL, so threads decode in parallel with no IPC at all
        from concurrent.futures import ThreadPoolExecutor
        inner = raw.datasets[0]
        for threads in (16, 32, 64, 128):
            with ThreadPoolExecutor(threads) as ex:
                t0 = time.time()
                nb = 64
                for bi in range(nb):
                    items = list(ex.map(inner.__getitem__, [(bi * 64 + k) % len(inner) for k in range(64)]))
                    im = torch.stack([x[0] for x in items]).pin_memory()
                    it = torch.stack([x[2] for x in items])
                    consume(im, it)
                torch.cuda.synchronize()
                rate = nb * 64 / (time.time() - t0)
            print("in-process pool, %3d decode threads -> upload -> rp_augment_pairs: %8.1f pairs/s = %.2f x one GPU" %
                  (threads, rate, rate / CONSUME), flush=True)


if __name__ == "__main__":
    main()
