"""Entry points on the GPU: train.py (synthetic data, few steps, checkpoint + auto-resume) and demo.py (reference's demo
images, random-init weights => plumbing only; the pretrained checkpoints are not obtainable offline)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(cmd, cwd):
    env = dict(os.environ, PYTHONPATH=ROOT)
    return subprocess.run([sys.executable] + cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=900)


def test_train_steps_checkpoint_and_resume(tmp_path):
    args = [os.path.join(ROOT, "train.py"), "--name", "t0", "--batch", "4", "--steps", "6", "--warmup", "2", "--fusion_transformer",
            "--image_size", "256", "320", "--num_workers", "0"]
    r = run(args, str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "finished training!" in r.stdout
    ck = tmp_path / "output" / "t0" / "checkpoints" / "000006.pth"
    assert ck.exists()
    import torch
    sd = torch.load(str(ck), map_location="cpu")
    assert set(sd) == {"model", "optimizer", "scheduler"} and len(sd["model"]) == 227
    # second invocation auto-resumes from the newest checkpoint (reference train.py:255-275)
    r2 = run(args[:6] + ["8"] + args[7:], str(tmp_path))
    assert r2.returncode == 0, r2.stderr[-2000:]
    assert "resumed from" in r2.stdout


def test_demo_runs_on_png_pair(tmp_path):
    import numpy as np
    import zlib, struct

    def write_png(path, arr):                       # 8-bit RGB, filter 0
        h, w, _ = arr.shape
        raw = b"".join(b"\x00" + arr[y].tobytes() for y in range(h))
        def chunk(t, d):
            return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
        with open(path, "wb") as f:
            f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                    chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))
    rng = np.random.default_rng(0)
    for n in ("a.png", "b.png"):
        write_png(str(tmp_path / n), rng.integers(0, 255, size=(480, 640, 3), dtype=np.uint8))
    r = run([os.path.join(ROOT, "demo.py"), "--img1", str(tmp_path / "a.png"), "--img2", str(tmp_path / "b.png")], str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "predicted R&t" in r.stdout
