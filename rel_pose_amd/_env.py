"""Process environment for the CNN front-end (imports nothing heavy; import it BEFORE the first convolution runs).

The ResNet-18 / ResidualBlock front-end runs its convolutions on MIOpen through PyTorch-ROCm.  MIOpen's
immediate mode picks convolution solvers from its find-db; for this model's fp32 channels-last configurations at 128 images
the shipped system db has no entries and the heuristic fallback is ~4 % slower on the whole step than what a search finds
(43.0 -> 41.3 ms, profiles/README.md).  rel_pose_amd/miopen_db/ holds the user find-db / perf-db written by one such search
(`MIOPEN_FIND_ENFORCE=4`, tools/tune_miopen.sh) on an MI355X with this image's MIOpen build, for 384x384 training steps at
1, 2, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96 and 128 pairs per GPU (~200 KB of text).  Each process gets a private
copy (MIOpen appends to its user db; one copy per local rank keeps the eight ranks of a node from sharing files and keeps the
repository clean) and MIOPEN_USER_DB_PATH points at it, so every fresh process uses the searched solvers with no search
(MIOpen's immediate mode consults the user find-db).  Other batch sizes are not in it and fall back to MIOpen's
heuristics unless the caller turns on torch.backends.cudnn.benchmark (train.py does; bench.py with RP_CUDNN_BENCHMARK=1).
A user-set MIOPEN_USER_DB_PATH wins."""
import os
import shutil
import tempfile

MIOPEN_DB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")


def setup():
    if "MIOPEN_USER_DB_PATH" in os.environ or not os.path.isdir(MIOPEN_DB):
        return
    rank = os.environ.get("LOCAL_RANK", "0")
    dst = os.path.join(tempfile.gettempdir(), "relpose_miopen_db_%d_%s" % (os.getuid(), rank))
    try:
        os.makedirs(dst, exist_ok=True)
        for f in os.listdir(MIOPEN_DB):
            if f.endswith(".txt") and not os.path.exists(os.path.join(dst, f)):
                shutil.copyfile(os.path.join(MIOPEN_DB, f), os.path.join(dst, f))
        os.environ["MIOPEN_USER_DB_PATH"] = dst
    except OSError:
        pass            # read-only temp dir: MIOpen falls back to its defaults


setup()
