"""Process environment for the CNN front-end (imports nothing heavy; import it BEFORE the first convolution runs).

The ResNet-18 / ResidualBlock front-end runs on MIOpen through PyTorch-ROCm.  MIOpen's immediate
mode picks convolution solvers from its find-db; for this model's fp32 channels-last configurations at 128 images the
shipped system db has no entries and the heuristic fallback is ~3 % slower on the whole step than what a search finds
(43.0 -> 41.3 ms, profiles/README.md).  rel_pose_amd/miopen_db/ holds the user find-db / perf-db written by one such search
(`MIOPEN_FIND_ENFORCE=4`, tools/tune_miopen.sh) on an MI355X with this image's MIOpen build; pointing MIOPEN_USER_DB_PATH at
it gives every fresh process the searched solvers with no search (MIOpen's immediate mode consults the user find-db).
Other batch sizes / resolutions are not in it and fall back to MIOpen's heuristics unless the caller turns on
torch.backends.cudnn.benchmark (train.py does; bench.py with RP_CUDNN_BENCHMARK=1).  A user-set MIOPEN_USER_DB_PATH wins."""
import os

MIOPEN_DB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")


def setup():
    if os.path.isdir(MIOPEN_DB) and os.access(MIOPEN_DB, os.W_OK):
        os.environ.setdefault("MIOPEN_USER_DB_PATH", MIOPEN_DB)


setup()
