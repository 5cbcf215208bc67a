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


_WARNED = False


def check_db(warn=True):
    """Does the shipped solver db belong to the MIOpen that is actually loaded?  The db file names carry MIOpen's version string
    (gfx950100.HIP.<major>_<minor>_<patch>_<build>.ufdb.txt); another MIOpen build looks for other names, finds nothing and silently
    falls back to its heuristic solvers (the 43.0 vs 41.3 ms gap of round 1).  Call after the first convolution has run.  Checks
    (a) the major_minor_patch of the shipped names against torch.backends.cudnn.version() (MIOpen's own number under ROCm) and
    (b) whether MIOpen has started db files under OTHER names in the user-db directory.  Returns a list of findings (empty = ok)
    and prints one warning per process."""
    global _WARNED
    import re
    findings = []
    if not os.path.isdir(MIOPEN_DB):
        return findings
    shipped = [f for f in os.listdir(MIOPEN_DB) if f.endswith(".txt")]
    vers = {m.group(1) for f in shipped for m in [re.search(r"\.HIP\.(\d+_\d+_\d+)_", f)] if m}
    try:
        import torch
        v = int(torch.backends.cudnn.version() or 0)
        have = "%d_%d_%d" % (v // 1000000, (v // 1000) % 1000, v % 1000) if v else None
    except Exception:
        have = None
    if have and vers and have not in vers:
        findings.append("shipped MIOpen solver db is for MIOpen %s, the loaded MIOpen reports %s" % (sorted(vers), have))
    dst = os.environ.get("MIOPEN_USER_DB_PATH")
    if dst and os.path.isdir(dst):
        known = {re.sub(r"^batchnorm_", "", f).rsplit(".", 2)[0] for f in shipped}
        other = sorted({f for f in os.listdir(dst) if f.endswith((".udb.txt", ".ufdb.txt"))
                        and re.sub(r"^batchnorm_", "", f).rsplit(".", 2)[0] not in known})
        if other:
            findings.append("MIOpen opened db files the shipped set does not contain: %s" % other[:3])
    if findings and warn and not _WARNED:
        _WARNED = True
        import warnings
        warnings.warn("rel_pose_amd: " + "; ".join(findings) + " -- the CNN front-end's convolutions run on MIOpen's heuristic solvers "
                      "(a few % slower); re-run tools/tune_miopen.sh on this ROCm to regenerate rel_pose_amd/miopen_db/", RuntimeWarning)
    return findings


setup()
