"""Build librelpose_hip.so (gfx950) in-tree with hipcc.  No CPU fallback exists: if the build or the load
fails, every op in rel_pose_amd raises."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librelpose_hip.so")
SOURCES = ["gemm.hip", "gemm_dma.hip", "rowwise.hip", "attention.hip", "attention_bf16.hip", "dw192_bf16.hip", "dw192_f32.hip", "dw192_split3.hip", "dx_lnbwd_bf16.hip", "emm.hip", "emm_bf16.hip", "batchnorm.hip", "se3loss.hip", "geom.hip", "augment.hip", "mlp_fused.hip", "linear_rows.hip", "conv_stem.hip", "conv_stem_bf16.hip", "conv_stem_wgrad_bf16.hip", "conv_stem_wgrad_f32.hip", "conv3x3_bf16.hip", "conv3x3_wgrad_bf16.hip", "conv3x3_wgrad_f32.hip", "conv3x3_f32.hip", "conv3x3_c128_f32.hip"]
ARCH = "gfx950"


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + _headers()
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """Compile under an exclusive file lock (eight ranks of a first `torchrun` would otherwise write the same .o / .so at
    once) and move the finished library into place atomically, so a concurrent loader never maps a half-written file."""
    import fcntl
    with open(os.path.join(HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():      # another rank built it while this one waited
                return LIB
            return _build_locked(verbose, force)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _headers():
    return [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith(".h")] + \
           [os.path.join(os.path.dirname(HERE), "include", "relpose_hip.h")]


def _build_locked(verbose, force=False):
    """force: every translation unit is recompiled; otherwise only objects older than their source or any header."""
    cc = _hipcc()
    objs = []
    procs = []
    hdr_t = max(os.path.getmtime(h) for h in _headers())
    for s in SOURCES:
        o = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(o)
        if not force and os.path.exists(o) and os.path.getmtime(o) > max(hdr_t, os.path.getmtime(os.path.join(CSRC, s))):
            continue
        cmd = [cc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed on " + s)
    tmp = LIB + ".tmp.%d" % os.getpid()
    cmd = [cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", tmp] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
