#!/bin/bash
# build the split3 weight-gradient kernel in its lab modes (csrc/dw192_split3.hip: SPLIT3_MODE) as small shared libraries
#   tools/lab/dw_split3_lab.sh build      (here, cross-compiled)       tools/lab/dw_split3_lab.sh run   (on the GPU box)
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
  for m in ${MODES:-0 1 2 3}; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSPLIT3_MODE=$m ${EXTRA} -shared rel_pose_amd/csrc/dw192_split3.hip rel_pose_amd/csrc/dw192_f32.hip -o tools/lab/libdw3_$m.so || exit 1
  done
  exit 0
fi
python - <<'PY'
import ctypes, glob, os, torch
Z = int(os.environ.get("Z", "128")); M = Z * 576
def timeit(fn, n=40, warm=8):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
libs = {os.path.basename(p)[6:-3]: ctypes.CDLL(os.path.abspath(p)) for p in sorted(glob.glob("tools/lab/libdw3_*.so"))}
for l in libs.values():
    l.rp_dw192_f32_workspace_bytes.restype = ctypes.c_size_t
st = torch.cuda.current_stream().cuda_stream
for N in (192, 576, 768):
    a = torch.randn(M, N, device="cuda"); b = torch.randn(M, 192, device="cuda")
    ref = (a.double().t() @ b.double())
    l0 = next(iter(libs.values()))
    nbytes = l0.rp_dw192_f32_workspace_bytes(M, N); sk = l0.rp_dw192_f32_splits(M, N)
    ws = torch.empty(nbytes // 4, device="cuda")
    call = lambda l, f: getattr(l, f)(ctypes.c_void_p(a.data_ptr()), N, ctypes.c_void_p(b.data_ptr()), M, N, ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(nbytes), ctypes.c_void_p(st))
    row = ["N=%d" % N]
    t0 = timeit(lambda: call(l0, "rp_dw192_f32"))
    row.append("exact %.1f us" % t0)
    for m, l in libs.items():
        t = timeit(lambda: call(l, "rp_dw192_split3"))
        out = ws.view(sk, N, 192).sum(0)
        err = float((out.double() - ref).abs().max() / ref.abs().max())
        row.append("mode %s: %.1f us (%.2fx) err %.1e" % (m, t, t0 / t, err))
    print("  ".join(row), flush=True)
PY
