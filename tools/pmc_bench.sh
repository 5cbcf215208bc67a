#!/bin/bash
# HBM traffic of our kernels over a bench run: two separate --pmc passes (FETCH_SIZE, WRITE_SIZE), as
# MI355X_MICROARCH.md prescribes (TCC slots: they do not fit one pass).  FETCH_SIZE is doubled (gfx950 counts 128-B
# requests at 64 B).  Writes gpurun_out/traffic.json: {kernel: {launches, fetch_bytes, write_bytes, hbm_bytes_per_launch}}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- \
    python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-supplementary --steps 3 --warmup 1 "$@" > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, json, os, collections
acc = collections.defaultdict(lambda: {"launches": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    n = collections.Counter()
    for f in glob.glob('/tmp/pmc_%s/**/*counter_collection.csv' % c, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('rpgemm::', '').replace('void ', '').split('(')[0]
            if not any(s in k for s in ('gemm_', 'attn_', 'emm_', 'colsum', 'ln_', 'splitk', 'tokens', 'rowdot', 'mlp_', 'linear_', 'ds_matmul', 'dw192', 'conv3x3', 'conv_stem', 'dx_lnbwd')):
                continue
            acc[k][c] += float(r['Counter_Value']); n[k] += 1
    for k, v in n.items():
        acc[k]["launches"] = v
out = {}
for k, v in acc.items():
    L = max(v["launches"], 1)
    fb, wb = 2.0 * v["FETCH_SIZE"] * 1024 / L, v["WRITE_SIZE"] * 1024 / L
    out[k] = {"launches": v["launches"], "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
              "hbm_bytes_per_launch": fb + wb}
import hashlib
root = os.environ.get('GRAFT_REPO_ROOT', '.')
hsh = hashlib.sha256()
for fn in sorted(glob.glob(os.path.join(root, 'rel_pose_amd', 'csrc', '*.hip')) + glob.glob(os.path.join(root, 'rel_pose_amd', 'csrc', '*.h'))):
    hsh.update(open(fn, 'rb').read())
p = os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'gpurun_out', os.environ.get('TRAFFIC_OUT', 'traffic.json'))
json.dump({"csrc_sha16": hsh.hexdigest()[:16], "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB units, FETCH x2 per MI355X_MICROARCH.md; "
                   "bench.py --steps 3 --warmup 1 " + " ".join(os.sys.argv[1:]), "kernels": out}, open(p, 'w'), indent=1, sort_keys=True)
for k in sorted(out, key=lambda k: -out[k]["hbm_bytes_per_launch"] * out[k]["launches"])[:12]:
    print("%-40s launches=%4d  fetch=%8.1f MB write=%8.1f MB per launch" % (k[:40], out[k]["launches"], out[k]["fetch_bytes_per_launch"] / 1e6, out[k]["write_bytes_per_launch"] / 1e6))
PY
