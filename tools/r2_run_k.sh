#!/bin/bash
# per-dispatch kernel trace of the hot scope (steady state) -> per-(kernel, grid) average durations
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace -P 7:2:1 --output-format csv -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 400 --warmup 3 --scope hot > $GRAFT_REPO_ROOT/gpurun_out/k_bench.log 2>&1
python3 - <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/k_trace_summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('/tmp/kt/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('rpgemm::', '').replace('void ', '').split('(')[0]
        key = (n[:48], r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Grid_Size_Z', ''))
        acc[key][0] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3
        acc[key][1] += 1
tot = sum(v[0] for v in acc.values())
steps = None
for k, v in acc.items():
    if k[0].startswith('attn_fwd_kernel<2, false'): steps = v[1] / 5.0
steps = steps or 1
print("total %.1f us/step over %.1f steps" % (tot / steps, steps))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print("%-50s grid %8s z %3s  calls/step %5.1f  avg %8.1f us  /step %8.1f us" % (k[0], k[1], k[2], v[1] / steps, v[0] / v[1], v[0] / steps))
PY
head -60 $GRAFT_REPO_ROOT/gpurun_out/k_trace_summary.txt
