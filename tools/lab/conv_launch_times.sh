#!/bin/bash
# per-position duration of the own convolution launches inside the step (kernel trace): which of forward / +stats / +res / +bn launches cost what
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/clt
rocprofv3 --kernel-trace --output-format csv -d /tmp/clt -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-supplementary --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
rows = []
for f in glob.glob('/tmp/clt/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
for pat, per in (('conv3x3_c64_f32_kernel', 8), ('conv3x3_c128_f32_kernel', 7)):
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3 for r in rows if pat in r['Kernel_Name']]
    d = d[-per * 8:]                      # the last 8 steps
    pos = collections.defaultdict(list)
    for i, v in enumerate(d):
        pos[i % per].append(v)
    print(pat, "  ".join("#%d %.0f" % (k, sum(v) / len(v)) for k, v in sorted(pos.items())), " us  (position in the step: forward launches first, then the input gradients in reverse layer order)")
PY
