#!/bin/bash
# is the fixed cost of the fully unrolled convolution kernels the cold instruction cache?  SQC instruction-cache counters + SQ instruction-fetch
# wait of conv3x3_c128_f32_kernel / conv3x3_c64_f32_kernel at two sizes (separate --pmc passes, --kernel-trace only)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o -i "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_IFETCH\|SQC_INST[A-Z_]*" | sort -u | tr '\n' ' '
echo
for N in 16 128; do
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/icp
  N=$N timeout 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/icp -o p -- python $GRAFT_REPO_ROOT/tools/conv3x3_c128_time.py > /dev/null 2>&1
  python3 - "$N" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob('/tmp/icp/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'conv3x3_c128_f32' in k:
            a = acc['c128'][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for k, v in acc.items():
    print("N=%s %s: " % (sys.argv[1], k) + "  ".join("%s=%.0f" % (c, s / max(n, 1)) for c, (s, n) in sorted(v.items())) + "   (per launch, %d launches)" % max(n for _, n in v.values()))
PY
done
done
