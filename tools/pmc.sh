#!/bin/bash
# usage: tools/pmc.sh <tag> <python args...>   -> gpurun_out/pmc_<tag>.csv (kernel, counter, value) for our kernels
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/pmc_out
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc_out -o p -- python "$@" > /dev/null 2>&1
  python - "$tag" "$grp" <<'PY'
import csv, sys, glob, collections
tag, grp = sys.argv[1], sys.argv[2]
files = glob.glob('/tmp/pmc_out/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: [0.0, 0])
for f in files:
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '')
        if not any(s in k for s in ('gemm_kernel', 'attn_', 'emm_', 'colsum', 'ln_')):
            continue
        key = (k.replace('(anonymous namespace)::', '')[:60], r['Counter_Name'])
        acc[key][0] += float(r['Counter_Value']); acc[key][1] += 1
import os
out = os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'gpurun_out', 'pmc_%s.csv' % tag)
with open(out, 'a') as fo:
    for (k, c), (v, n) in sorted(acc.items()):
        fo.write('%s,%s,%.1f,%d\n' % (k, c, v / max(n, 1), n))
PY
done
