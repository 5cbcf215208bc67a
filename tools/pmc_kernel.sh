#!/bin/bash
# usage: tools/pmc_kernel.sh <tag> <pattern[,pattern]> <python args...>
# SQ / LDS / memory counters of the kernels whose symbol contains <pattern>: separate rocprofv3 --pmc passes (--kernel-trace only, never
# with other trace domains), averaged per launch -> gpurun_out/pmck_<tag>.txt
tag=$1; pat=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmck && mkdir -p /tmp/pmck
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  timeout ${PMC_PASS_TIMEOUT:-120} rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmck/p$i -o p -- ${RUN:-python} "$@" > /tmp/pmck/log$i.txt 2>&1
  i=$((i+1))
done
python3 - "$tag" "$pat" <<'PY'
import csv, glob, os, sys, collections
tag, pats = sys.argv[1], sys.argv[2].split(',')
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
def short(k): return k.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
for d in sorted(glob.glob('/tmp/pmck/p*')):
    for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            if any(p in k for p in pats):
                dur[k][0] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3; dur[k][1] += 1
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            if any(p in k for p in pats):
                a = acc[k][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
root = os.environ.get('GRAFT_REPO_ROOT', '.')
os.makedirs(os.path.join(root, 'gpurun_out'), exist_ok=True)
with open(os.path.join(root, 'gpurun_out', 'pmck_%s.txt' % tag), 'w') as f:
    for k in sorted(acc):
        f.write("%s  (%.1f us per launch under the profiler, %d launches)\n" % (k, dur[k][0] / max(dur[k][1], 1), dur[k][1]))
        c = {n: v[0] / max(v[1], 1) for n, v in acc[k].items()}
        for n in sorted(c):
            f.write("    %-28s %16.1f\n" % (n, c[n]))
        wc = c.get('SQ_WAVE_CYCLES')
        if wc:
            f.write("    -- of wave cycles: wait_any %.3f  wait_inst_any %.3f  active_inst_any %.3f\n" % tuple(c.get(n, 0) / wc for n in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY')))
        if c.get('SQ_BUSY_CYCLES') and c.get('SQ_VALU_MFMA_BUSY_CYCLES'):
            f.write("    -- MFMA busy / (4 SIMD x busy cycles per SE-normalised) see README; raw ratio MFMA_BUSY/BUSY = %.3f\n" % (c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_BUSY_CYCLES']))
        if c.get('SQ_INSTS_MFMA'):
            f.write("    -- per MFMA: VALU %.2f  LDS %.2f  VMEM %.3f  SALU %.2f\n" % tuple(c.get(n, 0) / c['SQ_INSTS_MFMA'] for n in ('SQ_INSTS_VALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM', 'SQ_INSTS_SALU')))
        if c.get('SQ_LDS_IDX_ACTIVE'):
            f.write("    -- LDS bank-conflict rate %.3f\n" % (c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE']))
print(open(os.path.join(root, 'gpurun_out', 'pmck_%s.txt' % tag)).read())
PY
