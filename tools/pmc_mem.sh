#!/bin/bash
# usage: tools/pmc_mem.sh <tag> <pattern> <python args...>
# Memory-path counters of the kernels whose name contains <pattern> (comma list): separate rocprofv3 --pmc passes, --kernel-trace only.
# (The TA_* and TCC_* groups abort under rocprofv3 on this image after minutes of hanging -- signal 6 -- and are left out.)
# -> gpurun_out/pmcmem_<tag>.txt: per kernel symbol the average of every counter per launch + duration
tag=$1; pat=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcm && mkdir -p /tmp/pmcm
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum"; do
  if [ -n "$PMC_ONLY" ] && ! echo " $PMC_ONLY " | grep -q " $i "; then i=$((i+1)); continue; fi
  timeout ${PMC_PASS_TIMEOUT:-100} rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmcm/p$i -o p -- python "$@" > /tmp/pmcm/log$i.txt 2>&1
  i=$((i+1))
done
python3 - "$tag" "$pat" <<'PY'
import csv, glob, os, sys, collections
tag, pats = sys.argv[1], sys.argv[2].split(',')
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
def short(k): return k.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
for d in sorted(glob.glob('/tmp/pmcm/p*')):
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
with open(os.path.join(root, 'gpurun_out', 'pmcmem_%s.txt' % tag), 'w') as f:
    for k in sorted(acc):
        f.write("%s  (%.1f us per launch)\n" % (k, dur[k][0] / max(dur[k][1], 1)))
        for n in sorted(acc[k]):
            v = acc[k][n]
            f.write("    %-44s %16.1f\n" % (n, v[0] / max(v[1], 1)))
print(open(os.path.join(root, 'gpurun_out', 'pmcmem_%s.txt' % tag)).read())
PY
