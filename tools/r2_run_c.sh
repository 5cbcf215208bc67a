#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/peak_c.txt
rm -f $O
rocm-smi --showpower --showmaxpower --showclocks --showperflevel > gpurun_out/smi_idle.txt 2>&1
for k in 0 1 2; do
  for cfg in "1024 40" "1024 2000" "2048 2000" "4096 2000"; do
    tools/lab/peak_lab $k $cfg 5 0 >> $O
  done
  tools/lab/peak_lab $k 2048 2000 5 1 >> $O
done
# sample power / clocks while the fp32 ceiling kernel runs for a few seconds
( tools/lab/peak_lab 0 2048 2000 200 0 >> $O ) &
sleep 1.5
for i in 1 2 3; do rocm-smi --showpower --showclocks >> gpurun_out/smi_load.txt 2>&1; sleep 0.7; done
wait
# counters on the ceiling kernels
cd /tmp && export TMPDIR=/tmp
for k in 0 1; do
  rm -rf /tmp/pk
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d /tmp/pk -o p -- $GRAFT_REPO_ROOT/tools/lab/peak_lab $k 2048 2000 3 0 > /dev/null 2>&1
  python3 - $k <<'PY' >> $GRAFT_REPO_ROOT/$O
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob('/tmp/pk/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
dur = []
for f in glob.glob('/tmp/pk/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3)
m = {k: sum(v) / len(v) for k, v in acc.items()}
d = sum(dur) / max(len(dur), 1)
print("PMC kind %s: avg duration %.1f us; %s" % (sys.argv[1], d, ", ".join("%s=%.0f" % kv for kv in sorted(m.items()))))
if 'GRBM_GUI_ACTIVE' in m and d:
    print("   clock from GRBM_GUI_ACTIVE/8/duration = %.3f GHz; MFMA busy = %.3f of SIMD cycles" % (m['GRBM_GUI_ACTIVE'] / 8 / d * 1e-3, m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (m['GRBM_GUI_ACTIVE'] / 8 * 1024)))
PY
done
cat $GRAFT_REPO_ROOT/$O
