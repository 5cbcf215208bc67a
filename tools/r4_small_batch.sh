#!/bin/bash
# round-4 evidence for the reference's own batch (6 pairs per GPU): eager / side stream / HIP-graph replay, and WHERE the replay loses
# (gaps between consecutive kernels under rocprofv3 --kernel-trace).  Run on the GPU box from the repo root.
O=gpurun_out
{
echo "# bench.py --batch 6 (12 images), fp32, 1 GPU: ms per step / pairs per second"
for cfg in "RP_X=eager" "RP_SIDE_STREAM=1"; do
  env $cfg python bench.py --steps 100 --warmup 10 --no-supplementary --no-cpu-baseline --batch 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$cfg', d['ms_per_step'], 'ms/step', d['value'], 'pairs/s')"
done
python bench.py --steps 100 --warmup 10 --no-supplementary --no-cpu-baseline --batch 6 --graph 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('--graph', d['ms_per_step'], 'ms/step', d['value'], 'pairs/s')"
cd /tmp && export TMPDIR=/tmp
for mode in "" "--graph"; do
  rm -rf /tmp/gp
  rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -o g -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --no-supplementary --no-cpu-baseline --batch 6 $mode > /dev/null 2>&1
  echo "# kernel gaps, bench.py --batch 6 $mode (second half of the trace)"
  python $GRAFT_REPO_ROOT/tools/gap_stats.py $(find /tmp/gp -name "*kernel_trace.csv" | head -1) 0.5
done
} > $GRAFT_REPO_ROOT/$O/r4_small_batch.txt 2>&1
cat $GRAFT_REPO_ROOT/$O/r4_small_batch.txt
