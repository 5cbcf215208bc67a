#!/bin/bash
# usage: tools/profile_step.sh <tag> <delay_s> <collect_s> [bench args...] -> gpurun_out/<tag>_kernel_stats.csv
# steady-state window only (rocprofv3 -P delay:collect:1), so MIOpen's first-use solver search is excluded
tag=$1; delay=$2; coll=$3; shift 3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats -P $delay:$coll:1 --output-format csv -d /tmp/prof_$tag -o $tag -- \
  python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench.log 2>&1
find /tmp/prof_$tag -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv \;
grep -h '"metric"' $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench.log | cut -c1-230
