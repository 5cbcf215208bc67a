#!/bin/bash
# usage: tools/ab_lib.sh [rounds] [bench args...] -- interleaved A/B of two BUILDS of the library in one gpurun call:
# rel_pose_amd/librelpose_hip_A.so and _B.so (git-ignored, built by hand before the call) are copied over the live library in turn.
R=${1:-3}; shift
L=rel_pose_amd/librelpose_hip.so
cp $L /tmp/keep.so
for i in $(seq $R); do
  for v in A B; do
    cp rel_pose_amd/librelpose_hip_$v.so $L
    ms=$(python bench.py --steps 30 --warmup 5 --no-supplementary --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['frac'])")
    echo "round $i  [$v]  $ms"
  done
done
cp /tmp/keep.so $L
