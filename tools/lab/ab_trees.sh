#!/bin/bash
# interleaved whole-step A/B of two source TREES (.ab_old/ = a built copy of an older commit, ./ = the working tree) inside one gpurun call
R=${1:-3}; shift
for i in $(seq $R); do
  for t in .ab_old .; do
    ms=$(cd $t && python bench.py --steps 30 --warmup 5 --no-supplementary --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "round $i  [$t]  $ms ms/step"
  done
done
