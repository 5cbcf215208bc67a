#!/bin/bash
# usage: tools/ab_bench.sh "<ENV_A>" "<ENV_B>" [rounds] [bench args...]  -- interleaved A/B of the full bench step in ONE gpurun call
# (the chip's clock drifts by ~3 % between a cold and a warm box: only interleaved rounds are comparable)
A="$1"; B="$2"; R=${3:-3}; shift 3
for i in $(seq $R); do
  for cfg in "$A" "$B"; do
    ms=$(env $cfg python bench.py --steps 30 --warmup 5 --no-supplementary --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "round $i  [$cfg]  $ms ms/step"
  done
done
