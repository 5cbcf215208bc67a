#!/bin/bash
# interleaved whole-step A/B of two environments with LONG runs (the power state of the box settles over ~100 steps)
A="$1"; B="$2"; R=${3:-2}; S=${4:-150}
for i in $(seq $R); do
  for cfg in "$A" "$B"; do
    env $cfg python bench.py --steps $S --warmup 10 --no-supplementary --no-cpu-baseline --timer-instance attn_fwd_savep 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('round $i [$cfg] %d steps: %.3f ms/step' % (d['steps'], d['ms_per_step']))"
  done
done
