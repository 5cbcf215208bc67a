#!/bin/bash
# sample socket power and shader clock while the bench step runs (is the step power-bound?)   usage: tools/lab/power_probe.sh [env...]
env "$@" python bench.py --steps 400 --warmup 5 --no-supplementary --no-cpu-baseline > /tmp/bench_pp.json 2>/dev/null &
BP=$!
sleep 25
for i in $(seq 12); do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|Temperature \(Sensor (edge|junction|hotspot)" | tr '\n' ' ' | sed 's/GPU\[0\]//g;s/\s\+/ /g'
  echo
  sleep 0.7
done
wait $BP
python -c "import json; d=json.loads(open('/tmp/bench_pp.json').readline()); print('ms_per_step', d['ms_per_step'])"
