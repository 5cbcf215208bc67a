#!/bin/bash
# round-3 evidence set (run on the GPU box from the repo root; everything lands in gpurun_out/, copy what is judged into profiles/)
set -x
timeout 400 python bench.py --steps 40 --warmup 5 > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err
timeout 200 python bench.py --steps 100 --warmup 5 --batch 6 --no-supplementary --no-cpu-baseline > gpurun_out/r3_bench_batch6.json 2>/dev/null
timeout 300 bash tools/profile_step.sh r3_full_step 14 3 --steps 500 --warmup 3 --no-supplementary > /dev/null
timeout 300 bash tools/profile_step.sh r3_hot_scope 8 3 --steps 500 --warmup 3 --scope hot --no-supplementary > /dev/null
python tools/show_stats.py gpurun_out/r3_full_step_kernel_stats.csv 0 60 > gpurun_out/r3_full_step_summary.txt
python tools/show_stats.py gpurun_out/r3_hot_scope_kernel_stats.csv 0 45 > gpurun_out/r3_hot_scope_summary.txt
TRAFFIC_OUT=r3_traffic.json timeout 300 bash tools/pmc_bench.sh > gpurun_out/r3_traffic_summary.txt
TRAFFIC_OUT=r3_traffic_fwd.json timeout 300 bash tools/pmc_bench.sh --mode fwd > gpurun_out/r3_traffic_fwd_summary.txt
TRAFFIC_OUT=r3_traffic_bf16.json timeout 300 bash tools/pmc_bench.sh --precision bf16 --batch 128 > gpurun_out/r3_traffic_bf16_summary.txt
