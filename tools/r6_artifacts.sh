#!/bin/bash
# round-6 evidence set (run on the GPU box from the repo root: `bash tools/r6_artifacts.sh A|B|C`; everything lands in gpurun_out/,
# the files that are judged are copied into profiles/ by hand)
set -x
O=gpurun_out
export PYTHONPATH=$PWD
case "$1" in
A)  # bench lines + step profiles
  timeout 500 python bench.py --steps 40 --warmup 5 > $O/r6_bench.json 2> $O/r6_bench.err
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r6_bench_driver_cmd.json 2>/dev/null
  timeout 300 bash tools/profile_step.sh r6_full_step 14 3 --steps 500 --warmup 3 --no-supplementary > /dev/null
  timeout 300 bash tools/profile_step.sh r6_hot_scope 8 3 --steps 500 --warmup 3 --scope hot --no-supplementary > /dev/null
  timeout 300 bash tools/profile_step.sh r6_bf16_128pairs_full_step 9 3 --steps 600 --warmup 3 --precision bf16 --batch 128 --no-supplementary > /dev/null
  python tools/show_stats.py $O/r6_full_step_kernel_stats.csv 0 60 > $O/r6_full_step_summary.txt
  python tools/show_stats.py $O/r6_hot_scope_kernel_stats.csv 0 45 > $O/r6_hot_scope_summary.txt
  python tools/show_stats.py $O/r6_bf16_128pairs_full_step_kernel_stats.csv 0 70 > $O/r6_bf16_128pairs_full_step_summary.txt
  ;;
B)  # PMC: traffic of the three operating points + the per-kernel counter table
  TRAFFIC_OUT=r6_traffic.json timeout 300 bash tools/pmc_bench.sh > $O/r6_traffic_summary.txt
  TRAFFIC_OUT=r6_traffic_fwd.json timeout 300 bash tools/pmc_bench.sh --mode fwd > $O/r6_traffic_fwd_summary.txt
  TRAFFIC_OUT=r6_traffic_bf16.json timeout 300 bash tools/pmc_bench.sh --precision bf16 --batch 128 > $O/r6_traffic_bf16_summary.txt
  timeout 600 bash tools/pmc2.sh r6_kernels $GRAFT_REPO_ROOT/tools/all_kernels.py 3 > /dev/null
  timeout 600 bash tools/pmc2.sh r6_kernels_bf16 $GRAFT_REPO_ROOT/tools/bf16_kernels.py all 3 > /dev/null
  ;;
C)  # other operating points + the test report
  for cfg in "bf16_128pairs:--steps 40 --warmup 5 --batch 128 --precision bf16" "fwd_only:--steps 100 --warmup 5 --mode fwd" \
             "batch6:--steps 100 --warmup 10 --batch 6" "batch6_run2:--steps 100 --warmup 10 --batch 6" "batch6_run3:--steps 100 --warmup 10 --batch 6" \
             "batch6_eager:--steps 100 --warmup 10 --batch 6 --no-graph" "soak300:--steps 300 --warmup 5"; do
    name=${cfg%%:*}; args=${cfg#*:}
    timeout 300 python bench.py $args --no-supplementary --no-cpu-baseline > $O/r6_bench_$name.json 2>/dev/null
  done
  python tools/attn_p_time.py 2>&1 | grep -v amdgpu.ids > $O/r6_attn_stored_p_time.txt
  python tools/emm_s_time.py 2>&1 | grep -v amdgpu.ids > $O/r6_emm_stored_s_time.txt
  rm -f $O/test_report.txt
  timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/r6_pytest_tail.txt
  cp $O/test_report.txt $O/r6_test_report.txt
  ;;
esac
