#!/bin/bash
# round-4 evidence set (run on the GPU box from the repo root: `bash tools/r4_artifacts.sh A|B|C|D`; everything lands in gpurun_out/,
# the files that are judged are copied into profiles/ by hand)
set -x
O=gpurun_out
export PYTHONPATH=$PWD
case "$1" in
A)  # bench lines + step profiles
  timeout 500 python bench.py --steps 40 --warmup 5 > $O/r4_bench.json 2> $O/r4_bench.err
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r4_bench_driver_cmd.json 2>/dev/null
  timeout 300 bash tools/profile_step.sh r4_full_step 14 3 --steps 500 --warmup 3 --no-supplementary > /dev/null
  timeout 300 bash tools/profile_step.sh r4_hot_scope 8 3 --steps 500 --warmup 3 --scope hot --no-supplementary > /dev/null
  timeout 300 bash tools/profile_step.sh r4_bf16_128pairs_full_step 9 3 --steps 600 --warmup 3 --precision bf16 --batch 128 --no-supplementary > /dev/null
  python tools/show_stats.py $O/r4_full_step_kernel_stats.csv 0 60 > $O/r4_full_step_summary.txt
  python tools/show_stats.py $O/r4_hot_scope_kernel_stats.csv 0 45 > $O/r4_hot_scope_summary.txt
  python tools/show_stats.py $O/r4_bf16_128pairs_full_step_kernel_stats.csv 0 60 > $O/r4_bf16_128pairs_full_step_summary.txt
  ;;
B)  # PMC: traffic of the three operating points + the per-kernel counter tables (fp32 shapes at 64 pairs, bf16 shapes at 128 pairs)
  TRAFFIC_OUT=r4_traffic.json timeout 300 bash tools/pmc_bench.sh > $O/r4_traffic_summary.txt
  TRAFFIC_OUT=r4_traffic_fwd.json timeout 300 bash tools/pmc_bench.sh --mode fwd > $O/r4_traffic_fwd_summary.txt
  TRAFFIC_OUT=r4_traffic_bf16.json timeout 300 bash tools/pmc_bench.sh --precision bf16 --batch 128 > $O/r4_traffic_bf16_summary.txt
  timeout 400 bash tools/pmc2.sh r4_kernels $GRAFT_REPO_ROOT/tools/all_kernels.py 3 > /dev/null
  timeout 400 bash tools/pmc2.sh r4_kernels_bf16 $GRAFT_REPO_ROOT/tools/bf16_kernels.py all 3 > /dev/null
  cp $O/pmc_r4_kernels_summary.txt $O/r4_pmc_kernels_summary.txt; cp $O/pmc_r4_kernels.json $O/r4_pmc_kernels.json
  cp $O/pmc_r4_kernels_bf16_summary.txt $O/r4_pmc_kernels_bf16_summary.txt; cp $O/pmc_r4_kernels_bf16.json $O/r4_pmc_kernels_bf16.json
  ;;
C)  # other operating points, kernel timings, the test report
  for cfg in "soak300:--steps 300 --warmup 5" "bf16_128pairs:--steps 40 --warmup 5 --batch 128 --precision bf16" "fp32_128pairs:--steps 40 --warmup 5 --batch 128" \
             "fwd_only:--steps 100 --warmup 5 --mode fwd" "batch16:--steps 60 --warmup 5 --batch 16" "bf16_128pairs_round3_path:--steps 40 --warmup 5 --batch 128 --precision bf16"; do
    name=${cfg%%:*}; args=${cfg#*:}
    env $( [ $name = bf16_128pairs_round3_path ] && echo "RP_BF16_PATH=0 RP_DW192=0" || echo "RP_X=0" ) timeout 300 python bench.py $args --no-supplementary --no-cpu-baseline > $O/r4_bench_$name.json 2>/dev/null
  done
  { timeout 200 python tools/attn_bf16_time.py 256; timeout 200 python tools/emm_bf16_time.py 256; } 2>&1 | grep -v amdgpu.ids > $O/r4_bf16_attention_emm_times.txt
  bash tools/r4_small_batch.sh > /dev/null
  rm -f $O/test_report.txt
  timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/r4_pytest_tail.txt
  cp $O/test_report.txt $O/r4_test_report.txt
  ;;
D)  # memory-path counters of the fused bf16 MLP (bounded passes) and the bf16 soak
  ITERS=4 timeout 800 bash tools/pmc_mem.sh mlp "mlp_fused_kernel" $GRAFT_REPO_ROOT/tools/lab/mlp_bf16_probe.py > /dev/null
  cp $O/pmcmem_mlp.txt $O/r4_pmcmem_mlp_bf16.txt
  timeout 300 python bench.py --steps 300 --warmup 5 --batch 128 --precision bf16 --no-supplementary --no-cpu-baseline > $O/r4_bench_bf16_128pairs_soak300.json 2>/dev/null
  ;;
esac
