#!/bin/bash
# (historical: round-2/3 evidence recipe; the RP_ROWS_* / RP_FUSE_MLP* / RP_EMM_STATS_ONE_PASS switches it flips were retired in round 6 -- set the ops.* attribute instead)
# late round-3 evidence (run on the GPU box from the repo root; lands in gpurun_out/, copy what is judged into profiles/)
set -x
O=gpurun_out
timeout 300 bash tools/profile_step.sh r3_bf16_128pairs_full_step 9 3 --steps 600 --warmup 3 --precision bf16 --batch 128 --no-supplementary > /dev/null
python tools/show_stats.py $O/r3_bf16_128pairs_full_step_kernel_stats.csv 0 50 > $O/r3_bf16_128pairs_full_step_summary.txt
{
  echo "# tools/bf16_gemm_shapes.py -- per-launch time of a Block's Linear products at 128 pairs, bf16 configuration: register-staged rp_gemm (precision 1),"
  echo "# fp32 vs bf16 storage, and the row-resident bf16 form (ROWS ...)"
  timeout 200 python tools/bf16_gemm_shapes.py 2>&1 | grep -v amdgpu.ids
  echo; echo "# tools/lab/bw_probe.py -- plain fill / copy / read-sum bandwidth of the same chip"
  timeout 100 python tools/lab/bw_probe.py 2>&1 | grep -v amdgpu.ids
  echo; echo "# tools/dsmm_time.py -- rp_ds_matmul at 64 pairs: bf16 tiles vs fp32 tiles"
  RP_DSMM=32 timeout 100 python tools/dsmm_time.py x 2>&1 | grep -v amdgpu.ids
  echo; echo "# tools/lab/rows_probe [N] [ln] [bf16] [io_bf16] [act+pre] [M] -- shader-clock stamps per chunk of linear_rows_kernel"
  (cd tools/lab; for a in "576 1 0 0 0 73728" "768 1 0 0 1 73728" "576 1 1 0 0 147456" "768 1 1 2 1 147456"; do echo "== rows_probe $a"; timeout 60 ./rows_probe $a 2>&1 | grep -E "launches|grid|mean" | head -4; done)
} > $O/r3_bf16_rows_and_streams.txt
{
  echo "# tools/prof_ab.sh (kernel time per step from rocprofv3 --kernel-trace --stats, 64 pairs unless noted)"
  tools/prof_ab.sh "RP_EMM_STATS_ONE_PASS=0" attn_fwd_kernel\<3,colstats --steps 30 --warmup 3
  tools/prof_ab.sh "RP_EMM_STATS_ONE_PASS=1" attn_fwd_kernel\<3,colstats --steps 30 --warmup 3
  tools/prof_ab.sh "RP_FUSE_MLP_TRAIN=0" mlp_fused,linear_rows_kernel,gemm_dma_kernel\<0,\ 0 --steps 30 --warmup 3
  tools/prof_ab.sh "RP_FUSE_MLP_TRAIN=1" mlp_fused,linear_rows_kernel,gemm_dma_kernel\<0,\ 0 --steps 30 --warmup 3
  echo "# 6 pairs per GPU (12 images): two-wave vs one-wave attention workgroups (the launcher picks one-wave below 512 two-wave workgroups)"
  tools/prof_ab.sh "RP_ATTN_NW=2 RP_ATTN_FWD=22" attn_fwd_kernel\<,attn_bwd_dkdv --steps 60 --warmup 5 --batch 6
  tools/prof_ab.sh "RP_X=default" attn_fwd_kernel\<,attn_bwd_dkdv --steps 60 --warmup 5 --batch 6
  echo "# 16 pairs per GPU (32 images): the crossover"
  tools/prof_ab.sh "RP_ATTN_NW=1 RP_ATTN_FWD=12" attn_fwd_kernel\<,attn_bwd_dkdv --steps 40 --warmup 5 --batch 16
  tools/prof_ab.sh "RP_X=default" attn_fwd_kernel\<,attn_bwd_dkdv --steps 40 --warmup 5 --batch 16
} > $O/r3_late_fp32_ab.txt 2>&1
{
  echo "# bench.py ms_per_step against --steps / --warmup, fresh process each (the kernel timer is primed in the untimed priming step; RP_NO_TIMER=1 = no timer at all)"
  for cfg in "10 3" "20 5" "40 5" "100 20"; do set -- $cfg; python bench.py --steps $1 --warmup $2 --no-supplementary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('steps $1 warmup $2:', d['ms_per_step'], 'ms/step', d['value'], 'pairs/s')"; done
  RP_NO_TIMER=1 python bench.py --steps 20 --warmup 5 --no-supplementary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('RP_NO_TIMER=1 steps 20 warmup 5:', d['ms_per_step'], 'ms/step', d['value'], 'pairs/s')"
  echo "# tools/step_trace.py 40: GPU time of every step of a fresh process"
  timeout 200 python tools/step_trace.py 40 2>&1 | grep -v amdgpu.ids
  echo "# tools/host_phases.py: host enqueue time per phase at 6 pairs per GPU"
  timeout 200 python tools/host_phases.py 2>&1 | tail -1
  echo "# bench.py --batch 6: eager vs --graph"
  for g in "" "--graph"; do python bench.py --steps 100 --warmup 10 --no-supplementary --no-cpu-baseline --batch 6 $g 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('batch 6 $g', d['ms_per_step'], 'ms/step', d['value'], 'pairs/s')"; done
} > $O/r3_bench_hygiene.txt 2>&1
