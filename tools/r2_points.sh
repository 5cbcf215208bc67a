#!/bin/bash
# (historical: round-2/3 evidence recipe; the RP_ROWS_* / RP_FUSE_MLP* / RP_EMM_STATS_ONE_PASS switches it flips were retired in round 6 -- set the ops.* attribute instead)
# round-2 end state: the other operating points of bench.py and the per-kernel timing tools (all 1 GPU)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline"
$B --steps 300 --warmup 5 2>/dev/null | tail -1 > gpurun_out/pt_soak300.json
$B --steps 30 --warmup 5 --batch 128 2>/dev/null | tail -1 > gpurun_out/pt_fp32_128pairs.json
$B --steps 100 --warmup 10 --batch 128 --precision bf16 2>/dev/null | tail -1 > gpurun_out/pt_bf16_128pairs.json
$B --steps 100 --warmup 10 --batch 64 --precision bf16 2>/dev/null | tail -1 > gpurun_out/pt_bf16_64pairs.json
$B --steps 40 --warmup 5 --precision split3 2>/dev/null | tail -1 > gpurun_out/pt_split3.json
$B --steps 40 --warmup 5 --mode fwd 2>/dev/null | tail -1 > gpurun_out/pt_fwd_only.json
RP_FUSE_MLP=0 RP_ROWS_LINEAR=0 $B --steps 40 --warmup 5 --mode fwd 2>/dev/null | tail -1 > gpurun_out/pt_fwd_only_round1_kernels.json
RP_FUSE_MLP_BWD=0 RP_ROWS_LINEAR=0 RP_ROWS_DX=0 $B --steps 40 --warmup 5 2>/dev/null | tail -1 > gpurun_out/pt_train_without_row_resident_kernels.json
$B --steps 100 --warmup 5 --batch 6 2>/dev/null | tail -1 > gpurun_out/pt_batch6.json
for f in gpurun_out/pt_*.json; do echo "$f: $(python -c "import json,sys; d=json.load(open('$f')); print(d['value'], 'pairs/s', d['ms_per_step'], 'ms/step')")"; done
{ python tools/rows_time.py; python tools/dx_time.py; python tools/mlp_bwd_time.py; python tools/mlp_time.py; } 2>&1 | grep -v amdgpu.ids > gpurun_out/pt_kernel_times.txt
cat gpurun_out/pt_kernel_times.txt
timeout 120 tools/lab/rows_probe 576 1 > gpurun_out/pt_rows_probe.txt 2>&1; grep -E "mean|launches" gpurun_out/pt_rows_probe.txt | head -4
timeout 300 tools/lab/chain_lab > gpurun_out/pt_chain_lab.txt 2>&1; tail -3 gpurun_out/pt_chain_lab.txt
