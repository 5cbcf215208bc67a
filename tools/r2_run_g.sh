#!/bin/bash
# full GPU test-suite, the bench line, steady-state kernel profiles (full + hot scope)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/g_tests.txt 2>&1
echo "gpu tests rc=$?"; tail -4 gpurun_out/g_tests.txt
timeout 900 python bench.py --steps 40 --warmup 5 > gpurun_out/g_bench.json 2> gpurun_out/g_bench.err
echo "bench rc=$?"; cut -c1-400 gpurun_out/g_bench.json
RP_GEMM_NO_DMA=1 timeout 900 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/g_bench_nodma.json 2>> gpurun_out/g_bench.err
cut -c1-200 gpurun_out/g_bench_nodma.json
timeout 900 tools/profile_step.sh g_hot 12 3 --steps 400 --warmup 3 --scope hot
python tools/show_stats.py gpurun_out/g_hot_kernel_stats.csv 0 30 > gpurun_out/g_hot_summary.txt; head -24 gpurun_out/g_hot_summary.txt
