#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/n_tests.txt 2>&1
echo "gpu tests rc=$?"; tail -6 gpurun_out/n_tests.txt
grep -i "bf16\|svd" gpurun_out/test_report.txt | tail -8
timeout 900 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/n_bench.json 2> gpurun_out/n_bench.err
cut -c1-330 gpurun_out/n_bench.json
timeout 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --precision bf16 --batch 128 > gpurun_out/n_bench_bf16_b128.json 2>> gpurun_out/n_bench.err
cut -c1-330 gpurun_out/n_bench_bf16_b128.json
timeout 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --batch 128 > gpurun_out/n_bench_fp32_b128.json 2>> gpurun_out/n_bench.err
cut -c1-330 gpurun_out/n_bench_fp32_b128.json
echo "--- attention/EMM kernel times fp32"; python tools/attn_time.py 2>&1 | grep -v amdgpu.ids
echo "--- bf16"; RP_ATTN_BF16=1 python tools/attn_time.py 2>&1 | grep -v amdgpu.ids
