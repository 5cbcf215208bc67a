#!/bin/bash
# round-2 evidence set: bench line (with cpu baseline), steady-state kernel stats (full + hot), per-kernel HBM traffic,
# SQ/GRBM/TCC counters of every hot kernel, GPU test report
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/test_report.txt
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/art_tests.txt 2>&1; tail -3 gpurun_out/art_tests.txt
timeout 900 python bench.py --steps 40 --warmup 5 > gpurun_out/art_bench.json 2> gpurun_out/art_bench.err; cut -c1-250 gpurun_out/art_bench.json
timeout 900 tools/profile_step.sh art_full 14 3 --steps 500 --warmup 3
python tools/show_stats.py gpurun_out/art_full_kernel_stats.csv 0 48 > gpurun_out/art_full_summary.txt; head -5 gpurun_out/art_full_summary.txt
timeout 900 tools/profile_step.sh art_hot 7 3 --steps 500 --warmup 3 --scope hot
python tools/show_stats.py gpurun_out/art_hot_kernel_stats.csv 0 40 > gpurun_out/art_hot_summary.txt; head -5 gpurun_out/art_hot_summary.txt
timeout 900 tools/pmc_bench.sh --scope hot > gpurun_out/art_traffic_summary.txt 2>&1; head -14 gpurun_out/art_traffic_summary.txt
timeout 1200 tools/pmc2.sh art $GRAFT_REPO_ROOT/tools/all_kernels.py 3 > /dev/null 2>&1; head -12 gpurun_out/pmc_art_summary.txt | cut -c1-200
