#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q > gpurun_out/o_tests.txt 2>&1
echo "gpu tests rc=$?"; tail -4 gpurun_out/o_tests.txt
timeout 900 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/o_bench.json 2> gpurun_out/o_bench.err
cut -c1-330 gpurun_out/o_bench.json; python -c "
import json; d=json.loads(open('gpurun_out/o_bench.json').read().strip().splitlines()[-1]); print(d['roofline'])"
timeout 900 python bench.py --steps 200 --warmup 5 --no-cpu-baseline --scope hot > gpurun_out/o_bench_hot.json 2>> gpurun_out/o_bench.err
cut -c1-330 gpurun_out/o_bench_hot.json
