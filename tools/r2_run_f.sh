#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q > gpurun_out/f_kernels.txt 2>&1
echo "kernel tests rc=$?"; tail -5 gpurun_out/f_kernels.txt
timeout 900 python tools/gemm_tiles.py > gpurun_out/f_tiles_dma.txt 2>&1
RP_GEMM_NO_DMA=1 timeout 900 python tools/gemm_tiles.py qkv fc2 > gpurun_out/f_tiles_nodma.txt 2>&1
echo tiles done
