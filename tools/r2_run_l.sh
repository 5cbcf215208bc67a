#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "attention or emm" 2>&1 | tail -5
echo "--- 16x16x4 forward"; python tools/attn_time.py 2>&1 | grep -v amdgpu.ids
echo "--- 32x32x2 forward"; RP_ATTN_MFMA32=1 python tools/attn_time.py 2>&1 | grep -v amdgpu.ids | head -2
