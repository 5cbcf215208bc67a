#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 tools/lab/gemm_lab rel_pose_amd/librelpose_hip.so > gpurun_out/lab_$1.txt 2>&1
echo "lab rc=$?"
tail -3 gpurun_out/lab_$1.txt
