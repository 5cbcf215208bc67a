#!/bin/bash
# round-2 GPU session A: MFMA ceiling + GEMM lab, PMC counters of the shipped kernels, counter list
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( cd /tmp && rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/rocprof_counters.txt 2>&1 )
timeout 600 tools/lab/gemm_lab rel_pose_amd/librelpose_hip.so > gpurun_out/lab_a.txt 2>&1
echo "lab rc=$?"
tail -5 gpurun_out/lab_a.txt
rm -f gpurun_out/pmc_r2base.csv
timeout 1200 tools/pmc.sh r2base $GRAFT_REPO_ROOT/tools/all_kernels.py 2
echo "pmc rc=$?"
wc -l gpurun_out/pmc_r2base.csv
