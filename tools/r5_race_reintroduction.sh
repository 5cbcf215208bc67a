#!/bin/bash
# VERDICT r4 item 2, "done" criterion: re-introduce the round-4 race of rp_dx_lnbwd_bf16 (the barrier wait without lgkmcnt(0)) in the
# GPU box's scratch copy of the tree, rebuild, and show that the full-size VALUE test fails; then restore and show it passes.
# Run on the GPU box:  gpurun -- 'bash tools/r5_race_reintroduction.sh'   -> gpurun_out/r5_race_reintroduction.txt
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r5_race_reintroduction.txt
mkdir -p gpurun_out
SRC=rel_pose_amd/csrc/dx_lnbwd_bf16.hip
cp $SRC /tmp/dx_lnbwd_bf16.hip.orig
{
echo "== the barrier waits as shipped:"; grep -n "s_waitcnt vmcnt" $SRC
sed -i 's/s_waitcnt vmcnt(3) lgkmcnt(0)/s_waitcnt vmcnt(3)/; s/s_waitcnt vmcnt(0) lgkmcnt(0)/s_waitcnt vmcnt(0)/' $SRC
echo "== with the round-4 bug re-introduced:"; grep -n "s_waitcnt vmcnt" $SRC
python -m rel_pose_amd._build > /tmp/build_bug.log 2>&1 || { echo "BUILD FAILED"; tail -5 /tmp/build_bug.log; }
for i in 1 2 3; do
  echo "-- buggy build, run $i: value test at full size (expected: FAILED)"
  timeout 900 python -m pytest tests/test_gpu_bf16_fullsize.py -q -x -k dx_lnbwd 2>&1 | grep -E "passed|failed|AssertionError|fullsize_dx" | head -5
  echo "-- buggy build, run $i: the round-4 guard, bit-reproducibility at full size"
  timeout 900 python -m pytest tests/test_gpu_bf16_path.py -q -x -k reproducible 2>&1 | grep -E "passed|failed" | head -3
done
cp /tmp/dx_lnbwd_bf16.hip.orig $SRC
python -m rel_pose_amd._build > /tmp/build_ok.log 2>&1 || { echo "BUILD FAILED"; tail -5 /tmp/build_ok.log; }
echo "== restored:"; grep -n "s_waitcnt vmcnt" $SRC
echo "-- shipped build: value test at full size (expected: passed)"
timeout 900 python -m pytest tests/test_gpu_bf16_fullsize.py -q -x -k dx_lnbwd 2>&1 | grep -E "passed|failed" | head -3
} > $OUT 2>&1
cat $OUT
