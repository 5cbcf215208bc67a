#!/bin/bash
# is 0.74 of the fp32 MFMA peak "the ceiling" of dw192_f32_kernel because of POWER (clock) or because of ISSUE slots?
# Side by side, on one box: tools/lab/peak_lab kind 1 (v_mfma_f32_16x16x4_f32 back to back, nothing else) and the shipped dw192_f32_kernel:
# sustained TF, shader clock (s_memtime for the lab; GRBM_GUI_ACTIVE / duration for both under the profiler), rocm-smi power / sclk samples
# taken every 100 ms WHILE each runs, and the SQ issue / wait counters of both.  -> gpurun_out/r5_ceiling.txt
cd "${GRAFT_REPO_ROOT:-.}"; export PYTHONPATH=$PWD
O=$PWD/gpurun_out/r5_ceiling.txt; mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/peak_lab tools/lab/peak_lab.hip
sample() {  # $1 = tag: power / clock samples until /tmp/stop_$1 exists
  while [ ! -e /tmp/stop_$1 ]; do
    /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ' ' | sed "s/^/[$1] /"; echo
    sleep 0.1
  done
}
{
echo "== A: peak_lab, v_mfma_f32_16x16x4_f32 only (8 accumulators per wave, 4 waves per CU), random data, ~6 s"
rm -f /tmp/stop_A; sample A > /tmp/smi_A.txt & 
/tmp/peak_lab 1 2048 20000 40
touch /tmp/stop_A; wait
echo "   rocm-smi while it ran (every 100 ms; first / median / last of $(wc -l < /tmp/smi_A.txt) samples):"
sed -n '3p' /tmp/smi_A.txt; sed -n "$(( $(wc -l < /tmp/smi_A.txt) / 2 ))p" /tmp/smi_A.txt; tail -1 /tmp/smi_A.txt
echo "== A0: the same with zero operands (no toggling in the multipliers)"
/tmp/peak_lab 1 2048 20000 10 1
echo "== B: dw192_f32_kernel in a loop (M = 73 728 x 768 x 192), ~6 s"
rm -f /tmp/stop_B; sample B > /tmp/smi_B.txt &
python tools/lab/dw_loop.py 30000
touch /tmp/stop_B; wait
echo "   rocm-smi while it ran (first / median / last of $(wc -l < /tmp/smi_B.txt) samples):"
sed -n '3p' /tmp/smi_B.txt; sed -n "$(( $(wc -l < /tmp/smi_B.txt) / 2 ))p" /tmp/smi_B.txt; tail -1 /tmp/smi_B.txt
echo "== counters (rocprofv3 --pmc, separate passes)"
} > $O 2>&1
RUN=env bash tools/pmc_kernel.sh r5_ceiling_lab peak /tmp/peak_lab 1 2048 4000 6 > /dev/null 2>&1
bash tools/pmc_kernel.sh r5_ceiling_dw dw192_f32 $PWD/tools/lab/dw_loop.py 200 > /dev/null 2>&1
cat gpurun_out/pmck_r5_ceiling_lab.txt gpurun_out/pmck_r5_ceiling_dw.txt >> $O
cat $O
