#!/bin/bash
# lab: where do the non-MFMA cycles of dw192_f32_kernel go?  Builds of the library with parts of the stage loop removed (wrong results,
# same MFMA count) are swapped in and the kernel timed alone (tools/lab/dw_loop.py): L = operand reads from LDS in the k-loop,
# D = the next stage's DMA issue, B = the per-stage wait + barrier.  The variant libraries are built by hand (not committed).
L=rel_pose_amd/librelpose_hip.so; cp $L /tmp/keep.so
for r in 1 2; do for n in L1D1B1 L0D1B1 L1D0B1 L0D0B1 L0D0B0; do
  cp rel_pose_amd/librelpose_hip_$n.so $L; echo -n "$n  "; python tools/lab/dw_loop.py 3000 | cut -c1-110
done; done
cp /tmp/keep.so $L
