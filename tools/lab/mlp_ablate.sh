#!/bin/bash
# lab: what bounds the bf16 fused MLP kernels?  Builds of the library with parts of mlp_fused.hip's bf16 chunk loop removed (wrong results,
# same MFMA count) are swapped in and tools/lab/mlp_bf16_probe.py times forward (training form) and backward at 128 pairs:
# S = hidden-tensor stores, H = the h store alone, G = GELU / GELU', D = weight-stage DMA, L = weight reads from LDS; 1 = present.
L=rel_pose_amd/librelpose_hip.so; cp $L /tmp/keep.so
for r in 1 2; do for n in S1H1G1D1L1 S0H1G1D1L1 S1H0G1D1L1 S1H1G0D1L1 S1H1G1D0L1 S1H1G1D1L0 S0H1G0D0L0; do
  cp rel_pose_amd/librelpose_hip_$n.so $L; echo -n "$n  "; python tools/lab/mlp_bf16_probe.py 2>/dev/null | cut -c19-
done; done
cp /tmp/keep.so $L
