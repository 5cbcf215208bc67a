#!/bin/bash
# in-box A/B of RP_MLP_DEBUG variants: three interleaved rounds of the probe (boxes of the pool differ by > 10 % on HBM-bound kernels)
export PYTHONPATH=$PWD
for r in 1 2 3; do for d in "$@"; do RP_MLP_DEBUG=$d python tools/lab/mlp_bf16_probe.py 2>&1 | grep RP_MLP; done; done
