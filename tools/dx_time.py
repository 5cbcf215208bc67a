#!/usr/bin/env python3
"""Input-gradient GEMMs that contract over 192 (fc2 with GELU' + column sums, attention proj): row-resident kernel vs rp_gemm."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rel_pose_amd import ops
from tools.rows_time import timeit

M = 64 * 2 * 576
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)          # noqa: E731
dy, aux, w2, wp = r(M, 192), r(M, 768), r(192, 768) * 768 ** -0.5, r(192, 192) * 192 ** -0.5
for name, new, old, fl in (
        ("fc2 dX (GELU' + colsum)", lambda: ops.linear_dx(dy, w2, dact=1, aux=aux, want_colsum=True),
         lambda: ops.gemm(dy, w2, M, 768, 192, b_layout=1, dact=1, aux=aux, want_colsum=True), 2.0 * M * 192 * 768),
        ("proj dX", lambda: ops.linear_dx(dy, wp), lambda: ops.gemm(dy, wp, M, 192, 192, b_layout=1), 2.0 * M * 192 * 192)):
    tn, to = timeit(new), timeit(old)
    print("%-26s rows kernel (+ transpose%s) %7.1f us (%.1f TF) | rp_gemm %7.1f us (%.1f TF)" %
          (name, ", colsum" if "colsum" in name else "", tn, fl / tn / 1e6, to, fl / to / 1e6), flush=True)
