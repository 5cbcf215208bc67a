#!/usr/bin/env python3
"""Derivation of common.h's gelu_fast / gelu_grad_fast constants: 1/2 erfc(t / sqrt 2) = exp2(-t q(t) - 1) on t = |x| in [0, 5.7],
q a degree-7 polynomial fitted (Chebyshev nodes, weighted by the effect of an error in q on Phi) to -log2(erfc(t / sqrt 2)) / t.
Prints the coefficients (constant term first) and the error of the fp32 Horner evaluation against float64."""
import numpy as np
from numpy.polynomial import chebyshev as Ch, polynomial as Pl
from scipy.special import erf, erfc

X0, DEG = 5.7, 7
t = np.maximum((np.cos(np.pi * (np.arange(8000) + 0.5) / 8000) + 1) / 2 * X0, 1e-9)
q = -np.log2(erfc(t / np.sqrt(2))) / t
coef = Ch.Chebyshev.fit(t, q, DEG, domain=[0, X0], w=erfc(t / np.sqrt(2)) * t).convert(kind=Pl.Polynomial).coef
print("coefficients c0..c7:", ", ".join("%.9ef" % np.float32(c) for c in coef))


def gelu_fast32(x, c):
    x = x.astype(np.float32)
    tt = np.minimum(np.abs(x), np.float32(X0))
    p = np.full_like(tt, np.float32(c[-1]))
    for k in c[-2::-1]:
        p = p * tt + np.float32(k)
    e = np.exp2((-(tt * p) - np.float32(1)).astype(np.float32)).astype(np.float32)
    return (x * np.where(x < 0, e, np.float32(1) - e)).astype(np.float32)


if __name__ == "__main__":
    x = np.linspace(-9, 9, 1800001).astype(np.float32).astype(np.float64)      # fp32-representable arguments
    ref = 0.5 * x * (1 + erf(x / np.sqrt(2)))
    err = np.abs(gelu_fast32(x, coef).astype(np.float64) - ref)
    print("max |gelu_fast - exact| over [-9, 9]: %.3e at x = %.3f; relative to |x|: %.3e" %
          (err.max(), x[err.argmax()], (err / np.maximum(np.abs(x), 1e-3)).max()))
