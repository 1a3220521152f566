#!/usr/bin/env python3
"""Minimax fit behind gelu2() in tokenreduction_amd/csrc/tr_gemm.hip:
Phi(x) ~= sigmoid(a1 x + a3 x^3 + a5 x^5) on [-8, 8]; prints the coefficients and max |x*sigmoid(.) - gelu_erf(x)|."""
import numpy as np
from scipy.optimize import minimize
from scipy.special import erf

x = np.linspace(-8, 8, 400001)
g = x * 0.5 * (1 + erf(x / np.sqrt(2)))


def err(p):
    z = np.clip(p[0] * x + p[1] * x ** 3 + p[2] * x ** 5, -80, 80)
    return np.max(np.abs(x / (1 + np.exp(-z)) - g))


r = minimize(err, [1.5976, 0.07056, 0.0], method="Nelder-Mead", options=dict(xatol=1e-9, fatol=1e-12, maxiter=20000))
print("a1, a3, a5 =", r.x, " max abs err =", r.fun)
