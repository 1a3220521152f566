#!/usr/bin/env python3
"""Fit behind gelu2_grad() in tokenreduction_amd/csrc/tr_common.h: d/dx [x Phi(x)] = 1/2 + o(x), o odd;
o(x) ~= x P(t), t = 2 x^2 / c^2 - 1, on |x| <= c (clamped beyond: o(c) is within 7e-5 of its limit 1/2 at c = 4.5).  P is a
degree-9 polynomial in the MONOMIAL basis of t (in [-1, 1]: no cancellation in fp32 Horner; in x^2 the same fit loses 2e-4).
Prints the coefficients and the fp32-Horner error against the exact-erf derivative."""
import numpy as np
from scipy.special import erf

C, DEG = 4.5, 9


def o(x):
    return 0.5 * erf(x / np.sqrt(2)) + x * np.exp(-0.5 * x * x) / np.sqrt(2 * np.pi)


x = np.cos(np.linspace(0, np.pi, 40001)) * C / 2 + C / 2
x = x[x > 1e-9]
t = 2 * x * x / (C * C) - 1
coef, *_ = np.linalg.lstsq(np.polynomial.chebyshev.chebvander(t, DEG) * x[:, None], o(x), rcond=None)
p = np.polynomial.chebyshev.cheb2poly(coef)
print("c =", C, " P(t) low -> high:", ", ".join(f"{v:.9e}f" for v in p))
xx = np.linspace(-12, 12, 2400001).astype(np.float32)
xc = np.clip(xx, -C, C).astype(np.float32)
tt = (xc * xc * np.float32(2 / (C * C)) - np.float32(1)).astype(np.float32)
acc = np.full_like(xx, np.float32(p[-1]))
for k in range(DEG - 1, -1, -1):
    acc = (acc * tt + np.float32(p[k])).astype(np.float32)
val = (acc * xc + np.float32(0.5)).astype(np.float32)
true = 0.5 + o(xx.astype(np.float64))
e = np.abs(val - true)
print("fp32 Horner: max |fit - exact| =", e.max(), "at x =", xx[e.argmax()])
