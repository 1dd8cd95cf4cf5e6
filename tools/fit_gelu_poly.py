"""Coefficients of gelu_poly2 (csrc/gemm.hip): 0.5 erf(x / sqrt 2) / x as a polynomial in x^2 on [0, X0], weighted for the absolute
error of x * Q(x^2); prints the float32 coefficients (low order first) and the error of the resulting GELU, evaluated in float32.
    python tools/fit_gelu_poly.py [X0=4.0] [degree=8]"""
import sys

import numpy as np
from numpy.polynomial import chebyshev as Ch
from numpy.polynomial import polynomial as P
from scipy.special import erf

X0 = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 8
x = np.linspace(1e-6, X0, 20001)
u = 2 * x * x / X0 ** 2 - 1
c = Ch.chebfit(u, 0.5 * erf(x / np.sqrt(2)) / x, deg, w=x)
pt, powu = np.zeros(1), np.array([1.0])
for ck in Ch.cheb2poly(c):
    pt = P.polyadd(pt, ck * powu)
    powu = P.polymul(powu, np.array([-1.0, 2 / X0 ** 2]))
pt = pt.astype(np.float32)
xs = np.concatenate([np.linspace(-8, 8, 400001), -np.logspace(1, 4, 2001), np.logspace(1, 4, 2001)]).astype(np.float32)
xc = np.clip(xs, -X0, X0)
t = xc * xc
q = np.full_like(t, pt[-1])
for ck in pt[-2::-1]:
    q = q * t + ck
g = np.maximum(xs, np.float32(-X0)) * (np.float32(0.5) + xc * q)      # the multiplicand is clamped below as well: the factor does not decay for x < -X0
ref = xs.astype(np.float64) * 0.5 * (1 + erf(xs.astype(np.float64) / np.sqrt(2)))
err = np.abs(g - ref)
inside = np.abs(xs) <= X0
print("coefficients (x^0 .. x^%d of x^2):" % deg, [float(v) for v in pt])
rel = err / np.maximum(np.abs(ref), 1e-30)
print(f"max |gelu error| over [-1e4, 1e4]: {err[xs <= X0].max():.3e} at x = {xs[xs <= X0][err[xs <= X0].argmax()]:.3f};  inside [-{X0}, {X0}]: {err[inside].max():.3e};  "
      f"x > {X0}: max relative error {rel[xs > X0].max():.3e}")
