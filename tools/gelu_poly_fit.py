#!/usr/bin/env python3
"""Coefficients and error of vv_common.h::gelu_poly2 (round 6): erf(z) / z on [0, 3.5] as a degree-12 polynomial in w = 2 z^2 / 3.5^2 - 1 (Chebyshev least squares at
Chebyshev nodes -> monomial basis), evaluated as the kernel does (fp32 Horner, clamp, gelu = x/2 + |x|/2 * z * Q(w)) against scipy's erf in double."""
import numpy as np
from numpy.polynomial import chebyshev as C
from scipy.special import erf

Z, DEG = 3.5, 12
u = np.cos(np.pi * (np.arange(8000) + 0.5) / 8000)
z = np.sqrt((u + 1) / 2) * Z
mono = C.cheb2poly(C.chebfit(u, np.where(z > 1e-12, erf(z) / np.maximum(z, 1e-12), 2 / np.sqrt(np.pi)), DEG))
print("coefficients (constant term first):", ", ".join(f"{float(np.float32(c))!r}f" for c in mono))
m32 = mono.astype(np.float32)
x = np.linspace(-12, 12, 3000001).astype(np.float32)
ax = np.abs(x)
zz = np.minimum(ax * np.float32(0.70710678118654752), np.float32(Z))
w = (zz * zz) * np.float32(2 / (Z * Z)) - np.float32(1)
q = np.full_like(w, m32[-1])
for k in range(DEG - 1, -1, -1):
    q = (q * w + m32[k]).astype(np.float32)
g = (np.float32(0.5) * x + (np.float32(0.5) * ax) * (zz * q)).astype(np.float32)
ref = 0.5 * x.astype(np.float64) * (1 + erf(x.astype(np.float64) / np.sqrt(2)))
err = np.abs(g - ref)
print(f"max |gelu_poly2 - gelu| over [-12, 12]: {err.max():.3e} at x = {x[err.argmax()]:.3f}")
