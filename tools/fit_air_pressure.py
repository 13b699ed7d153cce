#!/usr/bin/env python3
"""Coefficients of the air-pressure polynomial of the raw-driver forms (mod16_capi.hip, kPressurePoly):
MOD16.air_pressure (reference mod16/__init__.py:414-447), 101325 (1 - 0.0065 z / 288.15)^5.2559,
interpolated at the Chebyshev nodes of [-2000 m, 12000 m] and converted to powers of
u = (z - 5000) / 7000. Prints the hex coefficients and the worst relative error of a float64
Horner evaluation on the interval."""
import numpy as np
from numpy.polynomial import chebyshev as C

LO, HI, DEG = -2000.0, 12000.0, 9
f = lambda z: 101325.0 * (1.0 - 0.0065 * z / 288.15) ** (9.80665 / (0.0065 * (8.3143 / 28.9644e-3)))
mid, half = 0.5 * (LO + HI), 0.5 * (HI - LO)
x = np.cos(np.pi * (np.arange(DEG + 1) + 0.5) / (DEG + 1))
p = C.cheb2poly(C.chebfit(x, f(mid + half * x), DEG))
u = np.linspace(-1, 1, 400001)
acc = np.full_like(u, p[-1])
for a in p[-2::-1]:
    acc = acc * u + a
print('mid %g half %g' % (mid, half))
print(', '.join(float.hex(float(a)) for a in p))
print('max relative error %.3g' % np.max(np.abs(acc / f(mid + half * u) - 1)))
