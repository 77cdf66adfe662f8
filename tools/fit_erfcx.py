"""Derivation of the polynomial used by erfc_given_exp() in aztotmd_amd/csrc/pair_tile.hip.h.

erfcx(x) = exp(x^2) erfc(x) on 0 <= x <= 4 is fitted by a degree-16 polynomial in t = 3u - 2, u = 1/(1 + x/2)
(Chebyshev interpolation, converted to the monomial basis; coefficients decay, Horner is well conditioned).
Maximum relative error against scipy.special.erfcx printed below (8e-15).  Run: python tools/fit_erfcx.py
"""
import numpy as np
from numpy.polynomial import chebyshev as C
from scipy import special

X, n = 4.0, 16
a, b = 1 / (1 + X / 2), 1.0
nodes = np.cos(np.pi * (np.arange(8 * n) + 0.5) / (8 * n))
un = (nodes + 1) * (b - a) / 2 + a
mono = C.cheb2poly(C.chebfit(nodes, special.erfcx(2 * (1 / un - 1)), n))
xs = np.linspace(0, X, 400001)
t = 3.0 / (1 + xs / 2) - 2.0
acc = np.zeros_like(t) + mono[-1]
for c in mono[-2::-1]:
    acc = acc * t + c
print("max relative error on [0, 4]: %.2e" % (np.abs(acc - special.erfcx(xs)) / special.erfcx(xs)).max())
for c in mono:
    print("%.17e," % c)
