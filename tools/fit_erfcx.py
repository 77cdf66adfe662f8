"""Derivation of the polynomials of erfc_given_exp() and exp_nonpos() in aztotmd_amd/csrc/kernels.hip.h (table kCoulCoef).

erfcx(x) = exp(x^2) erfc(x) on 0 <= x <= 4: degree-15 polynomial in t = 3u - 2, u = 1/(1 + x/2) (Chebyshev interpolation converted to the
monomial basis; coefficients decay, Horner is well conditioned): max relative error 7e-14 against scipy.special.erfcx (round 4; degree 16: 8.5e-15,
one more FMA per visit and two more scalar registers in a loop that had run out of them).
exp(r) on |r| <= ln2 / 2: 1 + r (1 + r q(r)) with q of degree 8 fitted to (e^r - 1 - r) / r^2: max relative error 1.2e-15 (round 4; Taylor to r^12: 1.7e-16,
two more FMAs).                                                                                         Run: python tools/fit_erfcx.py
"""
import numpy as np
from numpy.polynomial import chebyshev as C
from scipy import special
import mpmath as mp

mp.mp.dps = 40
X, n = 4.0, 15
a, b = 1 / (1 + X / 2), 1.0
nodes = np.cos(np.pi * (np.arange(8 * n) + 0.5) / (8 * n))
un = (nodes + 1) * (b - a) / 2 + a
mono = C.cheb2poly(C.chebfit(nodes, special.erfcx(2 * (1 / un - 1)), n))
xs = np.linspace(0, X, 400001)
t = 3.0 / (1 + xs / 2) - 2.0
acc = np.zeros_like(t) + mono[-1]
for c in mono[-2::-1]:
    acc = acc * t + c
print("erfcx, degree %d: max relative error on [0, 4]: %.2e" % (n, (np.abs(acc - special.erfcx(xs)) / special.erfcx(xs)).max()))
print("    // erfcx fit, highest degree first")
print("    " + ", ".join("%.17e" % c for c in mono[::-1]) + ",")

h = float(mp.log(2) / 2)
nq = 8
nodes = np.cos(np.pi * (np.arange(6 * nq) + 0.5) / (6 * nq))
f = [float((mp.e ** (mp.mpf(x) * h) - 1 - mp.mpf(x) * h) / (mp.mpf(x) * h) ** 2) for x in nodes]
mono = C.cheb2poly(C.chebfit(nodes, f, nq))
co = [mono[k] / h ** k for k in range(nq + 1)]
worst = 0
for x in np.linspace(-h, h, 2001):
    q = mp.mpf(0)
    for c in co[::-1]:
        q = q * mp.mpf(x) + mp.mpf(c)
    p = 1 + mp.mpf(x) * (1 + mp.mpf(x) * q)
    worst = max(worst, abs(p - mp.e ** mp.mpf(x)) / mp.e ** mp.mpf(x))
print("exp, q of degree %d: max relative error on |r| <= ln2/2: %.2e" % (nq, float(worst)))
print("    // q(r), highest degree first")
print("    " + ", ".join("%.17e" % c for c in co[::-1]) + ",")
