#!/usr/bin/env python3
"""Generates the minimax-like coefficients used by kmc_device_math.hip.h for
    atan(q) / (2*pi)  ~=  q * P(q^2),  q in [0, 1]
(Chebyshev-node interpolation of g(t) = atan(sqrt t)/(2 pi sqrt t) on t in [0,1], converted to the power
basis).  Run: python tools/gen_atan_coeffs.py  -> prints the C initialiser and the measured max error.

    python tools/gen_atan_coeffs.py --f64
prints kRedoTable[0..11] of kmc_device_math.hip.h: atan(r) / r as a degree-11 polynomial in r^2 on r^2 <= tan^2(pi/8) for
the f64 atan2 of the near-origin guard's redo (interpolation at Chebyshev nodes in long double, exact rational conversion to
the power basis), and checks the complete atan2_f64_lean algorithm -- emulated in numpy f64 -- on 2 M points against long double."""
import sys

import numpy as np
from numpy.polynomial import chebyshev as Ch, polynomial as P

DEG = 7  # 8 coefficients


def g(t):
    t = np.asarray(t, dtype=np.float64)
    s = np.sqrt(t)
    return np.where(t > 0, np.arctan(s) / (2 * np.pi * np.where(s > 0, s, 1)), 1 / (2 * np.pi))


def main():
    k = np.arange(DEG + 1)
    x = np.cos((2 * k + 1) * np.pi / (2 * (DEG + 1)))
    c = Ch.chebfit(x, g((x + 1) / 2), DEG)
    pc = Ch.cheb2poly(c)
    poly = np.zeros(1)
    for i, a in enumerate(pc):
        poly = P.polyadd(poly, a * P.polypow([-1, 2], i))
    q = np.linspace(0, 1, 400001)
    err = np.abs(q * P.polyval(q * q, poly) - np.arctan(q) / (2 * np.pi)).max()
    print("// max |error| = %.3e turns (f64 evaluation), degree %d in q^2" % (err, DEG))
    print("static constexpr float kAtanTurns[%d] = {" % (DEG + 1))
    for a in poly:
        print("    %sf,  // %s" % (repr(float(np.float32(a))), float(a).hex()))
    print("};")


def main_f64():
    from fractions import Fraction as Fr

    ld = np.longdouble
    deg = 11
    tan_pi8 = np.tan(np.pi / 8)
    zmax = ld(tan_pi8) ** 2 * ld(1.000001)

    def f(z):
        z = np.asarray(z, dtype=ld)
        r = np.sqrt(z)
        return np.where(z > 1e-12, np.arctan(r) / np.where(r == 0, 1, r), 1 - z / 3 + z * z / 5)

    k = np.arange(deg + 1)
    nodes = np.cos(np.pi * (k + 0.5) / (deg + 1)).astype(ld)
    t = [Fr(float(v)) for v in ((nodes + 1) / 2)]              # nodes in [0, 1], exact rationals of their f64 values
    y = f(np.array([float(v) for v in t], dtype=ld) * zmax)
    yf = [Fr(float(v)) + Fr(float(v - ld(float(v)))) for v in y]  # long double = sum of two doubles, exactly
    n = deg + 1
    coef = list(yf)                                              # Newton divided differences, exact
    for j in range(1, n):
        for i in range(n - 1, j - 1, -1):
            coef[i] = (coef[i] - coef[i - 1]) / (t[i] - t[i - j])
    mono, basis = [Fr(0)] * n, [Fr(1)]
    for i in range(n):
        for d, b in enumerate(basis):
            mono[d] += coef[i] * b
        nb = [Fr(0)] * (len(basis) + 1)
        for d, b in enumerate(basis):
            nb[d + 1] += b
            nb[d] -= b * t[i]
        basis = nb
    zm = Fr(float(zmax)) + Fr(float(zmax - ld(float(zmax))))
    c = [float(mono[d] / zm ** d) for d in range(n)]
    print("// atan(r) / r in r^2, highest degree first")
    print(", ".join("%.17e" % v for v in c[::-1]))
    rng = np.random.default_rng(1)
    N = 2_000_000
    x = rng.normal(size=N) * rng.choice([1e-3, 1, 50], N)
    yv = rng.normal(size=N) * rng.choice([1e-3, 1, 50], N)
    ax, ay = np.abs(x), np.abs(yv)
    mx, mn = np.maximum(ax, ay), np.minimum(ax, ay)
    big = mn > tan_pi8 * mx
    num, den = np.where(big, mn - mx, mn), np.where(big, mn + mx, mx)
    r = np.where(den == 0, 0.0, num / np.where(den == 0, 1, den))
    z = r * r
    p = np.full_like(z, c[-1])
    for v in c[-2::-1]:
        p = p * z + v
    a = r * p
    a = np.where(big, np.pi / 4 + a, a)
    a = np.where(ay > ax, np.pi / 2 - a, a)
    a = np.where(np.signbit(x), np.pi - a, a)
    got = np.copysign(a, yv)
    ref = np.arctan2(yv.astype(ld), x.astype(ld))
    print("// max |error| of the complete atan2 on %d points: %.3e rad" % (N, float(np.max(np.abs(got.astype(ld) - ref)))))


if __name__ == "__main__":
    main_f64() if "--f64" in sys.argv else main()
