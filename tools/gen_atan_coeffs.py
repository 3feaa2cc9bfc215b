#!/usr/bin/env python3
"""Generates the minimax-like coefficients used by kmc_device_math.hip.h for
    atan(q) / (2*pi)  ~=  q * P(q^2),  q in [0, 1]
(Chebyshev-node interpolation of g(t) = atan(sqrt t)/(2 pi sqrt t) on t in [0,1], converted to the power
basis).  Run: python tools/gen_atan_coeffs.py  -> prints the C initialiser and the measured max error."""
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


if __name__ == "__main__":
    main()
