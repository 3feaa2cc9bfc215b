#!/usr/bin/env python3
"""Generates the coefficients of the `kWide` tier of kmc_device_math.hip.h (se3_coefficients<kWide>): the three SE(3) exponential
coefficients as polynomials in u = theta^2 on the whole range a frame between two poses can reach (|phi| <= pi from Log,
|s| <= 1, so theta <= pi; fitted on theta <= 3.25):

    A(u) = sin(t)/t        B(u) = (1 - cos t)/t^2        C(u) = (t - sin t)/t^3,      t = sqrt(u)

All three are entire functions of u with factorially decaying Taylor coefficients, so interpolation at Chebyshev nodes of
[0, 3.25^2] converges very fast: degree 6 / 5 / 5 is below f32 rounding.  No square root, no division, no 1 - cos
cancellation -- 16 fma per point instead of ocml sincosf + sqrt + two divisions.

    python tools/gen_wide_coeffs.py      -> the C initialisers, the fit error and the error of an f32 Horner evaluation
"""
import math

import numpy as np
from numpy.polynomial import chebyshev as Ch, polynomial as P

THETA_MAX = 3.25
U = THETA_MAX ** 2
DEGREES = {"A": 6, "B": 5, "C": 5}
FIRST_FACTORIAL = {"A": 1, "B": 2, "C": 3}  # f(u) = sum_k (-1)^k u^k / (2k + k0)!


def truth(u, k0):
    u = np.asarray(u, dtype=np.longdouble)
    acc = np.zeros_like(u)
    for k in range(40, -1, -1):
        acc = acc + ((-1) ** k) * u ** k / np.longdouble(math.factorial(2 * k + k0))
    return acc


def fit(name):
    deg, k0 = DEGREES[name], FIRST_FACTORIAL[name]
    n = deg + 1
    x = np.cos(np.pi * (np.arange(n) + 0.5) / n)
    c = Ch.chebfit(x, truth((x + 1) * U / 2, k0).astype(np.float64), deg)
    px = Ch.cheb2poly(c)
    lin = np.array([-1.0, 2.0 / U])  # x = 2u/U - 1
    pu = np.zeros(1)
    for coef in px[::-1]:
        pu = P.polyadd(P.polymul(pu, lin), [coef])
    return pu


def horner_f32(coeffs, u):
    c32 = coeffs.astype(np.float32)
    u32 = u.astype(np.float32)
    acc = np.full_like(u32, c32[-1])
    for coef in c32[-2::-1]:  # one rounding per step, like v_fma_f32
        acc = (acc.astype(np.float64) * u32.astype(np.float64) + np.float64(coef)).astype(np.float32)
    return acc


def main():
    us = np.linspace(0.0, U, 400001)
    print("// tools/gen_wide_coeffs.py: theta <= %.2f, u = theta^2 <= %.4f" % (THETA_MAX, U))
    for name in "ABC":
        pu = fit(name)
        t = truth(us, FIRST_FACTORIAL[name]).astype(np.float64)
        e_fit = np.abs(P.polyval(us, pu) - t).max()
        e_f32 = np.abs(horner_f32(pu, us).astype(np.float64) - t).max()
        print("// %s: degree %d, fit error %.2e, f32 Horner error %.2e (absolute; %s(0) = %g)" % (name, DEGREES[name], e_fit, e_f32, name, t[0]))
        print("static constexpr float kWide%s[%d] = {%s};" % (name, len(pu), ", ".join("%sf" % repr(float(np.float32(a))) for a in pu)))


if __name__ == "__main__":
    main()
