#!/usr/bin/env python3
"""Coefficients of the table-driven exp2_f64 (olmc_kernels.h, exp2_f64_tab): 2^(j/64 + r/64) = T[j mod 64] * 2^(j div 64) * (1 + r q(r)),
|r| <= 1/2, T[k] = 2^(k/64) rounded to fp64, q of degree 4 fitted at Chebyshev nodes in 50-digit arithmetic.  Prints the C arrays.
    python tools/fit_exp2_table.py"""
import mpmath as mp

mp.mp.dps = 50
N_TAB, DEG = 256, 3                # q has DEG + 1 coefficients: 2^(r/256) - 1 = r (c1 + c2 r + c3 r^2 + c4 r^3)


def target(r):                     # (2^(r/N_TAB) - 1) / r, analytic at 0
    r = mp.mpf(r)
    if abs(r) < mp.mpf(10) ** -20:
        return mp.log(2) / N_TAB
    return mp.expm1(r * mp.log(2) / N_TAB) / r


nodes = [mp.mpf(1) / 2 * mp.cos(mp.pi * (2 * i + 1) / (2 * (DEG + 1))) for i in range(DEG + 1)]
A = mp.matrix(DEG + 1, DEG + 1)
b = mp.matrix(DEG + 1, 1)
for i, x in enumerate(nodes):
    for j in range(DEG + 1):
        A[i, j] = x ** j
    b[i] = target(x)
c = mp.lu_solve(A, b)
worst = 0
for i in range(-2000, 2001):
    r = mp.mpf(i) / 4000
    approx = 1 + r * sum(c[j] * r ** j for j in range(DEG + 1))
    worst = max(worst, abs(approx / mp.power(2, r / N_TAB) - 1))
print(f"// interpolation error of 1 + r q(r) against 2^(r/N_TAB): {mp.nstr(worst, 3)} relative ({mp.nstr(worst / mp.mpf(2) ** -53, 3)} ulp)")
print("constexpr double kExp2Q[%d] = {%s};" % (DEG + 1, ", ".join(repr(float(c[j])) for j in range(DEG + 1))))
print("__constant__ double kExp2Tab[%d] = {" % N_TAB)
for k in range(0, N_TAB, 4):
    print("    " + ", ".join(repr(float(mp.power(2, mp.mpf(kk) / N_TAB))) for kk in range(k, k + 4)) + ",")
print("};")
