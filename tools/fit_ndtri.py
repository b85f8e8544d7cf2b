#!/usr/bin/env python3
"""Generates the coefficients of ndtri_w (optionslab_amd/csrc/olmc_kernels.h): Chebyshev interpolants of
f(w) = sqrt(2) erfinv(x)/x, x = sqrt(1 - e^-w)  (Phi^-1(p) = x f(w), x = 2p - 1, w = -ln 4p(1-p); the sqrt(2) sits in the
coefficients since round 5: one multiplication fewer per point), in 50-digit arithmetic (mpmath), converted to powers of (w - centre) and of
(sqrt(w) - centre), with the double-precision Horner error of every candidate degree.  CPU only: python tools/fit_ndtri.py [out.json]"""
import mpmath as mp, numpy as np, json, sys
mp.mp.dps = 50
def f_of_w(w):
    w = mp.mpf(w)
    if w == 0: return mp.sqrt(2) * mp.sqrt(mp.pi)/2
    x = mp.sqrt(1 - mp.e**(-w))
    return mp.sqrt(2) * mp.erfinv(x)/x
def cheb_fit(g, a, b, deg):
    # Chebyshev interpolation coefficients in mp, then monomial in t=(v-c) (unscaled centre c=(a+b)/2)
    n = deg + 1
    c, h = (mp.mpf(a)+b)/2, (mp.mpf(b)-a)/2
    nodes = [mp.cos(mp.pi*(k+mp.mpf(1)/2)/n) for k in range(n)]
    vals = [g(c + h*t) for t in nodes]
    coef = []
    for j in range(n):
        s = sum(vals[k]*mp.cos(mp.pi*j*(k+mp.mpf(1)/2)/n) for k in range(n))
        coef.append((2 if j else 1)*s/n)
    # chebyshev -> monomial in t
    T = [[mp.mpf(1)], [mp.mpf(0), mp.mpf(1)]]
    for j in range(2, n):
        nxt = [mp.mpf(0)] + [2*x for x in T[j-1]]
        for i, x in enumerate(T[j-2]): nxt[i] -= x
        T.append(nxt)
    mono = [mp.mpf(0)]*n
    for j in range(n):
        for i, x in enumerate(T[j]): mono[i] += coef[j]*x
    # t = (v - c)/h  -> powers of (v - c)
    mono = [m / h**i for i, m in enumerate(mono)]
    return float(c), [float(m) for m in mono]
def horner(coefs, t):
    r = np.zeros_like(t) + coefs[-1]
    for c in coefs[-2::-1]: r = r*t + c
    return r
res = {}
for name, g, a, b, degs in (("A", f_of_w, 0.0, 6.25, (18, 20, 22, 24)),
                            ("B", lambda s: f_of_w(mp.mpf(s)**2), 2.5, 4.70, (16, 18, 20, 22))):
    for deg in degs:
        c, mono = cheb_fit(g, a, b, deg)
        grid = np.linspace(a, b, 4001)
        ref = np.array([float(g(v)) for v in grid])
        got = horner(mono, grid - c)
        err = np.max(np.abs(got/ref - 1))
        print(name, deg, "max rel err", err, flush=True)
        res[f"{name}{deg}"] = dict(c=c, coefs=mono, err=float(err))
json.dump(res, open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/ndtri_fits.json", "w"))


# ---- the atanh-series polynomial of neg_log_unit: ln m = 2 s + s^3 Q(s^2), s = (m - 1)/(m + 1), m in [sqrt(1/2), sqrt(2)]
def log_q(u):
    u = mp.mpf(u)
    if u == 0:
        return mp.mpf(2) / 3
    s = mp.sqrt(u)
    return (2 * mp.atanh(s) - 2 * s) / s**3


smax2 = ((mp.sqrt(2) - 1) / (mp.sqrt(2) + 1))**2 * mp.mpf("1.02")
for deg in (5, 6, 7):
    c, mono = cheb_fit(log_q, 0, smax2, deg)
    # powers of (u - c) -> powers of u
    poly = [mp.mpf(0)] * (deg + 1)
    for i, m in enumerate(mono):
        for k in range(i + 1):
            poly[k] += mp.mpf(m) * mp.binomial(i, k) * (-mp.mpf(c))**(i - k)
    print("log Q degree", deg, [float(x) for x in poly])
