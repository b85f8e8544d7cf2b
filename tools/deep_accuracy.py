#!/usr/bin/env python3
"""One-off bias hunt: prices at 2^32 paths (se ~ 1e-4) against Black-Scholes, several step counts / contracts.
Any systematic error of the fp32 Box-Muller (hardware log2/sin/cos), the 32-bit uniform tail cap or the
reductions would show as |z| >> 3.  Usage (GPU box): python tools/deep_accuracy.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import optionslab_amd as ol  # noqa: E402
from optionslab_amd import _hip  # noqa: E402

if "--round4" in sys.argv:
    # Round 4: (i) the Sobol price against Black-Scholes as the point count grows -- through all three launch shapes (split workgroups up
    # to 2^18 points, one point per thread, eight points per thread from 2^20): a wrong point, dimension or association would show as a
    # plateau; (ii) the fp64-normals kernel of the instrumented build against the product kernel and Black-Scholes at 2^30 paths.
    from optionslab_amd.monte_carlo import sobol_tables
    from tools.probe import binding as probe
    S, K, T, r, v = 100.0, 100.0, 1.0, 0.05, 0.2
    bs = ol.black_scholes(S, K, T, r, v, "call")
    for dims in (16, 64, 252):
        sv, sh = sobol_tables(dims, 42)
        for log_n in (10, 12, 14, 16, 18, 19, 20, 22, 24):
            st = _hip.european_qmc(S, K, T, r, v, 0.0, True, 1 << log_n, sv, sh)
            print(json.dumps(dict(kind="scrambled Sobol vs Black-Scholes", dims=dims, log2_points=log_n, price=st.price, bs=bs, abs_err=st.price - bs,
                                  shape="split" if (log_n <= 18 and dims >= 16) else ("one point" if log_n < 20 else "eight points"))), flush=True)
    import math
    for M, batches in ((16, 16), (252, 4)):              # batches of 2^26 paths (the probe kernel's launch limit), distinct seeds, sums combined
        tot = {"f64": [0.0, 0.0, 0], "f32": [0.0, 0.0, 0]}
        for k in range(batches):
            for name, st in (("f64", probe.european_f64_normals(S, K, T, r, v, 0.0, True, 1 << 26, M, 100 + k)),
                             ("f32", _hip.european(S, K, T, r, v, 0.0, True, 1 << 26, M, 100 + k, True))):
                tot[name][0] += st.sum; tot[name][1] += st.sumsq; tot[name][2] += st.n
        out = {}
        for name, (sx, sxx, n) in tot.items():
            mean = sx / n
            out[name] = (math.exp(-r * T) * mean, math.exp(-r * T) * math.sqrt(max(sxx / n - mean * mean, 0.0) / n))
        print(json.dumps(dict(kind="fp64 normals (instrumented build) vs fp32 normals (product) vs Black-Scholes", n_paths=batches << 26, n_steps=M, bs=bs,
                              f64_price=out["f64"][0], f64_z=(out["f64"][0] - bs) / out["f64"][1], f32_price=out["f32"][0], f32_z=(out["f32"][0] - bs) / out["f32"][1],
                              std_error=out["f64"][1], diff_over_se=(out["f64"][0] - out["f32"][0]) / (2 ** 0.5 * out["f64"][1]),
                              note="std_error is the naive formula over 2n antithetic samples (overstates the error): z is conservative")), flush=True)
    raise SystemExit(0)

N = 1 << 32
rows = []
for (S, K, T, r, v, q, call, M, seed) in [(100.0, 100.0, 1.0, 0.05, 0.2, 0.0, True, 16, 1), (100.0, 100.0, 1.0, 0.05, 0.2, 0.0, False, 16, 2),
                                           (100.0, 130.0, 1.0, 0.05, 0.2, 0.0, True, 4, 3), (100.0, 70.0, 2.0, 0.03, 0.4, 0.02, False, 8, 4),
                                           (100.0, 100.0, 1.0, 0.05, 0.2, 0.0, True, 1, 5), (100.0, 160.0, 1.0, 0.05, 0.2, 0.0, True, 252, 6)]:
    n = N if M <= 16 else N // 16
    st = _hip.european(S, K, T, r, v, q, call, n, M, seed, True)
    bs = ol.black_scholes(S, K, T, r, v, "call" if call else "put", q)
    # the reported std_error is the reference's naive formula (overstates the antithetic error): z is conservative
    rows.append(dict(S=S, K=K, T=T, r=r, sigma=v, q=q, call=call, n_paths=n, n_steps=M, price=st.price, bs=bs,
                     std_error=st.std_error, z=(st.price - bs) / st.std_error, rel_err=(st.price - bs) / bs))
    print(json.dumps(rows[-1]), flush=True)

# Path-dependent kernels against closed forms at 2^26 paths (se ~ 1e-3).  The geometric average over the dates
# t_i = i T / M, i = 1..M is lognormal exactly: ln G ~ N(ln S + (r - q - s^2/2) T (M+1)/(2M), s^2 T (M+1)(2M+1)/(6 M^2)),
# so the DISCRETE closed form is the target (the reference's price_geometric_closed_form, exotic_options.py:133-160, is
# the continuous limit and sits 5e-3 lower at M = 1024).  Heston with sigma_v -> 0 and Merton against Black-Scholes / series.
import math


def discrete_geometric_call(S, K, T, r, v, q, M):
    mu = (r - q - 0.5 * v * v) * T * (M + 1) / (2 * M)
    s2 = v * v * T * (M + 1) * (2 * M + 1) / (6 * M * M)
    s, m = math.sqrt(s2), math.log(S) + mu
    cdf = lambda x: 0.5 * math.erfc(-x / math.sqrt(2.0))
    d1 = (m - math.log(K) + s2) / s
    return math.exp(-r * T) * (math.exp(m + 0.5 * s2) * cdf(d1) - K * cdf(d1 - s))


n = 1 << 26
a = ol.AsianOption(100.0, 100.0, 1.0, 0.05, 0.2, seed=11)
for M in (1024, 252):
    p, se = a.price(n, M, "geometric", "call", return_error=True)
    cf = discrete_geometric_call(100.0, 100.0, 1.0, 0.05, 0.2, 0.0, M)
    rows.append(dict(kind=f"geometric asian 2^26 x {M} vs the discrete-monitoring closed form", price=float(p), closed_form=cf, std_error=se,
                     z=(float(p) - cf) / se, continuous_closed_form=float(a.price_geometric_closed_form("call"))))
    print(json.dumps(rows[-1]), flush=True)
import warnings
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    h = ol.HestonPricer(kappa=2.0, theta=0.04, sigma_v=1e-7, rho=0.0, v0=0.04)
p, se = h.price_monte_carlo(100.0, 100.0, 1.0, 0.05, 0.0, "call", n, 64, 12, return_error=True)
bs = ol.black_scholes(100.0, 100.0, 1.0, 0.05, 0.2, "call")
rows.append(dict(kind="heston sigma_v->0 2^26 x 64 vs Black-Scholes", price=float(p), bs=float(bs), std_error=se, z=(float(p) - float(bs)) / se))
print(json.dumps(rows[-1]), flush=True)
m = ol.MertonJumpDiffusion(0.5, -0.1, 0.2)
p, se = m.price_monte_carlo(100.0, 100.0, 1.0, 0.05, 0.2, "call", 0.0, n, 64, 13, return_error=True)
series = m.price(100.0, 100.0, 1.0, 0.05, 0.2, "call")
rows.append(dict(kind="merton lambda=0.5 2^26 x 64 vs series", price=float(p), series=float(series), std_error=se, z=(float(p) - float(series)) / se))
print(json.dumps(rows[-1]), flush=True)
# Round 2: the reference-precision arithmetic Asian (fp64 cumulative log-return + fp64 exp2 per date) against the opt-in
# fp32-exponent kernel on the SAME normals at 2^24 paths x 1024 dates: their difference is the fp32 exponent's whole
# effect on a price (stated bound 2e-6 relative); se ~ 2e-3 here, so the statistical error is 1000x larger than it.
n = 1 << 24
p64 = _hip.asian(100.0, 100.0, 1.0, 0.05, 0.2, 0.0, True, False, n, 1024, 21, False)
p32 = _hip.asian(100.0, 100.0, 1.0, 0.05, 0.2, 0.0, True, False, n, 1024, 21, False, fast=True)
rows.append(dict(kind="arithmetic asian 2^24 x 1024: fp64-exponent kernel vs fp32-exponent kernel, same normals", price_fp64=p64.price, price_fp32=p32.price,
                 rel_diff=(p32.price - p64.price) / p64.price, std_error=p64.std_error, se_rel_diff=(p32.std_error - p64.std_error) / p64.std_error))
print(json.dumps(rows[-1]), flush=True)
# Round 3: (a) sixteen times deeper on the headline contract: 2^36 paths x 16 steps (se ~ 3e-5, 3e-6 of the price) -- the antithetic
# naive se overstates the error, so |z| stays conservative; (b) the fused finite-difference Greeks (sum-only kernel, folded
# exchanges, workgroup-wide row sums) at 2^30 paths x 16 steps against the analytic Black-Scholes Greeks: what is left is the
# finite-difference truncation of unified_greeks.py's steps (h_S = 1, h_sigma = 0.01, h_r = 1e-4, h_T = 1/365), listed beside it from
# the closed form itself; (c) the arithmetic Asian with the table-driven exp2 against the degree-11 form is a pointwise matter
# (tests: <= 2 ulp) -- here its price at 2^26 x 252 next to the geometric one as a sanity bracket (arithmetic >= geometric).
st = _hip.european(100.0, 100.0, 1.0, 0.05, 0.2, 0.0, True, 1 << 36, 16, 77, True)
bs = ol.black_scholes(100.0, 100.0, 1.0, 0.05, 0.2, "call", 0.0)
rows.append(dict(kind="european ATM call 2^36 paths x 16 steps", n_paths=1 << 36, price=st.price, bs=float(bs), std_error=st.std_error,
                 z=(st.price - bs) / st.std_error, rel_err=(st.price - bs) / bs))
print(json.dumps(rows[-1]), flush=True)
from oracle import numpy_reference as orc  # noqa: E402  (closed-form Greeks only; nothing is priced on the CPU)
g = ol.MonteCarloPricer(1 << 30, 16, 5).greeks(100.0, 100.0, 1.0, 0.05, 0.2, "call", include_second_order=True)
exact = orc.bs_greeks(100.0, 100.0, 1.0, 0.05, 0.2, "call")
fd_of_bs = orc.fd_greeks(lambda S, K, T, r, v, typ, q=0.0: float(ol.black_scholes(S, K, T, r, v, typ, q)), 100.0, 100.0, 1.0, 0.05, 0.2, "call", 0.0,
                         include_second_order=True)
rows.append(dict(kind="fused FD Greeks 2^30 paths x 16 steps vs analytic / vs the same finite differences of the closed form",
                 greeks={k: g[k] for k in g}, analytic={k: exact[k] for k in exact if k in g},
                 fd_of_closed_form={k: fd_of_bs[k] for k in g}, abs_diff_vs_fd_of_closed_form={k: g[k] - fd_of_bs[k] for k in g}))
print(json.dumps(rows[-1]), flush=True)
a = ol.AsianOption(100.0, 100.0, 1.0, 0.05, 0.2, seed=31)
pa, sea = a.price(1 << 26, 252, "arithmetic", "call", return_error=True)
pg, seg = a.price(1 << 26, 252, "geometric", "call", return_error=True)
rows.append(dict(kind="arithmetic (table exp2) vs geometric Asian 2^26 x 252, same seed", arithmetic=float(pa), geometric=float(pg), se=sea,
                 geometric_closed_form=discrete_geometric_call(100.0, 100.0, 1.0, 0.05, 0.2, 0.0, 252)))
print(json.dumps(rows[-1]), flush=True)
