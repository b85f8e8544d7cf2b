#!/usr/bin/env python3
"""Interleaved A/B of whole libolmc builds: one subprocess per (library, round), each timing the
European path kernel with HIP events (olmc_kernel_time).  Usage (GPU box):
    python tools/ab_libs.py libA.so libB.so ... [--n 1000000] [--m 252] [--rounds 7] [--case european|greeks8|greeks14|greeks8_lean|greeks14_lean|asian|asian_fast|asian_fast_anti|asian_anti|asian_geo|barrier|heston|merton|kou|autocall[_anti]|cliquet[_anti]|american|{barrier,lookback}_greeks{8,14}[_anti]|asian_greeks8|asian_greeks14[_rho]|qmc|qmc_cv|qmc_greeks8|qmc_greeks14]"""
import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import json, sys
sys.path.insert(0, %r)
from optionslab_amd import _hip
N, M = int(sys.argv[1]), int(sys.argv[2])
_hip.lib(); _hip.profile_enable(True)
P = (100.0, 100.0, 1.0, 0.05, 0.2, 0.0)
CASES = {
    "european": lambda s: _hip.european(*P, True, N, M, s, True),
    "greeks8": lambda s: _hip.european_greeks_fd(*P, True, N, M, s, False)[1][0],
    "greeks14": lambda s: _hip.european_greeks_fd(*P, True, N, M, s, True)[1][0],
    # the call MonteCarloPricer.greeks() makes: no per-evaluation statistics asked for (prices-only kernel from round 3 on)
    "greeks8_lean": lambda s: type("R", (), dict(zip(("price", "sum"), (lambda v: (v[0], v[1]))(_hip.european_greeks_fd(*P, True, N, M, s, False, want_evals=False)[0]))))(),
    "greeks14_lean": lambda s: type("R", (), dict(zip(("price", "sum"), (lambda v: (v[0], v[8]))(_hip.european_greeks_fd(*P, True, N, M, s, True, want_evals=False)[0]))))(),
    "asian": lambda s: _hip.asian(*P, True, False, N, M, s, False),
    "asian_fast": lambda s: _hip.asian(*P, True, False, N, M, s, False, fast=True),
    "asian_fast_anti": lambda s: _hip.asian(*P, True, False, N, M, s, True, fast=True),
    "asian_anti": lambda s: _hip.asian(*P, True, False, N, M, s, True),
    "asian_geo": lambda s: _hip.asian(*P, True, True, N, M, s, False),
    "barrier": lambda s: _hip.barrier(*P, True, 120.0, 0, N, M, s, False),
    "merton": lambda s: _hip.jump_diffusion(*P, True, False, 0.5, -0.1, 0.2, 0.0, N, M, s),
    "kou": lambda s: _hip.jump_diffusion(*P, True, True, 1.0, 0.4, 10.0, 5.0, N, M, s),
    "autocall": lambda s: _hip.autocallable(100.0, 1.0, 0.05, 0.2, 0.0, 1.0, 0.8, 0.08, 0.6, 21, N, M, s),
    "autocall_anti": lambda s: _hip.autocallable(100.0, 1.0, 0.05, 0.2, 0.0, 1.0, 0.8, 0.08, 0.6, 21, N, M, s, True),
    "cliquet": lambda s: _hip.cliquet(100.0, 1.0, 0.05, 0.2, 0.0, 0.05, -0.05, 0.3, 0.0, 12, N, M, s),
    "cliquet_anti": lambda s: _hip.cliquet(100.0, 1.0, 0.05, 0.2, 0.0, 0.05, -0.05, 0.3, 0.0, 12, N, M, s, True),
    "heston": lambda s: _hip.heston(100.0, 100.0, 1.0, 0.05, 0.0, True, 2.0, 0.04, 0.3, -0.7, 0.04, N, M, s, False),
}
# round 4: the per-date launches of the American option, and the Sobol kernels (a table of `M` dimensions built once, seed 42)
CASES["asian_greeks8"] = lambda s: _hip.asian_greeks_fd(*P, True, N, M, s, False, False)[1][0]
CASES["asian_greeks14"] = lambda s: _hip.asian_greeks_fd(*P, True, N, M, s, False, True)[1][0]
CASES["asian_greeks14_rho"] = lambda s: _hip.asian_greeks_fd(*P, True, N, M, s, False, True)[1][6]       # the r + h evaluation
CASES["american"] = lambda s: _hip.american_lsm(*P, False, N, M, 3, s)
# round 5: the fused barrier / lookback Greeks (payoff 0 = up-and-out barrier at 120, 4 = floating lookback), 8 / 14 contracts, +- antithetic
for _name, _payoff, _level in (("barrier", 0, 120.0), ("lookback", 4, 0.0)):
    for _k, _second in ((8, False), (14, True)):
        for _anti in (False, True):
            CASES[f"{_name}_greeks{_k}" + ("_anti" if _anti else "")] = (lambda s, p=_payoff, l=_level, a=_anti, o=_second: _hip.extrema_greeks_fd(*P, True, p, l, N, M, s, a, o)[1][0])
if sys.argv[3].startswith("qmc"):
    import numpy as np
    from optionslab_amd.monte_carlo import sobol_tables
    SV, SH = sobol_tables(M, 42)
    CASES["qmc"] = lambda s: _hip.european_qmc(*P, True, N, SV, SH)
    CASES["qmc_cv"] = lambda s: (lambda m: type("R", (), dict(price=m.value, sum=m.sum_d))())(_hip.european_qmc_cv(*P, True, N, SV, SH))
    CASES["qmc_greeks8"] = lambda s: _hip.european_qmc_greeks_fd(*P, True, N, SV, SH, False)[1][0]
    CASES["qmc_greeks14"] = lambda s: _hip.european_qmc_greeks_fd(*P, True, N, SV, SH, True)[1][0]
run = CASES[sys.argv[3]]
import os, time
if os.environ.get("OLMC_AB_TUNE"):                      # "knob=value,knob=value" applied before anything runs
    for kv in os.environ["OLMC_AB_TUNE"].split(","):
        k, v = kv.split("=")
        _hip.tune(int(k), int(v))
_hip.profile_enable(False)
for i in range(400): run(1 + i)                          # clocks up (an idle device needs tens of ms of load)
t0 = time.perf_counter()
for i in range(100): st = run(42 + i)
wall = (time.perf_counter() - t0) / 100
_hip.profile_enable(True)
_hip.profile_reset()
for i in range(60): st = run(42 + i)
n, ms = _hip.kernel_time()
print(json.dumps({"us": (ms / n * 1e3) if n else wall * 1e6, "wall_us": wall * 1e6, "price": st.price, "sum": st.sum}))
""" % ROOT

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--m", type=int, default=252)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--case", default="european")
a = ap.parse_args()
# a library may carry a tuning suffix: "libolmc.so@7=-1" runs it with olmc_tune(7, -1) (e.g. split workgroups off)
res = {l: [] for l in a.libs}
wall = {l: [] for l in a.libs}
price = {}
for r in range(a.rounds):
    for l in a.libs:
        path, _, tune = l.partition("@")
        env = dict(os.environ, OLMC_LIBRARY=os.path.abspath(path))
        if tune:
            env["OLMC_AB_TUNE"] = tune
        out = subprocess.run([sys.executable, "-c", CHILD, str(a.n), str(a.m), a.case], env=env, capture_output=True, text=True)
        if out.returncode != 0:
            raise SystemExit(f"{l}: child failed (rc {out.returncode}): {out.stderr[-600:]}")
        d = json.loads(out.stdout.strip().splitlines()[-1])
        res[l].append(d["us"])
        wall[l].append(d["wall_us"])
        price[l] = (d["price"], d["sum"])
for l in a.libs:
    v, w = res[l], wall[l]
    print(f"{os.path.basename(l):34s} kernel median {statistics.median(v):8.2f} us  min {min(v):8.2f}  max {max(v):8.2f} | blocking call median "
          f"{statistics.median(w):8.2f} us | price {price[l][0]:.12f} sum {price[l][1]!r}", flush=True)
