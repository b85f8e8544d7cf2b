#!/usr/bin/env python3
"""QMC (scrambled Sobol) pricing time by size.  Usage (GPU box): python tools/qmc_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import optionslab_amd as ol  # noqa: E402

for n, m in ((2**14, 16), (2**17, 64), (2**20, 64), (2**20, 252), (2**22, 252)):
    p = ol.MonteCarloPricer(n, m, 42, ol.MCMethod.QMC)
    for _ in range(2):
        p.price(100.0, 100.0, 1.0, 0.05, 0.2, "call")
    t0 = time.perf_counter()
    for _ in range(3):
        res = p.price(100.0, 100.0, 1.0, 0.05, 0.2, "call", return_error=True)
    dt = (time.perf_counter() - t0) / 3
    print(f"{n:>8} x {m:<4} {1e3 * dt:9.3f} ms  {n * m / dt:.3e} point-dims/s  price {res.price:.6f}", flush=True)
