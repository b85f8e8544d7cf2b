#!/usr/bin/env python3
"""Wall of MCMethod.QMC pricings (GPU box): repeated price() on one table, and the literal FD Greeks (8 pricings, one table)."""
import json, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import optionslab_amd as ol
ATM = (100.0, 100.0, 1.0, 0.05, 0.2)
for n, m in ((1 << 14, 16), (1 << 17, 64), (1 << 17, 252), (1 << 20, 252)):
    p = ol.MonteCarloPricer(n, m, 42, ol.MCMethod.QMC)
    for _ in range(20): p.price(*ATM, "call")
    ts = []
    for _ in range(60):
        t0 = time.perf_counter(); p.price(*ATM, "call", return_error=True); ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); g = ol.compute_greeks_unified(p, *ATM, "call", include_second_order=False); tg = time.perf_counter() - t0
    print(json.dumps(dict(points=n, dims=m, price_us=statistics.median(ts) * 1e6, greeks8_us=tg * 1e6, delta=g["delta"])), flush=True)
