#!/usr/bin/env python3
"""Wall of MCMethod.QMC calls (GPU box): repeated price() on one table, the FD Greeks as ONE launch (fused: olmc_european_qmc_greeks_fd)
and as the literal 8 / 14 pricings (one table), medians."""
import json, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import optionslab_amd as ol
ATM = (100.0, 100.0, 1.0, 0.05, 0.2)


def med(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return statistics.median(ts) * 1e6


for n, m in ((1 << 14, 16), (1 << 17, 64), (1 << 17, 252), (1 << 20, 252)):
    p = ol.MonteCarloPricer(n, m, 42, ol.MCMethod.QMC)
    for _ in range(20): p.price(*ATM, "call")
    row = dict(points=n, dims=m, price_us=med(lambda: p.price(*ATM, "call", return_error=True), 60))
    for second, tag in ((False, "greeks8"), (True, "greeks14")):
        p.greeks(*ATM, "call", include_second_order=second)
        row[tag + "_fused_us"] = med(lambda: p.greeks(*ATM, "call", include_second_order=second), 20)
        row[tag + "_literal_us"] = med(lambda: ol.compute_greeks_unified(p, *ATM, "call", include_second_order=second, fused=False), 7)
    row["delta"] = p.greeks(*ATM, "call", include_second_order=False)["delta"]
    print(json.dumps(row), flush=True)
