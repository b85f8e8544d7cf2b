#!/usr/bin/env python3
"""Times every BASELINE.json config shape through the public API on one GPU (blocking calls,
result on host) and prints one JSON object per line.  Usage (GPU box): python tools/measure_configs.py"""
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import optionslab_amd as ol  # noqa: E402
from optionslab_amd import _hip  # noqa: E402

ATM = (100.0, 100.0, 1.0, 0.05, 0.2)
BS = ol.black_scholes(*ATM, "call")


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts), min(ts), out


def kernel_us(fn, reps=10):
    _hip.profile_enable(True)
    _hip.profile_reset()
    for _ in range(reps):
        fn()
    n, ms = _hip.kernel_time()
    _hip.profile_enable(False)
    return ms / max(n, 1) * 1e3, n // reps


def report(name, path_steps, fn, extra=None, reps=20):
    med, best, out = timeit(fn, reps)
    kus, launches = kernel_us(fn)
    row = dict(config=name, path_steps=path_steps, wall_ms_median=med * 1e3, wall_ms_min=best * 1e3,
               path_steps_per_s=path_steps / med, kernel_us=kus, kernel_launches_per_call=launches)
    if extra:
        row.update(extra(out))
    print(json.dumps(row), flush=True)


def main():
    print(json.dumps(dict(device=_hip.device_info())), flush=True)
    N, M = 1_000_000, 252
    p = ol.MonteCarloPricer(N, M, 42)
    report("C2 european call 1M x 252 antithetic, price(return_error=True)", N * M,
           lambda: p.price(*ATM, "call", return_error=True),
           lambda r: dict(price=r.price, std_error=r.std_error, z_vs_bs=(r.price - BS) / r.std_error))
    p1 = ol.MonteCarloPricer(N, 1, 42)
    report("single-step european 1M x 1 (reference default num_steps=1)", N, lambda: p1.price(*ATM, "call", return_error=True),
           lambda r: dict(price=r.price, std_error=r.std_error))
    report("C3 greeks fused first-order (8 contracts, 1 RNG pass) 1M x 252", N * M, lambda: p.greeks(*ATM, "call", include_second_order=False),
           lambda g: dict(greeks=dict(g)))
    report("C3 greeks fused second-order (14 contracts, 1 RNG pass) 1M x 252", N * M, lambda: p.greeks(*ATM, "call", include_second_order=True),
           lambda g: dict(greeks=dict(g)))
    report("C3 greeks literal bump-and-reprice (8 launches, common Philox key) 1M x 252", 8 * N * M,
           lambda: ol.compute_greeks_unified(p, *ATM, "call", include_second_order=False, fused=False), lambda g: dict(greeks=dict(g)), reps=10)
    report("control variate 1M x 252", N * M, lambda: p.price_with_control_variate(*ATM, "call"), lambda v: dict(value=v))
    report("terminal array 1M x 252 -> 2M fp64 on host (16 MB D2H included)", N * M, lambda: p._simulate(100.0, 1.0, 0.05, 0.2, 0.0),
           lambda a: dict(mean_terminal=float(a.mean())), reps=10)
    a = ol.AsianOption(*ATM, seed=42)
    MA = 1024
    report("C4 arithmetic Asian call 1M x 1024, no antithetic (reference semantics)", N * MA,
           lambda: a.price(N, MA, "arithmetic", "call", return_error=True), lambda r: dict(price=float(r[0]), std_error=r[1]), reps=10)
    report("C4 arithmetic Asian call 1M x 1024, antithetic", N * MA,
           lambda: a.price(N, MA, "arithmetic", "call", antithetic=True, return_error=True), lambda r: dict(price=float(r[0]), std_error=r[1]), reps=10)
    report("geometric Asian call 1M x 1024", N * MA,
           lambda: a.price(N, MA, "geometric", "call", return_error=True), lambda r: dict(price=float(r[0]), std_error=r[1], closed_form=a.price_geometric_closed_form("call")), reps=10)
    p8 = ol.MonteCarloPricer(8_000_000, M, 42)
    report("C5 per-GPU shard: european call 8M x 252", 8_000_000 * M, lambda: p8.price(*ATM, "call", return_error=True),
           lambda r: dict(price=r.price, std_error=r.std_error, z_vs_bs=(r.price - BS) / r.std_error), reps=10)
    p64 = ol.MonteCarloPricer(64_000_000, M, 42)
    report("C5 total on ONE GPU: european call 64M x 252", 64_000_000 * M, lambda: p64.price(*ATM, "call", return_error=True),
           lambda r: dict(price=r.price, std_error=r.std_error, z_vs_bs=(r.price - BS) / r.std_error), reps=3)


if __name__ == "__main__":
    main()
