#!/usr/bin/env python3
"""Concurrent callers: N Python threads, each its own MonteCarloPricer, each calling price() back to back at the
sizes the one live UI caller uses (streamlit_app/pages/1_MonteCarlo_Basic.py:111-126).  ctypes releases the GIL for the duration
of a call, so the threads really are concurrent inside libolmc.so, where each leases a context of its own (round 4; round 3 held
one mutex from launch to result).  Prints one JSON line per (size, threads): whole-process calls/s, per-call latency p50 / p99.

    python tools/thread_bench.py [--seconds 2.0] [--lib path]      (GPU box)
"""
import argparse
import json
import os
import statistics
import sys
import threading
import time

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=2.0)
ap.add_argument("--lib", default=None, help="another libolmc build to measure (e.g. tools/ab/libolmc_r03.so)")
a = ap.parse_args()
if a.lib:
    os.environ["OLMC_LIBRARY"] = os.path.abspath(a.lib)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import optionslab_amd as ol  # noqa: E402
from optionslab_amd import _hip  # noqa: E402

ATM = (100.0, 100.0, 1.0, 0.05, 0.2)
info = _hip.device_info()
for N, M in ((10_000, 50), (100_000, 100), (200_000, 1), (1_000_000, 252)):
    single = None
    for n_threads in (1, 2, 4, 8, 16):
        lat = [[] for _ in range(n_threads)]
        prices = [None] * n_threads
        go = threading.Barrier(n_threads + 1)
        stop = threading.Event()

        def work(k):
            p = ol.MonteCarloPricer(N, M, 42)
            for _ in range(50):
                p.price(*ATM, "call")
            go.wait()
            mine = lat[k]
            while not stop.is_set():
                t0 = time.perf_counter()
                prices[k] = p.price(*ATM, "call")
                mine.append(time.perf_counter() - t0)

        ts = [threading.Thread(target=work, args=(k,)) for k in range(n_threads)]
        [t.start() for t in ts]
        go.wait()
        t0 = time.perf_counter()
        time.sleep(a.seconds)
        stop.set()
        [t.join() for t in ts]
        wall = time.perf_counter() - t0
        all_lat = sorted(x for l in lat for x in l)
        calls = len(all_lat)
        rec = {"library": os.path.basename(_hip.LIBRARY_PATH), "n_paths": N, "n_steps": M, "threads": n_threads, "calls_per_s": round(calls / wall, 1),
               "latency_us_p50": round(1e6 * all_lat[calls // 2], 2), "latency_us_p99": round(1e6 * all_lat[min(calls - 1, int(calls * 0.99))], 2),
               "latency_us_mean": round(1e6 * statistics.fmean(all_lat), 2), "path_steps_per_s": calls / wall * N * M,
               "same_price_in_every_thread": len(set(prices)) == 1, "host_cpus": os.cpu_count(), "device": info["name"]}
        if n_threads == 1:
            single = rec["calls_per_s"]
        rec["throughput_vs_one_thread"] = round(rec["calls_per_s"] / single, 3)
        print(json.dumps(rec), flush=True)
