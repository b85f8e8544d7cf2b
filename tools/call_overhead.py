#!/usr/bin/env python3
"""Where a blocking price() call's wall time goes (GPU box): the device-side kernel duration (dispatch-attached HIP
events) against the wall of the C-ABI call and of the Python API call, at the headline size and at a tiny size whose
kernel is ~5 us (so the wall IS the fixed launch + completion + binding cost).  Usage: python tools/call_overhead.py"""
import ctypes as C
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import optionslab_amd as ol  # noqa: E402
from optionslab_amd import _hip  # noqa: E402

ATM = (100.0, 100.0, 1.0, 0.05, 0.2)


def med(fn, reps=300, warm=50):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts) * 1e6, min(ts) * 1e6


def main():
    lib = _hip.lib()
    for _ in range(2000):                       # clocks up
        _hip.european(*ATM, 0.0, True, 1_000_000, 252, 1)
    out = C.byref(_hip.Stats())
    rows = []
    for n, m in ((256, 1), (65_536, 16), (1_000_000, 252), (1_000_000, 1024), (8_000_000, 252), (64_000_000, 252)):
        reps = 300 if n * m <= 252_000_000 else (40 if n <= 8_000_000 else 12)
        p = ol.MonteCarloPricer(n, m, 42)
        api = med(lambda: p.price(*ATM, "call", return_error=True), reps)
        shim = med(lambda: _hip.european(*ATM, 0.0, True, n, m, 42), reps)
        raw = med(lambda: lib.olmc_european(100.0, 100.0, 1.0, 0.05, 0.2, 0.0, 1, n, m, 42, 1, out), reps)
        _hip.profile_enable(True)
        _hip.profile_reset()
        for _ in range(50):
            lib.olmc_european(100.0, 100.0, 1.0, 0.05, 0.2, 0.0, 1, n, m, 42, 1, out)
        k, ms = _hip.kernel_time()
        _hip.profile_enable(False)
        cpu0, thr0 = time.process_time(), time.thread_time()
        wall0 = time.perf_counter()
        for _ in range(20 if n <= 8_000_000 else 6):
            lib.olmc_european(100.0, 100.0, 1.0, 0.05, 0.2, 0.0, 1, n, m, 42, 1, out)
        # CPU seconds burnt per second of waiting: 1.0 = a core spinning for the whole call, ~0 = asleep.  calling thread / whole process
        wall = max(time.perf_counter() - wall0, 1e-9)
        cpu_share, proc_share = (time.thread_time() - thr0) / wall, (time.process_time() - cpu0) / wall
        rows.append(dict(paths=n, steps=m, calling_thread_cpu_share=cpu_share, process_cpu_share=proc_share, kernel_us=ms / k * 1e3, c_abi_call_us_median=raw[0], c_abi_call_us_min=raw[1],
                         ctypes_shim_us_median=shim[0], python_api_us_median=api[0], python_api_us_min=api[1],
                         host_overhead_us=raw[0] - ms / k * 1e3))
        print(json.dumps(rows[-1]), flush=True)
    noop = med(lambda: lib.olmc_abi_version(), 2000)
    print(json.dumps(dict(ctypes_noop_call_us=noop[0])), flush=True)


if __name__ == "__main__":
    main()
