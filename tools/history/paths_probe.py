#!/usr/bin/env python3
"""Where the time of a full-path request goes (kernel + D2H vs page faults of a fresh host buffer).
Usage (GPU box): python tools/paths_probe.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import optionslab_amd as ol  # noqa: E402
from optionslab_amd import _hip  # noqa: E402

N, M = 100_000, 252
for _ in range(2):
    ol.simulate_gbm_paths_hip(100.0, 1.0, 0.05, 0.2, 0.0, N, M, 42)
t0 = time.perf_counter()
a = ol.simulate_gbm_paths_hip(100.0, 1.0, 0.05, 0.2, 0.0, N, M, 42)
t1 = time.perf_counter()
print(f"simulate_gbm_paths_hip {N} x {M} -> {a.shape} C-order, fresh numpy buffer: {1e3 * (t1 - t0):.1f} ms")
out = np.empty((N, M + 1))
t0 = time.perf_counter()
out[:] = 0
print(f"first touch of {out.nbytes / 1e6:.0f} MB on the host: {1e3 * (time.perf_counter() - t0):.1f} ms")
lib = _hip.lib()
for layout in (1, 0):
    for _ in range(3):
        t0 = time.perf_counter()
        lib.olmc_gbm_paths(100.0, 1.0, 0.05, 0.2, 0.0, N, M, 42, layout, out.ctypes.data_as(C.POINTER(C.c_double)))
        dt = time.perf_counter() - t0
    print(f"olmc_gbm_paths path_major={layout} into a resident buffer: {1e3 * dt:.2f} ms ({out.nbytes / dt / 1e9:.1f} GB/s end to end)")
