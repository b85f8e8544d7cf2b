#!/usr/bin/env python3
"""Short-path launches (the reference's default num_steps = 1): kernel and wall time by path count, step
count and grid cap (workgroups per launch; 0 = one per 256 paths).  Usage (GPU box): python tools/single_step_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from optionslab_amd import _hip  # noqa: E402

_hip.lib()
_hip.profile_enable(True)
for N in (1_000_000, 8_000_000):
    for M in (1, 4, 32, 128, 252):
        row = []
        for cap in (0, 4096, 1 << 18):
            _hip.tune(_hip.TUNE_GRID_CAP, cap)
            for _ in range(3):
                _hip.european(100., 100., 1., .05, .2, 0., True, N, M, 1)
            _hip.profile_reset()
            t = time.perf_counter()
            for i in range(20):
                _hip.european(100., 100., 1., .05, .2, 0., True, N, M, i)
            wall = (time.perf_counter() - t) / 20
            n, ms = _hip.kernel_time()
            row.append(f"cap {cap:5d}: {ms / n * 1e3:6.1f}/{wall * 1e6:6.1f}")
        print(f"N={N:8d} M={M:3d}  kernel/wall us  " + "  ".join(row), flush=True)
_hip.tune(_hip.TUNE_GRID_CAP, 0)
