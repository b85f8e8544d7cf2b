#!/usr/bin/env python3
"""How long does a ROUND of k whole-path workgroups per compute unit take?  Launches exactly k x CUs workgroups of the European
kernel (256 paths x M steps each, split workgroups off) for k = 1 .. 2 x occupancy and prints the kernel time: the curve that
decides when a thin last round is worth handing to split workgroups (european_launch_shape, OLMC_TUNE_SPLIT_SAT).
Usage (GPU box): python tools/occupancy_probe.py [n_steps]"""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from optionslab_amd import _hip  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 252
P = (100.0, 100.0, 1.0, 0.05, 0.2, 0.0)
cus = _hip.device_info()["compute_units"]
for i in range(2000):
    _hip.european(*P, True, 1_000_000, 252, i, True)
_hip.tune(_hip.TUNE_SPLIT_TAIL, -1)
_hip.profile_enable(True)
rows = []
for k in list(range(1, 17)) + [21, 28]:
    n = 256 * cus * k
    for i in range(30):
        _hip.european(*P, True, n, M, i, True)
    ts = []
    for rep in range(5):
        _hip.profile_reset()
        for i in range(20):
            _hip.european(*P, True, n, M, 100 + i, True)
        launches, ms = _hip.kernel_time()
        ts.append(ms / launches * 1e3)
    rows.append({"workgroups_per_cu": k, "n_paths": n, "n_steps": M, "kernel_us": statistics.median(ts)})
    print(json.dumps(rows[-1]), flush=True)
