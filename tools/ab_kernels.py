#!/usr/bin/env python3
"""Interleaved A/B timing of launch shapes of the European kernel on ONE device in ONE process
(cdna_hip_programming.md rule 24).  Kernel time comes from HIP events around the path kernel.
Usage (GPU box): python tools/ab_kernels.py [n_paths] [n_steps] [rounds]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optionslab_amd import _hip  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 252
ROUNDS = int(sys.argv[3]) if len(sys.argv) > 3 else 15
VARIANTS = {          # name -> grid cap (0 = one workgroup per 256 paths, dispatcher-balanced)
    "grid<=2048": 2048,
    "grid<=4096": 4096,
    "grid=tiles": 0,
}


def measure(cap, reps=10):
    _hip.tune(_hip.TUNE_GRID_CAP, cap)
    _hip.profile_reset()
    for i in range(reps):
        st = _hip.european(100.0, 100.0, 1.0, 0.05, 0.2, 0.0, True, N, M, 42 + i, True)
    n, ms = _hip.kernel_time()
    return ms / n * 1e3, st.price


_hip.lib()
_hip.profile_enable(True)
for c in VARIANTS.values():
    measure(c, 3)
res = {k: [] for k in VARIANTS}
prices = {}
for r in range(ROUNDS):
    for k, c in VARIANTS.items():
        us, price = measure(c)
        res[k].append(us)
        prices[k] = price
print(f"European {N} x {M}, kernel us (median / min over {ROUNDS} rounds of 10 launches)")
for k, v in res.items():
    print(f"  {k:18s} median {statistics.median(v):8.2f}  min {min(v):8.2f}   price {prices[k]!r}")
