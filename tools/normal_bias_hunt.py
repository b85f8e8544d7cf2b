#!/usr/bin/env python3
"""Moments of 2^36 normals of the device stream (validation tap olmc_normal_moments) against N(0,1)."""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optionslab_amd import _hip
from tools.probe import binding as probe  # noqa: E402  (the instrumented build: include/olmc_probe.h)
for seed, n_paths, n_steps in [(1, 1 << 28, 256), (2, 1 << 28, 256), (3, 1 << 30, 4)]:
    s1, s2, s3, s4 = probe.normal_moments(seed, n_paths, n_steps)
    n = n_paths * n_steps
    m = [s1 / n, s2 / n, s3 / n, s4 / n]
    sd = [1 / math.sqrt(n), math.sqrt(2 / n), math.sqrt(15 / n), math.sqrt(96 / n)]
    want = [0, 1, 0, 3]
    print(json.dumps(dict(seed=seed, n=n, mean=m[0], var=m[1], m3=m[2], m4=m[3], z=[(m[k] - want[k]) / sd[k] for k in range(4)])), flush=True)
