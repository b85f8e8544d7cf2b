#!/usr/bin/env python3
"""What the payoff epilogue of the fused European kernel costs (GPU box): kernel time of olmc_european_batch at 1M x 252 for sets
of 8 / 16 contracts with a chosen number of distinct vols (= base contracts, each a pair of fp64 exps) -- the slope per base,
the slope per contract and the fixed cost of the wider reduction fall out.  Usage: python tools/epilogue_probe.py"""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from optionslab_amd import _hip  # noqa: E402

N, M = 1_000_000, 252


def kernel_us(opts, reps=40):
    for i in range(30):
        _hip.european_batch(opts, N, M, i)
    ts = []
    for rep in range(5):
        _hip.profile_reset()
        for i in range(reps):
            _hip.european_batch(opts, N, M, 100 + i)
        n, ms = _hip.kernel_time()
        ts.append(ms / n * 1e3)
    return statistics.median(ts)


_hip.lib()
for i in range(2000):
    _hip.european(100.0, 100.0, 1.0, 0.05, 0.2, 0.0, True, N, M, i, True)
_hip.profile_enable(True)
rows = []
one = kernel_us([(100.0, 100.0, 1.0, 0.05, 0.2, 0.0, True)])
print(json.dumps({"contracts": 1, "bases": 1, "kernel_us": one}), flush=True)
for k in (8, 16):
    for bases in sorted({1, 2, 4, k // 2, k}):
        # `bases` distinct sigmas, the other contracts are S-bumps of them (same vol: scaled prices)
        opts = [(100.0 + 0.01 * (j // bases), 100.0, 1.0, 0.05, 0.2 + 0.01 * (j % bases), 0.0, True) for j in range(k)]
        us = kernel_us(opts)
        print(json.dumps({"contracts": k, "distinct_vols": bases, "kernel_us": us, "over_one_contract_us": us - one}), flush=True)
