#!/usr/bin/env python3
"""Interleaved A/B of whole libolmc builds: one subprocess per (library, round), each timing the
European path kernel with HIP events (olmc_kernel_time).  Usage (GPU box):
    python tools/ab_libs.py libA.so libB.so ... [--n 1000000] [--m 252] [--rounds 7]"""
import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import json, sys
sys.path.insert(0, %r)
from optionslab_amd import _hip
N, M = int(sys.argv[1]), int(sys.argv[2])
_hip.lib(); _hip.profile_enable(True)
for i in range(5): _hip.european(100.0, 100.0, 1.0, 0.05, 0.2, 0.0, True, N, M, 1 + i, True)
_hip.profile_reset()
for i in range(40): st = _hip.european(100.0, 100.0, 1.0, 0.05, 0.2, 0.0, True, N, M, 42 + i, True)
n, ms = _hip.kernel_time()
print(json.dumps({"us": ms / n * 1e3, "price": st.price}))
""" % ROOT

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--m", type=int, default=252)
ap.add_argument("--rounds", type=int, default=7)
a = ap.parse_args()
res = {l: [] for l in a.libs}
price = {}
for r in range(a.rounds):
    for l in a.libs:
        env = dict(os.environ, OLMC_LIBRARY=os.path.abspath(l))
        out = subprocess.run([sys.executable, "-c", CHILD, str(a.n), str(a.m)], env=env, capture_output=True, text=True, check=True)
        d = json.loads(out.stdout.strip().splitlines()[-1])
        res[l].append(d["us"])
        price[l] = d["price"]
for l in a.libs:
    v = res[l]
    print(f"{os.path.basename(l):28s} median {statistics.median(v):8.2f} us  min {min(v):8.2f}  max {max(v):8.2f}  price {price[l]:.9f}", flush=True)
