#!/usr/bin/env python3
"""Per-class VALU issue cost on this device (olmc_issue_probe) at 1, 2, 4 and 8 waves per SIMD, with the shader clock held
under the headline kernel's load beside it.  Usage (GPU box): python tools/issue_probe.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optionslab_amd import _hip  # noqa: E402
from tools.probe import binding as probe  # noqa: E402  (the instrumented build: include/olmc_probe.h)

for _ in range(3000):
    _hip.european(100.0, 100.0, 1.0, 0.05, 0.2, 0.0, True, 1_000_000, 252, 1)
print(json.dumps(dict(clock=probe.clock_probe())))
for w in (1, 2, 4, 8):
    print(json.dumps(dict(waves_per_simd=w, ns={k: round(v, 4) for k, v in probe.issue_probe(w).items()})), flush=True)
print(json.dumps(dict(clock=probe.clock_probe())))
