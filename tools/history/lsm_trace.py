#!/usr/bin/env python3
"""One American LSM pricing (reference defaults 50k x 50, degree 3) for a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/lsm_trace.py [n_paths] [n_steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from optionslab_amd import _hip  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 50
for i in range(3):
    _hip.american_lsm(100.0, 100.0, 1.0, 0.05, 0.2, 0.0, False, N, M, 3, 40 + i)
t0 = time.perf_counter()
st = _hip.american_lsm(100.0, 100.0, 1.0, 0.05, 0.2, 0.0, False, N, M, 3, 42)
print(f"{N} x {M}: {1e3 * (time.perf_counter() - t0):.3f} ms  price {st.price:.6f}")
