#!/usr/bin/env python3
"""One-off bias hunt: prices at 2^32 paths (se ~ 1e-4) against Black-Scholes, several step counts / contracts.
Any systematic error of the fp32 Box-Muller (hardware log2/sin/cos), the 32-bit uniform tail cap or the
reductions would show as |z| >> 3.  Usage (GPU box): python tools/deep_accuracy.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import optionslab_amd as ol  # noqa: E402
from optionslab_amd import _hip  # noqa: E402

N = 1 << 32
rows = []
for (S, K, T, r, v, q, call, M, seed) in [(100.0, 100.0, 1.0, 0.05, 0.2, 0.0, True, 16, 1), (100.0, 100.0, 1.0, 0.05, 0.2, 0.0, False, 16, 2),
                                           (100.0, 130.0, 1.0, 0.05, 0.2, 0.0, True, 4, 3), (100.0, 70.0, 2.0, 0.03, 0.4, 0.02, False, 8, 4),
                                           (100.0, 100.0, 1.0, 0.05, 0.2, 0.0, True, 1, 5), (100.0, 160.0, 1.0, 0.05, 0.2, 0.0, True, 252, 6)]:
    n = N if M <= 16 else N // 16
    st = _hip.european(S, K, T, r, v, q, call, n, M, seed, True)
    bs = ol.black_scholes(S, K, T, r, v, "call" if call else "put", q)
    # the reported std_error is the reference's naive formula (overstates the antithetic error): z is conservative
    rows.append(dict(S=S, K=K, T=T, r=r, sigma=v, q=q, call=call, n_paths=n, n_steps=M, price=st.price, bs=bs,
                     std_error=st.std_error, z=(st.price - bs) / st.std_error, rel_err=(st.price - bs) / bs))
    print(json.dumps(rows[-1]), flush=True)
