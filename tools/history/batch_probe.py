import sys, time, statistics, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import optionslab_amd as ol
from optionslab_amd import _hip
uni = ol.MonteCarloPricerUni(100_000, 100, 42)
rng = np.random.default_rng(0)
for _ in range(500): uni.price(100.0, 100.0, 1.0, 0.05, 0.2, "call")
for nb in (1, 4, 16, 64, 256, 1024):
    S, K = rng.uniform(80, 120, nb), rng.uniform(80, 120, nb)
    T, r, v = rng.uniform(0.25, 2, nb), rng.uniform(0, 0.08, nb), rng.uniform(0.1, 0.5, nb)
    fn = lambda: uni.price_batch(S, K, T, r, v, "call", 0.01)
    for _ in range(20): fn()
    ts = []
    for _ in range(40):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    _hip.profile_enable(True); _hip.profile_reset()
    for _ in range(10): fn()
    n, ms = _hip.kernel_time(); _hip.profile_enable(False)
    fn2 = lambda: uni.delta_gamma_batch(S, K, T, r, v, "call", 0.01)
    for _ in range(5): fn2()
    t2 = []
    for _ in range(20):
        t0 = time.perf_counter(); fn2(); t2.append(time.perf_counter() - t0)
    print(json.dumps(dict(contracts=nb, price_batch_wall_us=statistics.median(ts) * 1e6, bracketed_device_us=ms / n * 1e3, delta_gamma_batch_wall_us=statistics.median(t2) * 1e6)), flush=True)
