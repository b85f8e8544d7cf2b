import sys, time, statistics
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import optionslab_amd as ol
from optionslab_amd import _hip
ATM = (100.0, 100.0, 1.0, 0.05, 0.2)
p = ol.MonteCarloPricer(1_000_000, 252, 42)
for _ in range(3000): p.price(*ATM, "call")
def med(fn, n=300):
    for _ in range(30): fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return statistics.median(ts) * 1e6
print("price", med(lambda: p.price(*ATM, "call", return_error=True)))
for second in (False, True):
    print("greeks second=%s" % second, med(lambda: p.greeks(*ATM, "call", include_second_order=second)),
          "via compute_greeks_unified", med(lambda: ol.compute_greeks_unified(p, *ATM, "call", include_second_order=second)),
          "_hip lean", med(lambda: _hip.european_greeks_fd(*ATM, 0.0, True, 1_000_000, 252, 42, second, want_evals=False)),
          "_hip full", med(lambda: _hip.european_greeks_fd(*ATM, 0.0, True, 1_000_000, 252, 42, second, want_evals=True)))
