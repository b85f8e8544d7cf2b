#!/usr/bin/env python3
"""Generate tests/golden/reference_vectors.json by RUNNING the reference.

Run only in the build container, where /root/reference is mounted read-only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference never travels: this script imports it in place, records inputs
and outputs (data only) and writes a small JSON fixture.  `import src` the
normal way fails here (its package __init__ files pull numba / streamlit,
which are absent), so empty parent packages are registered first and only the
hot-path modules -- which need NumPy/SciPy alone -- are imported.
"""

import json
import os
import sys
import types

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.json")


def load_reference():
    for name, rel in (("src", "src"), ("src.pricing_models", "src/pricing_models"), ("src.greeks", "src/greeks")):
        pkg = types.ModuleType(name)
        pkg.__path__ = [os.path.join(REF, rel)]
        sys.modules[name] = pkg
    sys.path.insert(0, REF)
    from src.greeks.unified_greeks import ExoticAdapter, compute_greeks_unified
    from src.pricing_models.black_scholes import black_scholes
    from src.pricing_models.exotic_options import (AmericanOption, AsianOption, AutocallableOption, BarrierOption, CliquetOption,
                                                    LookbackOption, price_asian, price_barrier)
    from src.pricing_models.heston import HestonPricer
    from src.simulation.gbm_numpy import simulate_gbm_paths
    from src.simulation.gbm_qmc import simulate_gbm_qmc_antithetic
    from src.pricing_models.jump_diffusion import KouJumpDiffusion, MertonJumpDiffusion
    from src.pricing_models.monte_carlo import MCMethod, MonteCarloPricer
    from src.pricing_models.monte_carlo_unified import MonteCarloPricerUni

    return dict(simulate_gbm_qmc_antithetic=simulate_gbm_qmc_antithetic, simulate_gbm_paths=simulate_gbm_paths, MertonJumpDiffusion=MertonJumpDiffusion, KouJumpDiffusion=KouJumpDiffusion, HestonPricer=HestonPricer, MonteCarloPricerUni=MonteCarloPricerUni, BarrierOption=BarrierOption, LookbackOption=LookbackOption, AutocallableOption=AutocallableOption,
                CliquetOption=CliquetOption, AmericanOption=AmericanOption,
                price_barrier=price_barrier, MonteCarloPricer=MonteCarloPricer, MCMethod=MCMethod, black_scholes=black_scholes,
                AsianOption=AsianOption, price_asian=price_asian,
                compute_greeks_unified=compute_greeks_unified, ExoticAdapter=ExoticAdapter)


ATM = dict(S=100.0, K=100.0, T=1.0, r=0.05, sigma=0.2)


def main():
    import numpy as np
    import scipy

    ref = load_reference()
    MCP, MCMethod = ref["MonteCarloPricer"], ref["MCMethod"]
    bs = ref["black_scholes"]
    doc = {
        "generator": "tests/golden/make_golden.py",
        "reference": "Diegotistical/OptionsLab @ /root/reference (snapshot 2026-01-30)",
        "numpy": np.__version__,
        "scipy": scipy.__version__,
    }

    # -- Black-Scholes ------------------------------------------------------
    doc["black_scholes"] = []
    for S, K, T, r, v, q in [(100, 100, 1.0, 0.05, 0.2, 0.0), (110, 100, 1.0, 0.05, 0.2, 0.0),
                             (90, 100, 1.0, 0.05, 0.2, 0.0), (100, 100, 0.25, 0.05, 0.2, 0.0),
                             (100, 100, 2.0, 0.05, 0.4, 0.02), (100, 95, 0.5, 0.03, 0.3, 0.02),
                             (100, 100, 0.0, 0.05, 0.2, 0.0), (120, 100, 0.0, 0.05, 0.2, 0.0)]:
        doc["black_scholes"].append(dict(args=[S, K, T, r, v, q],
                                         call=float(bs(S, K, T, r, v, "call", q)),
                                         put=float(bs(S, K, T, r, v, "put", q))))

    # -- pricer.price -------------------------------------------------------
    cases = []

    def add_price(N, M, seed, method, S, K, T, r, sigma, typ, q=0.0, call_seed=None):
        v = sigma
        p = MCP(N, M, seed, getattr(MCMethod, method))
        kw = {} if call_seed is None else {"seed": call_seed}
        res = p.price(S, K, T, r, v, typ, q, return_error=True, **kw)
        plain = p.price(S, K, T, r, v, typ, q, **kw)
        assert plain == res.price
        st = p._simulate(S, T, r, v, q, call_seed)
        cases.append(dict(ctor=[N, M, seed, method.lower()], args=[S, K, T, r, v, typ, q], call_seed=call_seed,
                          price=res.price, std_error=res.std_error, n_paths=int(res.n_paths),
                          plain_is_float=type(plain) is float,
                          terminal_head=[float(x) for x in st[:4]],
                          terminal_mid=[float(x) for x in st[len(st) // 2: len(st) // 2 + 4]],
                          terminal_len=int(len(st))))

    for typ in ("call", "put"):
        add_price(10000, 50, 42, "NUMPY", typ=typ, **ATM)                 # G1/G2
        add_price(10000, 50, 42, "NUMPY", typ=typ, q=0.02, **ATM)         # G9
        add_price(100000, 1, 42, "NUMPY", typ=typ, **ATM)                 # G5
        add_price(100000, 252, 42, "NUMPY", typ=typ, **ATM)               # G3 (config 1)
    add_price(1000000, 252, 42, "NUMPY", typ="call", **ATM)               # G4
    add_price(10000, 50, 42, "NUMBA", typ="call", **ATM)                  # numba absent => numpy path
    add_price(10000, 50, 7, "FAST", typ="call", **ATM)                    # FAST forces single step
    add_price(10000, 50, 42, "NUMPY", typ="call", call_seed=123, **ATM)   # per-call seed override
    add_price(2**14, 16, 42, "QMC", typ="call", **ATM)                    # G10
    add_price(2**12, 8, 5, "QMC", typ="put", **ATM)
    for S, K, T, r, v in [(110, 100, 1.0, 0.05, 0.2), (90, 100, 1.0, 0.05, 0.2), (100, 100, 1.0, 0.05, 0.1),
                          (100, 100, 1.0, 0.05, 0.4), (100, 100, 0.25, 0.05, 0.2), (100, 100, 2.0, 0.05, 0.2)]:
        add_price(10000, 50, 42, "NUMPY", S, K, T, r, v, "call")          # tests/test_monte_carlo.py:523-546
    add_price(7, 3, 1, "NUMPY", typ="call", **ATM)                        # tiny ragged case
    add_price(1, 1, 0, "NUMPY", typ="put", **ATM)                         # minimum size
    doc["price"] = cases

    # -- T <= 0 early-out (monte_carlo.py:133-135) ---------------------------
    p = MCP(1000, 10, 1)
    r0 = p.price(120, 100, 0.0, 0.05, 0.2, "call", return_error=True)
    doc["expired"] = [dict(args=[120, 100, 0.0, 0.05, 0.2, "call"], price=r0.price, std_error=r0.std_error,
                           n_paths=r0.n_paths, plain=p.price(120, 100, 0.0, 0.05, 0.2, "call")),
                      dict(args=[80, 100, -1.0, 0.05, 0.2, "put"], price=p.price(80, 100, -1.0, 0.05, 0.2, "put"),
                           std_error=0.0, n_paths=0, plain=p.price(80, 100, -1.0, 0.05, 0.2, "put"))]

    # -- control variate (G8) ----------------------------------------------
    doc["control_variate"] = []
    for N, M, seed, typ, q in [(100000, 252, 42, "call", 0.0), (10000, 50, 42, "put", 0.02), (100000, 1, 42, "call", 0.0)]:
        doc["control_variate"].append(dict(ctor=[N, M, seed, "numpy"], args=[100.0, 100.0, 1.0, 0.05, 0.2, typ, q],
                                           value=MCP(N, M, seed).price_with_control_variate(100.0, 100.0, 1.0, 0.05, 0.2, typ, q)))

    # -- unified FD Greeks (G6/G7) -----------------------------------------
    doc["greeks"] = []
    cgu = ref["compute_greeks_unified"]
    for N, M, seed, typ, q, second in [(100000, 1, 42, "call", 0.0, False), (100000, 252, 42, "call", 0.0, False),
                                       (100000, 252, 42, "call", 0.0, True), (20000, 50, 42, "put", 0.02, True)]:
        g = cgu(MCP(N, M, seed), 100.0, 100.0, 1.0, 0.05, 0.2, typ, q, include_second_order=second)
        doc["greeks"].append(dict(ctor=[N, M, seed, "numpy"], args=[100.0, 100.0, 1.0, 0.05, 0.2, typ, q],
                                  include_second_order=second, keys=list(g.keys()),
                                  values={k: float(x) for k, x in g.items()}))
    # short-dated branch T <= h_T (unified_greeks.py:313-314, 353-354)
    g = cgu(MCP(20000, 4, 42), 100.0, 100.0, 0.002, 0.05, 0.2, "call", 0.0, include_second_order=True)
    doc["greeks"].append(dict(ctor=[20000, 4, 42, "numpy"], args=[100.0, 100.0, 0.002, 0.05, 0.2, "call", 0.0],
                              include_second_order=True, keys=list(g.keys()), values={k: float(x) for k, x in g.items()}))

    # -- FD Greeks of a MCMethod.QMC pricer: every bumped contract on the same scrambled-Sobol points (gbm_qmc.py:14-46 under
    #    unified_greeks.py:280-358) -- the reference values of the device's one-launch form (olmc_european_qmc_greeks_fd)
    doc["qmc_greeks"] = []
    for N, M, seed, typ, q, second in [(2**14, 16, 42, "call", 0.0, False), (2**14, 16, 42, "call", 0.0, True),
                                       (2**12, 252, 7, "put", 0.02, True), (1000, 5, 3, "call", 0.0, True)]:
        g = cgu(MCP(N, M, seed, MCMethod.QMC), 100.0, 100.0, 1.0, 0.05, 0.2, typ, q, include_second_order=second)
        doc["qmc_greeks"].append(dict(ctor=[N, M, seed, "qmc"], args=[100.0, 100.0, 1.0, 0.05, 0.2, typ, q],
                                      include_second_order=second, keys=list(g.keys()), values={k: float(x) for k, x in g.items()}))

    # -- Asian (G11/G12) ----------------------------------------------------
    doc["asian"] = []
    A = ref["AsianOption"]
    for n, m, seed, avg, typ, q in [(100000, 252, 42, "arithmetic", "call", 0.0), (100000, 252, 42, "geometric", "call", 0.0),
                                    (10000, 252, 42, "arithmetic", "put", 0.0), (10000, 64, 7, "arithmetic", "call", 0.02),
                                    (10000, 64, 7, "geometric", "put", 0.02), (1000, 252, 42, "arithmetic", "call", 0.0)]:
        o = A(S=100.0, K=100.0, T=1.0, r=0.05, sigma=0.2, q=q, seed=seed)
        doc["asian"].append(dict(params=[100.0, 100.0, 1.0, 0.05, 0.2, q], seed=seed, n_paths=n, n_steps=m, avg_type=avg,
                                 option_type=typ, price=float(o.price(n, m, avg, typ)),
                                 geometric_closed_form=float(o.price_geometric_closed_form(typ))))
    # -- BASELINE configs[3]'s own shape: 1024 monitoring dates (the device's 1,000,000 x 1024 run is compared with THESE, not with
    #    a 252-date cousin).  100,000 x 1024 is what fits here: Z, log-returns, log S and exp are 0.82 GB each (exotic_options.py:59-67).
    #    The reference returns no standard error for an Asian, so one is recorded from the reference's OWN paths (`_generate_paths`
    #    re-seeds, so the second call walks the paths `price` just used -- asserted): disc * std(payoffs) / sqrt(n), ddof = 0 as
    #    monte_carlo.py:147-149 takes it.
    doc["asian_c4_shape"] = []
    for n, m, seed, avg, typ in [(100000, 1024, 42, "arithmetic", "call"), (100000, 1024, 42, "arithmetic", "put"),
                                 (100000, 1024, 42, "geometric", "call")]:
        o = A(S=100.0, K=100.0, T=1.0, r=0.05, sigma=0.2, q=0.0, seed=seed)
        price = float(o.price(n, m, avg, typ))
        paths = o._generate_paths(n, m)
        avg_price = np.mean(paths[:, 1:], axis=1) if avg == "arithmetic" else np.exp(np.mean(np.log(paths[:, 1:]), axis=1))
        del paths
        payoffs = np.maximum(avg_price - 100.0, 0) if typ == "call" else np.maximum(100.0 - avg_price, 0)
        assert float(np.exp(-0.05 * 1.0) * np.mean(payoffs)) == price
        doc["asian_c4_shape"].append(dict(params=[100.0, 100.0, 1.0, 0.05, 0.2, 0.0], seed=seed, n_paths=n, n_steps=m, avg_type=avg,
                                          option_type=typ, price=price,
                                          std_error_from_reference_paths=float(np.exp(-0.05 * 1.0) * np.std(payoffs) / np.sqrt(n)),
                                          geometric_closed_form=float(o.price_geometric_closed_form(typ))))
    doc["price_asian_helper"] = dict(args=[100.0, 100.0, 1.0, 0.05, 0.2, "arithmetic", "call", 20000, 42],
                                     value=float(ref["price_asian"](100.0, 100.0, 1.0, 0.05, 0.2, "arithmetic", "call", 20000, 42)))
    # ExoticAdapter Greeks on an Asian (unified_greeks.py:177-227)
    ad = ref["ExoticAdapter"](A(S=100.0, K=100.0, T=1.0, r=0.05, sigma=0.2, seed=42), n_paths=20000, n_steps=64, avg_type="arithmetic")
    g = cgu(ad, 100.0, 100.0, 1.0, 0.05, 0.2, "call", 0.0, include_second_order=False)
    doc["asian_greeks"] = dict(n_paths=20000, n_steps=64, seed=42, values={k: float(x) for k, x in g.items()})

    # -- ExoticAdapter Greeks of the payoffs the device prices in ONE fused launch (unified_greeks.py:177-227 over exotic_options.py:97-131,
    #    163-224, 347-401), each Greek with a standard error from the reference's OWN per-path payoffs, so that the device -- at many
    #    more paths -- is held to 3 sigma of the reference run instead of to a guessed absolute tolerance.  The payoffs are not restated
    #    here: every exotic price() ends in `np.exp(-r T) * np.mean(payoffs)`, and the array that last, axis-free np.mean receives IS the
    #    per-path payoff vector (asserted: its discounted mean is the returned price, bit for bit).  All bumps of one option walk the same
    #    normals (legacy np.random.seed(self.seed) in _generate_paths), so the per-path finite differences are the CRN differences.
    class PayoffSpy:
        def __enter__(self):
            self.real, self.last = np.mean, None

            def spy(a, *args, **kw):
                if not args and not kw and getattr(a, "ndim", 0) == 1:
                    self.last = np.array(a, dtype=np.float64)
                return self.real(a, *args, **kw)
            np.mean = spy
            return self

        def __exit__(self, *exc):
            np.mean = self.real

    def adapter_greeks_with_errors(make_option, n_paths, n_steps, typ, second, **kw):
        S, K, T, r, v, q = 100.0, 100.0, 1.0, 0.05, 0.2, 0.0
        ad = ref["ExoticAdapter"](make_option(), n_paths=n_paths, n_steps=n_steps, **kw)
        g = cgu(ad, S, K, T, r, v, typ, q, include_second_order=second)
        h_S, h_v, h_r, h_T = max(1e-4, 0.01 * S), 0.01, 1e-4, 1 / 365.0            # unified_greeks.py:274-277

        def pv(S_=S, T_=T, r_=r, v_=v):                                               # discounted per-path payoffs of one evaluation
            with PayoffSpy() as spy:
                price = ad.price(S_, K, T_, r_, v_, typ, q)
            assert spy.last is not None and len(spy.last) == n_paths and float(np.exp(-r_ * T_) * spy.real(spy.last)) == float(price)
            return np.exp(-r_ * T_) * spy.last

        mid, su, sd = pv(), pv(S_=S + h_S), pv(S_=S - h_S)
        vu, vd, td, ru, rd = pv(v_=v + h_v), pv(v_=v - h_v), pv(T_=T - h_T), pv(r_=r + h_r), pv(r_=r - h_r)
        per_path = {"price": mid, "delta": (su - sd) / (2 * h_S), "gamma": (su - 2 * mid + sd) / h_S**2, "vega": (vu - vd) / (2 * h_v),
                    "theta": (td - mid) / h_T, "rho": (ru - rd) / (2 * h_r)}
        if second:
            uu, ud = pv(S_=S + h_S, v_=v + h_v), pv(S_=S + h_S, v_=v - h_v)
            du, dd = pv(S_=S - h_S, v_=v + h_v), pv(S_=S - h_S, v_=v - h_v)
            ut, dt = pv(S_=S + h_S, T_=T - h_T), pv(S_=S - h_S, T_=T - h_T)
            per_path["vanna"] = (uu - ud - du + dd) / (4 * h_S * h_v)
            per_path["charm"] = ((ut - dt) / (2 * h_S) - per_path["delta"]) / h_T
            per_path["vomma"] = (vu - 2 * mid + vd) / h_v**2
        for k_, x in per_path.items():          # the per-path differences average to the reference's own Greeks (another association of the same sums)
            assert abs(float(x.mean()) - float(g[k_])) <= 1e-9 * max(1.0, abs(float(g[k_]))) + 1e-7 * float(x.std()), (k_, float(x.mean()), float(g[k_]))
        return dict(n_paths=n_paths, n_steps=n_steps, seed=42, option_type=typ, include_second_order=second, kwargs=kw, keys=list(g.keys()),
                    values={k_: float(x) for k_, x in g.items()},
                    std_errors={k_: float(per_path[k_].std() / np.sqrt(n_paths)) for k_ in g.keys()})

    doc["exotic_adapter_greeks"] = []
    for name, make, typ, second, kw in [
            ("asian", lambda: A(S=100.0, K=100.0, T=1.0, r=0.05, sigma=0.2, seed=42), "call", False, dict(avg_type="arithmetic")),
            ("asian", lambda: A(S=100.0, K=100.0, T=1.0, r=0.05, sigma=0.2, seed=42), "put", True, dict(avg_type="geometric")),
            ("barrier", lambda: ref["BarrierOption"](S=100.0, K=100.0, T=1.0, r=0.05, sigma=0.2, barrier=120.0, seed=42), "call", True, dict(barrier_type="up-and-out")),
            ("barrier", lambda: ref["BarrierOption"](S=100.0, K=100.0, T=1.0, r=0.05, sigma=0.2, barrier=85.0, seed=42), "put", False, dict(barrier_type="down-and-in")),
            ("lookback", lambda: ref["LookbackOption"](S=100.0, K=100.0, T=1.0, r=0.05, sigma=0.2, seed=42), "call", True, dict(lookback_type="floating")),
            ("lookback", lambda: ref["LookbackOption"](S=100.0, K=100.0, T=1.0, r=0.05, sigma=0.2, seed=42), "put", False, dict(lookback_type="fixed"))]:
        entry = adapter_greeks_with_errors(make, 20000, 64, typ, second, **kw)
        entry["option"] = name
        if name == "barrier":
            entry["barrier"] = float(make().barrier)
        doc["exotic_adapter_greeks"].append(entry)
    assert doc["exotic_adapter_greeks"][0]["values"] == doc["asian_greeks"]["values"]      # the case the fixture has held since round 1

    # -- barrier / lookback (exotic_options.py:163-224, 347-401) -------------------------
    doc["barrier"], doc["lookback"] = [], []
    B, L = ref["BarrierOption"], ref["LookbackOption"]
    for level, kind, typ, n, m, q in [(120.0, "up-and-out", "call", 100000, 252, 0.0), (120.0, "up-and-in", "call", 100000, 252, 0.0),
                                      (80.0, "down-and-out", "put", 100000, 252, 0.0), (80.0, "down-and-in", "put", 100000, 252, 0.0),
                                      (130.0, "up-and-out", "call", 50000, 252, 0.0), (90.0, "down-and-out", "call", 20000, 64, 0.02),
                                      (110.0, "up-and-in", "put", 20000, 64, 0.02), (100.0, "up-and-out", "call", 1000, 16, 0.0)]:
        o = B(S=100.0, K=100.0, T=1.0, r=0.05, sigma=0.2, q=q, barrier=level, seed=42)
        doc["barrier"].append(dict(params=[100.0, 100.0, 1.0, 0.05, 0.2, q], barrier=level, barrier_type=kind, option_type=typ,
                                   n_paths=n, n_steps=m, seed=42, price=float(o.price(n, m, kind, typ))))
    for kind, typ, n, m, q in [("floating", "call", 100000, 252, 0.0), ("floating", "put", 100000, 252, 0.0),
                               ("fixed", "call", 100000, 252, 0.0), ("fixed", "put", 100000, 252, 0.0), ("floating", "call", 20000, 50, 0.02)]:
        o = L(S=100.0, K=100.0, T=1.0, r=0.05, sigma=0.2, q=q, seed=42)
        doc["lookback"].append(dict(params=[100.0, 100.0, 1.0, 0.05, 0.2, q], lookback_type=kind, option_type=typ, n_paths=n,
                                    n_steps=m, seed=42, price=float(o.price(n, m, kind, typ))))
    doc["price_barrier_helper"] = dict(args=[100.0, 100.0, 1.0, 0.05, 0.2, 120.0, "up-and-out", "call", 20000, 42],
                                       value=float(ref["price_barrier"](100.0, 100.0, 1.0, 0.05, 0.2, 120.0, "up-and-out", "call", 20000, 42)))

    # -- autocallable / cliquet (exotic_options.py:404-554) -----------------------------------
    doc["autocallable"], doc["cliquet"] = [], []
    for kw, n, m, f in [(dict(), 100000, 252, 21), (dict(autocall_barrier=1.05, coupon_barrier=0.7, coupon_rate=0.08, ki_barrier=0.65), 50000, 252, 63),
                        (dict(q=0.02), 20000, 100, 30), (dict(ki_barrier=0.9), 20000, 50, 7)]:
        o = ref["AutocallableOption"](S=100.0, K=100.0, T=1.0, r=0.05, sigma=0.2, seed=42, **kw)
        doc["autocallable"].append(dict(kwargs=kw, n_paths=n, n_steps=m, observation_freq=f, seed=42, price=float(o.price(n, m, f))))
    for kw, n, m, p_ in [(dict(), 100000, 252, 12), (dict(local_cap=0.03, local_floor=-0.02, global_cap=0.2, global_floor=0.02), 50000, 252, 4),
                         (dict(q=0.01), 20000, 100, 7)]:
        o = ref["CliquetOption"](S=100.0, K=100.0, T=1.0, r=0.05, sigma=0.2, seed=42, **kw)
        doc["cliquet"].append(dict(kwargs=kw, n_paths=n, n_steps=m, n_periods=p_, seed=42, price=float(o.price(n, m, p_))))

    # -- American LSM (exotic_options.py:227-305) -----------------------------------------------
    doc["american"] = []
    for (S, K, T, r, v, q), typ, n, m, deg in [((100.0, 100.0, 1.0, 0.05, 0.2, 0.0), "put", 50000, 50, 3), ((100.0, 100.0, 1.0, 0.05, 0.2, 0.0), "call", 50000, 50, 3),
                                               ((90.0, 100.0, 0.5, 0.03, 0.3, 0.0), "put", 20000, 25, 2), ((100.0, 100.0, 1.0, 0.05, 0.2, 0.08), "call", 20000, 40, 3),
                                               ((100.0, 100.0, 1.0, 0.05, 0.2, 0.0), "put", 5000, 1, 3)]:
        o = ref["AmericanOption"](S=S, K=K, T=T, r=r, sigma=v, q=q, seed=42)
        doc["american"].append(dict(params=[S, K, T, r, v, q], option_type=typ, n_paths=n, n_steps=m, poly_degree=deg, seed=42,
                                    price=float(o.price(n, m, typ, deg))))

    # -- Heston (heston.py:131-255) ---------------------------------------------------------
    import warnings
    doc["heston"] = []
    for (kappa, theta, sv, rho, v0), (S, K, T, r, q), typ, n, m in [
            ((2.0, 0.04, 0.3, -0.7, 0.04), (100.0, 100.0, 1.0, 0.05, 0.0), "call", 100000, 252),
            ((2.0, 0.04, 0.3, -0.7, 0.04), (100.0, 100.0, 1.0, 0.05, 0.0), "put", 100000, 252),
            ((1.5, 0.06, 0.5, -0.5, 0.03), (100.0, 110.0, 0.5, 0.03, 0.01), "call", 50000, 100),
            ((3.0, 0.02, 0.8, 0.3, 0.05), (100.0, 90.0, 2.0, 0.02, 0.0), "put", 50000, 101)]:      # Feller violated: v hits 0
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            hp = ref["HestonPricer"](kappa=kappa, theta=theta, sigma_v=sv, rho=rho, v0=v0)
        doc["heston"].append(dict(model=[kappa, theta, sv, rho, v0], args=[S, K, T, r, q], option_type=typ, n_paths=n, n_steps=m, seed=42,
                                  mc=float(hp.price_monte_carlo(S, K, T, r, q, typ, n, m, 42)),
                                  semi_analytic=float(hp.price_european(S, K, T, r, q, typ))))

    # -- path generators and the exercise boundary (heston.py:257-305, jump_diffusion.py:227-272, exotic_options.py:309-345)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        hp = ref["HestonPricer"](kappa=2.0, theta=0.04, sigma_v=0.3, rho=-0.7, v0=0.04)
    sp, vp = hp.simulate_paths(100.0, 1.0, 0.05, 0.01, 2000, 12, 42)
    doc["heston_paths"] = dict(model=[2.0, 0.04, 0.3, -0.7, 0.04], args=[100.0, 1.0, 0.05, 0.01, 2000, 12, 42], shape=list(sp.shape),
                               spot_row0=[float(x) for x in sp[0]], var_row0=[float(x) for x in vp[0]],
                               spot_col_mean=[float(x) for x in sp.mean(axis=0)], var_col_mean=[float(x) for x in vp.mean(axis=0)])
    jd = ref["MertonJumpDiffusion"](lambda_j=3.0, mu_j=-0.1, sigma_j=0.2)
    doc["merton_path"] = dict(model=[3.0, -0.1, 0.2], args=[100.0, 1.0, 0.05, 0.2, 0.01, 24, 42],
                              path=[float(x) for x in jd.simulate_path(100.0, 1.0, 0.05, 0.2, 0.01, 24, 42)])
    doc["exercise_boundary"] = []
    for (S, K, T, r, v, q), typ, n, m in [((100.0, 100.0, 1.0, 0.05, 0.2, 0.0), "put", 10000, 50), ((100.0, 100.0, 1.0, 0.05, 0.2, 0.03), "call", 10000, 50),
                                          ((120.0, 100.0, 0.5, 0.03, 0.3, 0.0), "put", 4000, 10)]:
        o = ref["AmericanOption"](S=S, K=K, T=T, r=r, sigma=v, q=q, seed=42)
        times, b = o.early_exercise_boundary(n, m, typ)
        doc["exercise_boundary"].append(dict(params=[S, K, T, r, v, q], option_type=typ, n_paths=n, n_steps=m, seed=42,
                                             times=[float(x) for x in times], boundary=[None if np.isnan(x) else float(x) for x in b]))

    # -- full paths (gbm_numpy.py:86-118) ------------------------------------------------------
    fp = ref["simulate_gbm_paths"](100.0, 1.0, 0.05, 0.2, 0.01, 1000, 12, 42)
    doc["full_paths"] = dict(args=[100.0, 1.0, 0.05, 0.2, 0.01, 1000, 12, 42], shape=list(fp.shape), c_contiguous=bool(fp.flags["C_CONTIGUOUS"]),
                             row0=[float(x) for x in fp[0]], row999_tail=[float(x) for x in fp[999, -3:]],
                             col_mean=[float(x) for x in fp.mean(axis=0)])

    # -- antithetic Sobol backend (gbm_qmc.py:49-76; exported, not wired to the pricer) ------------
    import warnings as _w
    with _w.catch_warnings():
        _w.simplefilter("ignore")
        qa = ref["simulate_gbm_qmc_antithetic"](100.0, 1.0, 0.05, 0.2, 0.01, 1024, 12, 42)
    doc["qmc_antithetic"] = dict(args=[100.0, 1.0, 0.05, 0.2, 0.01, 1024, 12, 42], length=int(len(qa)), head=[float(x) for x in qa[:4]],
                                 mid=[float(x) for x in qa[1024:1028]], mean=float(qa.mean()))

    # -- jump diffusion (jump_diffusion.py) -----------------------------------------------------
    doc["merton"], doc["kou"] = [], []
    for (lam, mu, sj), (S, K, T, r, v, q), typ, n, m in [((0.5, -0.1, 0.2), (100.0, 100.0, 1.0, 0.05, 0.2, 0.0), "call", 40000, 50),
                                                         ((2.0, 0.05, 0.1), (100.0, 105.0, 0.5, 0.03, 0.15, 0.01), "put", 40000, 25),
                                                         ((0.0, -0.1, 0.2), (100.0, 100.0, 1.0, 0.05, 0.2, 0.0), "call", 20000, 10)]:
        jd = ref["MertonJumpDiffusion"](lambda_j=lam, mu_j=mu, sigma_j=sj)
        doc["merton"].append(dict(model=[lam, mu, sj], args=[S, K, T, r, v, q], option_type=typ, n_paths=n, n_steps=m, seed=42,
                                  kappa=float(jd.kappa), series=float(jd.price(S, K, T, r, v, typ, q)),
                                  mc=float(jd.price_monte_carlo(S, K, T, r, v, typ, q, n, m, 42))))
    for (lam, p_, e1, e2), (S, K, T, r, v, q), typ, n, m in [((1.0, 0.4, 10.0, 5.0), (100.0, 100.0, 1.0, 0.05, 0.2, 0.0), "call", 40000, 50),
                                                            ((3.0, 0.6, 25.0, 20.0), (100.0, 95.0, 0.5, 0.03, 0.15, 0.01), "put", 40000, 25)]:
        kj = ref["KouJumpDiffusion"](lambda_j=lam, p=p_, eta1=e1, eta2=e2)
        doc["kou"].append(dict(model=[lam, p_, e1, e2], args=[S, K, T, r, v, q], option_type=typ, n_paths=n, n_steps=m, seed=42,
                               kappa=float(kj.kappa), mc=float(kj.price_monte_carlo(S, K, T, r, v, typ, q, n, m, 42))))

    # -- MonteCarloPricerUni, NumPy backend (monte_carlo_unified.py:298-343, 451-689) ----
    Uni = ref["MonteCarloPricerUni"]
    batch = dict(S=[100.0, 110.0, 90.0, 100.0, 100.0], K=[100.0, 100.0, 100.0, 95.0, 105.0], T=[1.0, 1.0, 1.0, 0.5, 0.5],
                 r=[0.05] * 5, sigma=[0.2, 0.2, 0.2, 0.3, 0.15], q=[0.0, 0.0, 0.0, 0.02, 0.01])   # tests/test_monte_carlo.py:75-87
    u = Uni(num_simulations=10000, num_steps=50, seed=42, use_numba=False, use_gpu=False)
    arr = {k: np.array(v) for k, v in batch.items()}
    doc["uni"] = dict(
        ctor=[10000, 50, 42], batch=batch,
        price_call=u.price(100, 100, 1.0, 0.05, 0.2, "call"), price_put=u.price(100, 100, 1.0, 0.05, 0.2, "put"),
        price_seed7=u.price(100, 100, 1.0, 0.05, 0.2, "call", q=0.01, seed=7),
        delta_gamma_seed5=list(u.delta_gamma(100, 100, 1.0, 0.05, 0.2, "call", seed=5)),
        delta_gamma_h1_seed5=list(u.delta_gamma(100, 100, 1.0, 0.05, 0.2, "put", q=0.01, h=1.0, seed=5)),
        price_batch_call=[float(x) for x in u.price_batch(arr["S"], arr["K"], arr["T"], arr["r"], arr["sigma"], "call", arr["q"])],
        price_batch_put_scalar_q=[float(x) for x in u.price_batch(arr["S"], arr["K"], arr["T"], arr["r"], arr["sigma"], "put", 0.01)],
        delta_gamma_batch_h1=[[float(x) for x in a] for a in u.delta_gamma_batch(arr["S"], arr["K"], arr["T"], arr["r"], arr["sigma"], "call", arr["q"], h=1.0)],
        delta_gamma_batch=[[float(x) for x in a] for a in u.delta_gamma_batch(arr["S"], arr["K"], arr["T"], arr["r"], arr["sigma"], "call", arr["q"])],
    )
    # unseeded delta_gamma draws its seed from pricer.rng (:549-550): record the first draw
    u2 = Uni(num_simulations=2000, num_steps=10, seed=11, use_numba=False, use_gpu=False)
    doc["uni"]["delta_gamma_unseeded_first"] = list(u2.delta_gamma(100, 100, 1.0, 0.05, 0.2, "call", h=1.0))

    with open(OUT, "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
