"""The reference's own test cases for the hot path, run against this engine THROUGH the reference's
import paths (optionslab_amd.compat.install()): what a user who switches sees.

Each case below names the reference test it restates (tests/test_monte_carlo.py,
tests/test_exotic_options.py, tests/test_parity.py of OptionsLab) and keeps its inputs and its
acceptance bound.  Cases the reference itself skips (Greeks methods removed from the pricer, input
validation "changed in new API", numba absent) are listed at the bottom with what this engine does.
"""
import math
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ATM = dict(S=100, K=100, T=1.0, r=0.05, sigma=0.2)


@pytest.fixture(scope="module")
def ref():
    """The reference's modules, resolved to this engine."""
    from optionslab_amd import compat

    compat.install()
    try:
        import importlib
        import types

        # (importlib, not `import a.b.c as x`: as in the reference, the package re-exports the FUNCTION
        # black_scholes under the name of its own submodule, src/pricing_models/__init__.py:24)
        exc, exo, mc, uni = (importlib.import_module(m) for m in (
            "src.exceptions.montecarlo_exceptions", "src.pricing_models.exotic_options", "src.pricing_models.monte_carlo",
            "src.pricing_models.monte_carlo_unified"))
        from src.pricing_models.black_scholes import black_scholes

        assert getattr(mc, "__optionslab_amd__", False) and getattr(exo, "__optionslab_amd__", False)
        yield dict(mc=mc, uni=uni, bs=types.SimpleNamespace(black_scholes=black_scholes), exo=exo, exc=exc)
    finally:
        compat.uninstall()


@pytest.fixture
def basic_pricer(ref):          # tests/test_monte_carlo.py:37-44
    return ref["mc"].MonteCarloPricer(num_simulations=10000, num_steps=50, seed=42)


@pytest.fixture
def unified_pricer(ref):        # tests/test_monte_carlo.py:60-69
    return ref["uni"].MonteCarloPricerUni(num_simulations=10000, num_steps=50, seed=42, use_numba=False, use_gpu=False)


@pytest.fixture
def book():                     # tests/test_monte_carlo.py:72-85 (sample_options_df, as plain arrays)
    return dict(S_vals=np.array([100.0, 110.0, 90.0, 100.0, 100.0]), K_vals=np.array([100.0, 100.0, 100.0, 95.0, 105.0]),
                T_vals=np.array([1.0, 1.0, 1.0, 0.5, 0.5]), r_vals=np.full(5, 0.05),
                sigma_vals=np.array([0.2, 0.2, 0.2, 0.3, 0.15]), q_vals=np.array([0.0, 0.0, 0.0, 0.02, 0.01]))


def bs_call(S, K, T, r, sigma, q=0.0):      # tests/test_exotic_options.py:36-40
    d1 = (math.log(S / K) + (r - q + 0.5 * sigma * sigma) * T) / (sigma * math.sqrt(T))
    d2 = d1 - sigma * math.sqrt(T)
    cdf = lambda x: 0.5 * math.erfc(-x / math.sqrt(2.0))
    return S * math.exp(-q * T) * cdf(d1) - K * math.exp(-r * T) * cdf(d2)


def bs_put(S, K, T, r, sigma, q=0.0):       # tests/test_exotic_options.py:43-47
    return bs_call(S, K, T, r, sigma, q) - S * math.exp(-q * T) + K * math.exp(-r * T)


# ------------------------------------------------------------------ TestMonteCarloPricer
def test_initialization(ref):                                   # :95-104
    p = ref["mc"].MonteCarloPricer(num_simulations=5000, num_steps=100, seed=123)
    assert (p.num_simulations, p.num_steps, p.seed) == (5000, 100, 123)


def test_initialization_invalid_simulations(ref):               # :106-112
    for n in (0, -100):
        with pytest.raises(ValueError):
            ref["mc"].MonteCarloPricer(num_simulations=n)


@pytest.mark.parametrize("kind", ["call", "put"])
def test_price_call_and_put_option(ref, basic_pricer, kind):    # :119-141
    price = basic_pricer.price(option_type=kind, q=0.0, **ATM)
    assert isinstance(price, float) and 0 < price < 100
    assert abs(price - ref["bs"].black_scholes(100, 100, 1.0, 0.05, 0.2, kind, 0.0)) < 1.0


def test_price_reproducibility(basic_pricer):                   # :153-158
    assert basic_pricer.price(100, 100, 1.0, 0.05, 0.2, "call", seed=42) == basic_pricer.price(100, 100, 1.0, 0.05, 0.2, "call", seed=42)


def test_price_with_std_error(ref, basic_pricer):               # :160-168
    res = basic_pricer.price(100, 100, 1.0, 0.05, 0.2, "call", return_error=True)
    assert isinstance(res, ref["mc"].MCResult) and res.price > 0 and 0 < res.std_error < res.price


def test_backends_agree(ref):                                   # :204-225 (numba vs numpy there; every backend here)
    M = ref["mc"].MCMethod
    prices = {m: ref["mc"].MonteCarloPricer(50000, 100, 42, m).price(100, 100, 1.0, 0.05, 0.2, "call")
              for m in (M.NUMPY, M.NUMBA, M.HIP)}
    base = prices[M.NUMPY]
    assert all(abs(p - base) / base < 0.05 for p in prices.values())


# ------------------------------------------------------------------ TestMonteCarloPricerUni
def test_uni_initialization(ref):                               # :373-382
    p = ref["uni"].MonteCarloPricerUni(num_simulations=10000, num_steps=100, seed=42)
    assert (p.num_simulations, p.num_steps) == (10000, 100)


def test_uni_initialization_invalid(ref):                       # :384-390
    bad = (ref["exc"].InputValidationError, ref["uni"].InputValidationError)
    with pytest.raises(bad):
        ref["uni"].MonteCarloPricerUni(num_simulations=0)
    with pytest.raises(bad):
        ref["uni"].MonteCarloPricerUni(num_steps=-1)


@pytest.mark.parametrize("kind", ["call", "put"])
def test_uni_price(ref, unified_pricer, kind):                  # :392-412
    price = unified_pricer.price(option_type=kind, **ATM)
    assert price > 0 and abs(price - ref["bs"].black_scholes(100, 100, 1.0, 0.05, 0.2, kind)) < 1.5


def test_uni_price_invalid_inputs(ref, unified_pricer):         # :414-423
    bad = (ref["exc"].InputValidationError, ref["uni"].InputValidationError)
    for args in ((0, 100, 1.0, 0.05, 0.2, "call"), (100, 100, 1.0, 0.05, -0.1, "call"), (100, 100, 1.0, 0.05, 0.2, "invalid")):
        with pytest.raises(bad):
            unified_pricer.price(*args)


def test_uni_delta_gamma(unified_pricer):                       # :425-432
    delta, gamma = unified_pricer.delta_gamma(option_type="call", **ATM)
    # The reference also asks gamma > 0 here.  At its default h = 1e-4 the second difference of 20,000 CRN
    # payoffs is rounding noise / 1e-8 unless a terminal price lands within 1e-6 of the strike: a coin flip
    # in the reference too (its own value at seed 5 is -6.2e-06, tests/golden "delta_gamma_seed5").  The sign
    # is asserted where the estimator resolves it.
    assert 0 < delta < 1 and math.isfinite(gamma) and abs(gamma) < 1e-3
    delta1, gamma1 = unified_pricer.delta_gamma(option_type="call", h=1.0, **ATM)
    assert 0 < delta1 < 1 and gamma1 > 0


def test_uni_price_batch(unified_pricer, book):                 # :434-447
    prices = unified_pricer.price_batch(option_type="call", **book)
    assert len(prices) == 5 and all(p > 0 for p in prices)


def test_uni_delta_gamma_batch(unified_pricer, book):           # :449-462
    deltas, gammas = unified_pricer.delta_gamma_batch(option_type="call", **book)
    assert len(deltas) == 5 and len(gammas) == 5
    assert all(0 < d < 1 for d in deltas) and np.isfinite(gammas).all()      # stronger than the reference asks
    deltas, gammas = unified_pricer.delta_gamma_batch(option_type="call", h=1.0, **book)
    assert all(0 < d < 1 for d in deltas) and all(g > 0 for g in gammas)


# ------------------------------------------------------------------ TestIntegration
def test_put_call_parity(basic_pricer):                         # :509-521
    S, K, T, r, sigma, q = 100, 100, 1.0, 0.05, 0.2, 0.02
    gap = basic_pricer.price(S, K, T, r, sigma, "call", q) - basic_pricer.price(S, K, T, r, sigma, "put", q)
    assert abs(gap - (S * math.exp(-q * T) - K * math.exp(-r * T))) < 2.0


def test_itm_vs_otm(basic_pricer):                              # :523-532
    itm, atm, otm = (basic_pricer.price(s, 100, 1.0, 0.05, 0.2, "call") for s in (110, 100, 90))
    assert itm > atm > otm


def test_higher_vol_higher_price(basic_pricer):                 # :534-539
    assert basic_pricer.price(100, 100, 1.0, 0.05, 0.4, "call") > basic_pricer.price(100, 100, 1.0, 0.05, 0.1, "call")


def test_longer_maturity_higher_price(basic_pricer):            # :541-546
    assert basic_pricer.price(100, 100, 2.0, 0.05, 0.2, "call") > basic_pricer.price(100, 100, 0.25, 0.05, 0.2, "call")


def test_basic_pricer_throughput(basic_pricer):                 # :558-569 (marked slow there: 100 calls in < 30 s)
    basic_pricer.price(100, 100, 1.0, 0.05, 0.2, "call")
    t0 = time.perf_counter()
    for _ in range(100):
        basic_pricer.price(100, 100, 1.0, 0.05, 0.2, "call")
    assert time.perf_counter() - t0 < 30


# ------------------------------------------------------------------ tests/test_parity.py:15 (Black-Scholes identity)
@pytest.mark.parametrize("S,K,T,r,sigma", [(100, 100, 1.0, 0.05, 0.2), (120, 100, 0.5, 0.01, 0.35), (80, 100, 2.0, 0.03, 0.15)])
def test_black_scholes_put_call_parity(ref, S, K, T, r, sigma):
    bs = ref["bs"].black_scholes
    assert bs(S, K, T, r, sigma, "call") - bs(S, K, T, r, sigma, "put") == pytest.approx(S - K * math.exp(-r * T), abs=1e-9)


# ------------------------------------------------------------------ TestAsianOptions
@pytest.mark.parametrize("kind", ["call", "put"])
def test_asian_positive_price(ref, kind):                       # test_exotic_options.py:56-72
    assert ref["exo"].AsianOption(seed=42, **ATM).price(n_paths=10000, avg_type="arithmetic", option_type=kind) > 0


def test_asian_cheaper_than_european(ref):                      # :74-86
    assert ref["exo"].AsianOption(seed=42, **ATM).price(n_paths=50000, avg_type="arithmetic", option_type="call") < bs_call(**ATM)


def test_geometric_closed_form_matches_mc(ref):                 # :88-100
    a = ref["exo"].AsianOption(seed=42, **ATM)
    mc = a.price(n_paths=100000, n_steps=252, avg_type="geometric", option_type="call")
    cf = a.price_geometric_closed_form(option_type="call")
    assert abs(mc - cf) / cf < 0.05


def test_asian_reproducibility(ref):                            # :102-114
    mk = lambda: ref["exo"].AsianOption(seed=42, **ATM)
    assert mk().price(n_paths=1000) == mk().price(n_paths=1000)


# ------------------------------------------------------------------ TestBarrierOptions
@pytest.mark.parametrize("level,kind,typ", [(120, "up-and-out", "call"), (80, "down-and-out", "put")])
def test_knock_out_non_negative(ref, level, kind, typ):         # :123-143
    assert ref["exo"].BarrierOption(barrier=level, seed=42, **ATM).price(n_paths=10000, barrier_type=kind, option_type=typ) >= 0


def test_knock_out_cheaper_than_european(ref):                  # :145-160
    p = ref["exo"].BarrierOption(barrier=130, seed=42, **ATM).price(n_paths=50000, barrier_type="up-and-out", option_type="call")
    assert p < bs_call(**ATM)


def test_knock_in_plus_knock_out_equals_european(ref):          # :162-185
    mk = lambda: ref["exo"].BarrierOption(barrier=120, seed=42, **ATM)
    both = (mk().price(n_paths=100000, barrier_type="up-and-out", option_type="call")
            + mk().price(n_paths=100000, barrier_type="up-and-in", option_type="call"))
    assert abs(both - bs_call(**ATM)) / bs_call(**ATM) < 0.1


def test_invalid_barrier_raises(ref):                           # :187-193
    with pytest.raises(ValueError, match="positive"):
        ref["exo"].BarrierOption(barrier=0, **ATM).price(barrier_type="up-and-out")


# ------------------------------------------------------------------ TestAmericanOptions
def test_american_put_positive(ref):                            # :202-209
    assert ref["exo"].AmericanOption(seed=42, **ATM).price(n_paths=10000, option_type="put") > 0


def test_american_put_greater_than_european(ref):               # :211-223
    assert ref["exo"].AmericanOption(seed=42, **ATM).price(n_paths=50000, option_type="put") >= 0.95 * bs_put(**ATM)


def test_american_call_equals_european_no_dividend(ref):        # :225-237
    p = ref["exo"].AmericanOption(q=0.0, seed=42, **ATM).price(n_paths=50000, option_type="call")
    assert abs(p - bs_call(**ATM)) / bs_call(**ATM) < 0.1


def test_itm_put_early_exercise_premium(ref):                   # :239-252
    deep = dict(ATM, S=70)
    assert ref["exo"].AmericanOption(seed=42, **deep).price(n_paths=50000, option_type="put") > bs_put(**deep)


# ------------------------------------------------------------------ TestConvenienceFunctions
def test_price_asian(ref):                                      # :260-272
    assert ref["exo"].price_asian(avg_type="arithmetic", n_paths=10000, seed=42, **ATM) > 0


def test_price_barrier(ref):                                    # :274-287
    assert ref["exo"].price_barrier(barrier=120, barrier_type="up-and-out", n_paths=10000, seed=42, **ATM) >= 0


def test_price_american(ref):                                   # :289-301
    assert ref["exo"].price_american(option_type="put", n_paths=10000, seed=42, **ATM) > 0


# ------------------------------------------------------------------ what the reference skips
def test_cases_the_reference_skips(ref, basic_pricer):
    """:114-117 (num_steps unvalidated), :143-151 (option type / spot unvalidated), :170-202 (Greeks
    methods gone from the pricer).  Same behaviour here: no exception, anything but "call" is a put,
    a non-positive spot propagates NaN like NumPy; the Greeks live in compute_greeks_unified."""
    ref["mc"].MonteCarloPricer(num_simulations=10, num_steps=0)
    assert basic_pricer.price(100, 100, 1.0, 0.05, 0.2, "invalid") == basic_pricer.price(100, 100, 1.0, 0.05, 0.2, "put")
    assert math.isnan(basic_pricer.price(-1.0, 100, 1.0, 0.05, 0.2, "call"))
    assert not any(hasattr(basic_pricer, name) for name in ("delta", "gamma", "vega", "theta", "rho", "all_greeks"))
