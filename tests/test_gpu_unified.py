"""GPU parity of MonteCarloPricerUni (batch-of-options pricer) against golden
vectors captured from the reference's NumPy backend and against its own test assertions
(tests/test_monte_carlo.py:370-462 of the reference)."""
import math

import numpy as np
import pytest

import optionslab_amd as ol
from optionslab_amd import _hip
from optionslab_amd.monte_carlo_unified import InputValidationError, MonteCarloError, MonteCarloPricerUni
from oracle import numpy_reference as orc
from oracle import philox_oracle as po

pytestmark = pytest.mark.gpu
ATM = (100.0, 100.0, 1.0, 0.05, 0.2)


@pytest.fixture(scope="module")
def uni(golden):
    N, M, seed = golden["uni"]["ctor"]
    return MonteCarloPricerUni(num_simulations=N, num_steps=M, seed=seed, use_numba=False, use_gpu=False)


def _se(S, K, T, r, v, q, call, N, M, seed):
    return _hip.european(S, K, T, r, v, q, call, N, M, seed).std_error


def test_constructor_contract():
    p = MonteCarloPricerUni(num_simulations=10000, num_steps=100, seed=42)
    assert (p.num_simulations, p.num_steps, p.seed) == (10000, 100, 42)
    for kw in (dict(num_simulations=0), dict(num_steps=-1)):
        with pytest.raises(InputValidationError):
            MonteCarloPricerUni(**kw)
    assert ol.MonteCarloPricerUni is MonteCarloPricerUni


def test_price_against_reference_and_black_scholes(uni, golden):
    g = golden["uni"]
    for typ, key in (("call", "price_call"), ("put", "price_put")):
        price = uni.price(*ATM, typ)
        se = _se(*ATM, 0.0, typ == "call", uni.num_simulations, uni.num_steps, uni.seed)
        assert type(price) is float and price > 0
        assert abs(price - g[key]) <= 3 * math.sqrt(2) * se                     # both estimates carry ~se
        assert abs(price - ol.black_scholes(*ATM, typ)) < 1.5                    # reference test :392-412
    p7 = uni.price(*ATM, "call", q=0.01, seed=7)
    assert abs(p7 - g["price_seed7"]) <= 3 * math.sqrt(2) * _se(*ATM, 0.01, True, uni.num_simulations, uni.num_steps, 7)
    assert uni.price(*ATM, "call", seed=7) == uni.price(*ATM, "call", seed=7)
    # same engine as MonteCarloPricer at equal (N, M, seed)
    assert uni.price(*ATM, "call") == ol.MonteCarloPricer(uni.num_simulations, uni.num_steps, uni.seed).price(*ATM, "call")


def test_validation_and_error_wrapping(uni):
    for args in ((0, 100, 1.0, 0.05, 0.2, "call"), (100, 100, 1.0, 0.05, -0.1, "call"), (100, 100, 1.0, 0.05, 0.2, "invalid"),
                 (100, 0, 1.0, 0.05, 0.2, "put"), (100, 100, 0.0, 0.05, 0.2, "put")):
        with pytest.raises(InputValidationError):
            uni.price(*args)
    with pytest.raises(InputValidationError):
        uni.delta_gamma(100, 100, 1.0, 0.05, 0.2, "straddle")
    big = MonteCarloPricerUni(10, 10, 1)
    big.num_steps = 0                     # force a library argument error -> wrapped like the reference (:510-511)
    with pytest.raises(MonteCarloError, match="Monte Carlo pricing failed"):
        big.price(*ATM, "call")


def test_delta_gamma_crn(uni, golden):
    g = golden["uni"]
    exact = orc.bs_greeks(*ATM, "call")
    d, gm = uni.delta_gamma(*ATM, "call", h=1.0, seed=5)
    assert d == pytest.approx(exact["delta"], abs=0.02) and gm == pytest.approx(exact["gamma"], abs=0.004)
    dp, gp = uni.delta_gamma(*ATM, "put", q=0.01, h=1.0, seed=5)
    assert dp == pytest.approx(g["delta_gamma_h1_seed5"][0], abs=0.02) and gp == pytest.approx(g["delta_gamma_h1_seed5"][1], abs=0.004)
    # fused launch == three separate price() calls with the same seed (the reference's literal form)
    up, mid, dn = (uni.price(100.0 + s, 100.0, 1.0, 0.05, 0.2, "call", seed=5) for s in (1.0, 0.0, -1.0))
    assert d == pytest.approx((up - dn) / 2.0, rel=1e-9) and gm == pytest.approx(up - 2 * mid + dn, rel=1e-6, abs=1e-9)
    # default h = 1e-4 (:522): delta is sharp under CRN; gamma divides rounding noise by 1e-8 exactly as the reference
    d4, g4 = uni.delta_gamma(*ATM, "call", seed=5)
    assert d4 == pytest.approx(g["delta_gamma_seed5"][0], abs=0.02) and 0 < d4 < 1 and math.isfinite(g4)   # :425-432
    # unseeded: the seed is drawn from pricer.rng, first draw identical to the reference's (:549-550)
    u2 = MonteCarloPricerUni(2000, 10, 11)
    want_seed = int(np.random.default_rng(11).integers(0, 2**31))
    assert u2.delta_gamma(*ATM, "call", h=1.0) == MonteCarloPricerUni(2000, 10, 11).delta_gamma(*ATM, "call", h=1.0, seed=want_seed)


def test_gamma_at_the_default_step_is_rounding_noise_no_larger_than_fp64_libm_makes_it():
    """delta_gamma's default h = 1e-4 (monte_carlo_unified.py:522, 558): with a few thousand paths no path prices within h of
    the strike, every payoff is LINEAR in S over [S - h, S + h], and the exact second difference of the three prices is 0 -- what
    the reference returns as gamma there is rounding noise divided by h^2 = 1e-8, and so does the device.  This test makes "noise
    exactly as the reference's" a bound instead of a comment: on the same Philox stream the device's gamma and the fp64 / libm
    checker's gamma must both sit inside the envelope that fp64 rounding of ONE price allows, worst case coherent over the paths:
        relative error of a terminal price   <= ulp(a) (a = ln S + drift, the exponent: an absolute error there is a relative one in S_T)
                                                + 3 ulp (the exponential: libm <= 1, the device's <= 3) + 2 ulp (scale = exp(a - a_base), the product)
        error of a discounted price          <= e^{-rT} E[S_T ; in the money] x that  + 8 ulp(price) for the sums
        error of gamma                       <= 4 x that / h^2                           (|1| + |-2| + |1| coefficients)
    about 4e-5 -- 0.2 % of the true gamma, 0.0188: the default step measures nothing but this, on either implementation."""
    S, K, T, r, v = ATM
    N, M, h = 2_000, 10, 1e-4
    for seed in range(5, 25):           # the first seed none of whose 4,000 terminal prices lies within 4 windows of the strike (P ~ 0.94 each)
        st = _hip.european_terminal(S, T, r, v, 0.0, N, M, seed, True)
        if not np.any(np.abs(st - K) <= 4 * h * st / S):
            break
    else:
        pytest.fail("twenty seeds in a row with a path in the kink window")
    d, g = MonteCarloPricerUni(N, M, 1).delta_gamma(S, K, T, r, v, "call", h=h, seed=seed)
    chk = []
    for s_ in (S + h, S, S - h):
        sx, sxx, *_m, n = po.european_moments(s_, K, T, r, v, 0.0, True, N, M, seed, True)
        chk.append(po.price_and_error(sx, sxx, n, r, T)[0])
    up, mid, dn = chk
    g_chk, d_chk = (up - 2 * mid + dn) / h**2, (up - dn) / (2 * h)
    eps = 2.0 ** -53
    a = math.log(S) + (r - 0.5 * v * v) * T
    rel_st = float(np.spacing(a)) + (3 + 2) * 2 * eps
    itm_mass = math.exp(-r * T) * float(np.mean(np.where(st > K, st, 0.0)))          # e^{-rT} E[S_T ; S_T > K]
    bound = 4 * (itm_mass * rel_st + 8 * float(np.spacing(mid))) / h**2
    assert 1e-5 < bound < 1e-4
    assert abs(g_chk) <= bound, (g_chk, bound)                                        # the yardstick: fp64 + libm on this stream
    assert abs(g) <= bound, (g, bound)                                                # the device is inside the same envelope
    assert abs(g - g_chk) <= bound
    assert d == pytest.approx(d_chk, rel=2e-6) and d == pytest.approx(orc.bs_greeks(*ATM, "call")["delta"], abs=0.02)   # delta stays sharp under CRN


def test_a_batch_of_70000_contracts_completes_by_the_polled_word_as_by_the_stream():
    """olmc_european_multi beyond 65,535 contracts = two launches of one call; every contract's finisher writes its sums to pinned
    host memory and takes an acquire-release ticket, the last one raises the completion word.  Polling on must return exactly what
    waiting for the stream returns, for every one of the 70,000 contracts (a flag raised before some contract's sums had landed
    would show as a stale row)."""
    n = 70_000
    rng = np.random.default_rng(3)
    S, K = rng.uniform(80, 120, n), rng.uniform(80, 120, n)
    T, r, v, q = rng.uniform(0.25, 2.0, n), rng.uniform(0.0, 0.08, n), rng.uniform(0.1, 0.5, n), rng.uniform(0.0, 0.03, n)
    try:
        _hip.tune(_hip.TUNE_POLL, -1)
        by_stream = _hip.european_multi(S, K, T, r, v, q, True, 256, 8, 42).copy()
        _hip.tune(_hip.TUNE_POLL, 0)
        for rep in range(3):                                  # the pinned rows are reused: a stale row of the previous call must not pass
            by_flag = _hip.european_multi(S, K + rep, T, r, v, q, True, 256, 8, 42)
            if rep == 0:
                assert np.array_equal(by_flag["sum"], by_stream["sum"]) and np.array_equal(by_flag["sumsq"], by_stream["sumsq"])
                assert np.array_equal(by_flag["price"], by_stream["price"]) and np.all(by_flag["n"] == 512)
            else:
                assert not np.array_equal(by_flag["sum"], by_stream["sum"])
        _hip.tune(_hip.TUNE_POLL, -1)
        assert np.array_equal(_hip.european_multi(S, K + 2, T, r, v, q, True, 256, 8, 42)["sum"], by_flag["sum"])
    finally:
        _hip.tune(_hip.TUNE_POLL, 0)
    bs = np.array([ol.black_scholes(S[j], K[j], T[j], r[j], v[j], "call", q[j]) for j in range(0, n, 700)])
    z = (by_stream["price"][::700] - bs) / by_stream["std_error"][::700]
    assert abs(z.mean()) < 0.5 and np.abs(z).max() < 5.0


def test_batch_workspace_grows_in_paths_without_doubling_in_contracts():
    """A convergence sweep of price_batch over increasing n_paths on ONE batch: the workspace must widen its rows and
    keep its contract capacity -- round 2 doubled the capacity on every step and ran into gigabytes after ~20 steps."""
    rng = np.random.default_rng(4)
    n = 100
    S, K = rng.uniform(80, 120, n), rng.uniform(80, 120, n)
    args = (S, K, np.full(n, 1.0), np.full(n, 0.05), np.full(n, 0.2), np.zeros(n), True)
    _hip.european_multi(*args, 256, 4, 1)
    cap0, _ = _hip.multi_capacity()
    assert cap0 >= n
    last = None
    for k in range(1, 33):
        last = _hip.european_multi(*args, 256 * 40 * k, 4, 1)                 # 40, 80, ... workgroups per contract: thirty-two growth steps
        cap, bpo = _hip.multi_capacity()
        assert cap == cap0 and min(40 * k, 1024) <= bpo <= 1024, (k, cap, bpo)      # a contract spreads over at most 1024 workgroups
    bs = np.array([ol.black_scholes(S[j], K[j], 1.0, 0.05, 0.2, "call", 0.0) for j in range(n)])
    assert np.all(np.abs(last["price"] - bs) <= 4.5 * last["std_error"])
    _hip.european_multi(*[np.tile(a, 3) if isinstance(a, np.ndarray) else a for a in args], 256, 4, 1)     # more contracts: now the capacity may grow
    assert _hip.multi_capacity()[0] >= 3 * n


def test_price_batch(uni, golden):
    g = golden["uni"]
    b = {k: np.array(v) for k, v in g["batch"].items()}
    prices = uni.price_batch(S_vals=b["S"], K_vals=b["K"], T_vals=b["T"], r_vals=b["r"], sigma_vals=b["sigma"],
                             option_type="call", q_vals=b["q"])
    assert isinstance(prices, np.ndarray) and prices.dtype == np.float64 and prices.shape == (5,)
    assert all(p > 0 for p in prices)                                           # reference test :434-447
    stats = _hip.european_multi(b["S"], b["K"], b["T"], b["r"], b["sigma"], b["q"], True, uni.num_simulations, uni.num_steps, uni.seed)
    for j in range(5):
        assert abs(prices[j] - g["price_batch_call"][j]) <= 3 * math.sqrt(2) * stats["std_error"][j], j
        bs = ol.black_scholes(b["S"][j], b["K"][j], b["T"][j], b["r"][j], b["sigma"][j], "call", b["q"][j])
        assert abs(prices[j] - bs) <= 3.5 * stats["std_error"][j], j
        assert stats["n"][j] == 2 * uni.num_simulations
    puts = uni.price_batch(b["S"], b["K"], b["T"], b["r"], b["sigma"], "put", 0.01)          # scalar q broadcast (:608-611)
    for j in range(5):
        assert abs(puts[j] - g["price_batch_put_scalar_q"][j]) <= 0.25
    assert np.array_equal(prices, uni.price_batch(b["S"], b["K"], b["T"], b["r"], b["sigma"], "call", b["q"]))   # CRN across calls
    # contract 0 uses stream tag 0 = the single-contract stream
    assert prices[0] == pytest.approx(uni.price(b["S"][0], b["K"][0], b["T"][0], b["r"][0], b["sigma"][0], "call", b["q"][0]), rel=1e-12)
    # contracts draw INDEPENDENT normals: identical contracts give different estimates
    same = uni.price_batch(np.full(4, 100.0), np.full(4, 100.0), np.ones(4), np.full(4, 0.05), np.full(4, 0.2), "call")
    assert len(set(same.tolist())) == 4
    assert uni.price_batch(np.array([]), np.array([]), np.array([]), np.array([]), np.array([]), "call").shape == (0,)


def test_delta_gamma_batch(uni, golden):
    g = golden["uni"]
    b = {k: np.array(v) for k, v in g["batch"].items()}
    d, gm = uni.delta_gamma_batch(b["S"], b["K"], b["T"], b["r"], b["sigma"], "call", b["q"], h=1.0)
    assert len(d) == len(gm) == 5                                               # reference test :449-462
    for j in range(5):
        assert d[j] == pytest.approx(g["delta_gamma_batch_h1"][0][j], abs=0.02)
        assert gm[j] == pytest.approx(g["delta_gamma_batch_h1"][1][j], abs=0.004)
        ex = orc.bs_greeks(b["S"][j], b["K"][j], b["T"][j], b["r"][j], b["sigma"][j], "call", b["q"][j])
        assert d[j] == pytest.approx(ex["delta"], abs=0.02) and gm[j] == pytest.approx(ex["gamma"], abs=0.004)
    # one fused launch of 3n contracts == three price_batch calls (stream tag j shared by the three copies)
    lo, mid, hi = (uni.price_batch(b["S"] + s, b["K"], b["T"], b["r"], b["sigma"], "call", b["q"]) for s in (-1.0, 0.0, 1.0))
    assert np.allclose(d, (hi - lo) / 2.0, rtol=1e-10, atol=0) and np.allclose(gm, hi - 2 * mid + lo, rtol=1e-7, atol=1e-10)
    d4, g4 = uni.delta_gamma_batch(b["S"], b["K"], b["T"], b["r"], b["sigma"], "call", b["q"])
    assert np.allclose(d4, g["delta_gamma_batch"][0], atol=0.02) and np.isfinite(g4).all()


def test_large_batch_one_launch():
    rng = np.random.default_rng(3)
    n = 3000
    S, K = rng.uniform(80, 120, n), rng.uniform(80, 120, n)
    T, r, v = rng.uniform(0.1, 2.0, n), rng.uniform(0.0, 0.08, n), rng.uniform(0.1, 0.5, n)
    u = MonteCarloPricerUni(4096, 16, 9)
    got = u.price_batch(S, K, T, r, v, "put", 0.01)
    bs = np.array([ol.black_scholes(S[j], K[j], T[j], r[j], v[j], "put", 0.01) for j in range(n)])
    st = _hip.european_multi(S, K, T, r, v, 0.01, False, 4096, 16, 9)
    live = bs > 0.5                      # deep out-of-the-money puts have (almost) no paying path: se ~ 0
    z = (got[live] - bs[live]) / st["std_error"][live]
    assert live.sum() > 2000
    # unbiased, and UNDER-dispersed: std_error is the reference's naive estimator that treats the 2N
    # antithetic samples as independent (monte_carlo.py:148-150), which overstates the true error
    assert np.abs(z).max() < 5.0 and abs(z.mean()) < 0.1 and 0.4 < z.std() < 1.05
    assert np.all(got[~live] >= 0) and np.all(np.abs(got[~live] - bs[~live]) < 0.5)
