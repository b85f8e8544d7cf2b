"""GPU parity of BarrierOption / LookbackOption: same-stream C checker
(tight), golden reference prices (3 sigma), and the reference's own assertions
(tests/test_exotic_options.py:120-193)."""
import math

import numpy as np
import pytest

import optionslab_amd as ol
from optionslab_amd import _hip
from oracle import philox_oracle as po

pytestmark = pytest.mark.gpu
P = (100.0, 100.0, 1.0, 0.05, 0.2)
REL = 2e-6


@pytest.mark.parametrize("payoff,level,call,anti,N,M", [
    (0, 120.0, True, False, 20000, 64), (1, 120.0, True, False, 20000, 64), (2, 80.0, False, False, 20000, 64),
    (3, 80.0, False, True, 5001, 13), (0, 100.0, True, False, 1000, 16), (4, 0.0, True, False, 20000, 50),
    (4, 0.0, False, True, 7777, 252), (5, 0.0, True, False, 20000, 50), (5, 0.0, False, False, 3000, 7),
])
def test_matches_same_stream_checker(payoff, level, call, anti, N, M):
    if payoff <= 3:
        st = _hip.barrier(*P, 0.01, call, level, payoff, N, M, 11, anti)
    else:
        st = _hip.lookback(*P, 0.01, call, payoff == 5, N, M, 11, anti)
    sx, sxx, n = po.extrema_moments(*P, 0.01, call, payoff, level, N, M, 11, anti)
    assert st.n == n
    assert st.sum == pytest.approx(sx, rel=REL, abs=1e-9) and st.sumsq == pytest.approx(sxx, rel=4 * REL, abs=1e-9)


def test_barrier_against_reference_golden(golden):
    for c in golden["barrier"]:
        S, K, T, r, v, q = c["params"]
        o = ol.BarrierOption(S=S, K=K, T=T, r=r, sigma=v, q=q, barrier=c["barrier"], seed=c["seed"])
        price, se = o.price(c["n_paths"], c["n_steps"], c["barrier_type"], c["option_type"], return_error=True)
        assert isinstance(price, np.float64) and price >= 0
        assert abs(price - c["price"]) <= 3 * math.sqrt(2) * se + 1e-12, c        # barrier at spot: both exactly 0
    h = golden["price_barrier_helper"]
    S, K, T, r, v, level, kind, typ, n, seed = h["args"]
    assert abs(ol.price_barrier(S, K, T, r, v, level, kind, typ, n, seed) - h["value"]) < 0.25


def test_lookback_against_reference_golden(golden):
    for c in golden["lookback"]:
        S, K, T, r, v, q = c["params"]
        o = ol.LookbackOption(S=S, K=K, T=T, r=r, sigma=v, q=q, seed=c["seed"])
        price, se = o.price(c["n_paths"], c["n_steps"], c["lookback_type"], c["option_type"], return_error=True)
        assert abs(price - c["price"]) <= 3 * math.sqrt(2) * se, c
    lb = ol.LookbackOption(*P, seed=1)
    assert lb.price(20000, 50, "floating", "call") > ol.black_scholes(*P, "call")      # buys at the minimum
    assert lb.price(20000, 50, "fixed", "call") >= lb.price(20000, 50, "floating", "call") * 0.5


def test_reference_barrier_test_suite():
    B = ol.BarrierOption
    assert B(S=100, K=100, T=1.0, r=0.05, sigma=0.2, barrier=120, seed=42).price(n_paths=10000, barrier_type="up-and-out", option_type="call") >= 0
    assert B(S=100, K=100, T=1.0, r=0.05, sigma=0.2, barrier=80, seed=42).price(n_paths=10000, barrier_type="down-and-out", option_type="put") >= 0
    bs = ol.black_scholes(*P, "call")
    assert B(S=100, K=100, T=1.0, r=0.05, sigma=0.2, barrier=130, seed=42).price(n_paths=50000, barrier_type="up-and-out", option_type="call") < bs
    out = B(S=100, K=100, T=1.0, r=0.05, sigma=0.2, barrier=120, seed=42).price(n_paths=100000, barrier_type="up-and-out", option_type="call")
    inn = B(S=100, K=100, T=1.0, r=0.05, sigma=0.2, barrier=120, seed=42).price(n_paths=100000, barrier_type="up-and-in", option_type="call")
    assert abs(out + inn - bs) / bs < 0.1                                         # :158-182
    with pytest.raises(ValueError, match="positive"):
        B(S=100, K=100, T=1.0, r=0.05, sigma=0.2, barrier=0).price(barrier_type="up-and-out")


def test_in_out_parity_is_exact_on_common_paths():
    """knock-in + knock-out on the SAME paths is the European payoff path by path."""
    N, M, seed = 50_000, 32, 5
    o = _hip.barrier(*P, 0.0, True, 115.0, 0, N, M, seed)
    i = _hip.barrier(*P, 0.0, True, 115.0, 1, N, M, seed)
    sx, _sxx, _n = po.asian_moments(*P, 0.0, True, False, 1, 1, seed)      # warm the checker (unused)
    eu_sum = o.sum + i.sum
    # European payoff from the per-step path (not the summed-normal kernel): via the checker's extrema with an unreachable barrier
    sx_eu, _, _ = po.extrema_moments(*P, 0.0, True, 0, 1e9, N, M, seed)
    assert eu_sum == pytest.approx(sx_eu, rel=REL)
    assert _hip.barrier(*P, 0.0, True, 100.0, 0, 1000, 8, 1).sum == 0.0   # barrier at spot: crossed at t = 0
    with pytest.raises(ol.AccelerationError):
        _hip.barrier(*P, 0.0, True, -1.0, 0, 1000, 8, 1)
    with pytest.raises(ol.AccelerationError):
        _hip.barrier(*P, 0.0, True, 120.0, 7, 1000, 8, 1)


def test_greeks_through_exotic_adapter():
    ad = ol.ExoticAdapter(ol.BarrierOption(*P, barrier=130.0, seed=3), n_paths=200_000, n_steps=64, barrier_type="up-and-out")
    g = ol.compute_greeks_unified(ad, *P, "call", 0.0, include_second_order=False)
    assert 0 < g["price"] < ol.black_scholes(*P, "call") and -1 < g["delta"] < 1
    ad2 = ol.ExoticAdapter(ol.LookbackOption(*P, seed=3), n_paths=100_000, n_steps=64, lookback_type="floating")
    g2 = ol.compute_greeks_unified(ad2, *P, "call", 0.0, include_second_order=False)
    assert g2["price"] > 0 and g2["vega"] > 0
    # floating lookback is homogeneous of degree 1 in S: delta = price / S
    assert g2["delta"] == pytest.approx(g2["price"] / 100.0, rel=1e-6)


# ------------------------------------------------------------------ Heston
H = dict(kappa=2.0, theta=0.04, sigma_v=0.3, rho=-0.7, v0=0.04)


@pytest.mark.parametrize("call,anti,N,M,model", [
    (True, False, 20000, 64, (2.0, 0.04, 0.3, -0.7, 0.04)), (False, True, 5001, 13, (1.5, 0.06, 0.5, -0.5, 0.03)),
    (False, False, 4000, 101, (3.0, 0.02, 0.8, 0.3, 0.05)),      # odd step count; Feller violated: v is truncated at 0
    (True, True, 3000, 9, (2.0, 0.04, 0.3, -0.7, -0.01)),        # v0 < 0 (C ABI only): the first step sees v+ = 0
    (True, False, 1000, 1, (2.0, 0.04, 0.3, -0.7, -0.5)),
])
def test_heston_matches_same_stream_checker(call, anti, N, M, model):
    st = _hip.heston(100.0, 95.0, 1.5, 0.03, 0.01, call, *model, N, M, 21, anti)
    sx, sxx, n = po.heston_moments(100.0, 95.0, 1.5, 0.03, 0.01, call, *model, N, M, 21, anti)
    assert st.n == n
    assert st.sum == pytest.approx(sx, rel=REL) and st.sumsq == pytest.approx(sxx, rel=4 * REL)


def test_heston_against_reference_golden(golden):
    import warnings
    for c in golden["heston"]:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            hp = ol.HestonPricer(*c["model"])
        S, K, T, r, q = c["args"]
        price, se = hp.price_monte_carlo(S, K, T, r, q, c["option_type"], c["n_paths"], c["n_steps"], c["seed"], return_error=True)
        assert isinstance(price, np.float64)
        assert abs(price - c["mc"]) <= 3 * math.sqrt(2) * se, c
        # the reference's semi-analytic formula is mirrored bit-for-bit-ish (it is NOT an accuracy anchor: it
        # disagrees with the reference's own Monte Carlo by 13-80 %)
        assert hp.price_european(S, K, T, r, q, c["option_type"]) == pytest.approx(c["semi_analytic"], rel=1e-12)


def test_heston_limits_and_parity():
    S, K, T, r, q = 100.0, 100.0, 1.0, 0.05, 0.02
    hp = ol.HestonPricer(kappa=2.0, theta=0.04, sigma_v=1e-6, rho=0.0, v0=0.04)      # no vol-of-vol: Black-Scholes, sigma = 0.2
    p, se = hp.price_monte_carlo(S, K, T, r, q, "call", 400_000, 64, 3, return_error=True)
    assert abs(p - ol.black_scholes(S, K, T, r, 0.2, "call", q)) <= 3.5 * se
    h2 = ol.HestonPricer(**H)
    c, se_c = h2.price_monte_carlo(S, K, T, r, q, "call", 400_000, 128, 9, return_error=True)
    pt, se_p = h2.price_monte_carlo(S, K, T, r, q, "put", 400_000, 128, 9, return_error=True)
    assert abs((c - pt) - (S * math.exp(-q * T) - K * math.exp(-r * T))) <= 3.5 * math.hypot(se_c, se_p)
    assert h2.price_monte_carlo(S, K, T, r, q, "call", 1000, 16, 5) == h2.price_monte_carlo(S, K, T, r, q, "call", 1000, 16, 5)
    for bad in (dict(kappa=0), dict(theta=-1), dict(sigma_v=0), dict(rho=1.5), dict(v0=0)):
        with pytest.raises(ValueError):
            ol.HestonPricer(**{**H, **bad})
    g = ol.compute_greeks_unified(ol.HestonAdapter(h2), 100.0, 100.0, 1.0, 0.05, 0.2, "call", include_second_order=False)
    assert set(g) == {"price", "delta", "gamma", "vega", "theta", "rho"} and h2.v0 == 0.04


# ------------------------------------------------------------------ autocallable / cliquet
AC = dict(autocall_barrier=1.0, coupon_barrier=0.8, coupon_rate=0.10, ki_barrier=0.6)


@pytest.mark.parametrize("anti,N,M,freq,kw", [(False, 20000, 252, 21, {}), (True, 5001, 100, 30, dict(autocall_barrier=1.05, ki_barrier=0.9)),
                                              (False, 3000, 50, 7, dict(coupon_barrier=0.95, coupon_rate=0.2)), (False, 2000, 13, 13, {})]
                         # observation dates against the four-step Philox blocks: every alignment of the fast / slow path
                         + [(a, 600, M, f, {}) for a in (False, True) for M, f in ((1, 1), (8, 1), (9, 4), (21, 5), (23, 8), (40, 9), (64, 64), (100, 33))])
def test_autocallable_matches_same_stream_checker(anti, N, M, freq, kw):
    k = {**AC, **kw}
    st = _hip.autocallable(100.0, 1.0, 0.05, 0.2, 0.01, k["autocall_barrier"], k["coupon_barrier"], k["coupon_rate"], k["ki_barrier"], freq, N, M, 5, anti)
    sx, sxx, n = po.autocall_moments(100.0, 1.0, 0.05, 0.2, 0.01, k["autocall_barrier"], k["coupon_barrier"], k["coupon_rate"], k["ki_barrier"], freq, N, M, 5, anti)
    assert st.n == n
    assert st.sum == pytest.approx(sx, rel=REL) and st.sumsq == pytest.approx(sxx, rel=4 * REL)
    assert st.price == pytest.approx(sx / n, rel=REL)            # payoffs carry their own discount: no outer factor


@pytest.mark.parametrize("anti,N,M,periods,kw", [(False, 20000, 252, 12, {}), (True, 5001, 100, 7, dict(local_cap=0.03, local_floor=-0.02, global_cap=0.2, global_floor=0.02)),
                                                 (False, 3000, 50, 50, {}), (False, 2000, 10, 3, {})]
                         # reset dates against the four-step Philox blocks and the 16-normal fp32 groups; trailing unused steps
                         + [(a, 600, M, p_, {}) for a in (False, True) for M, p_ in ((1, 1), (8, 2), (9, 2), (21, 4), (23, 1), (40, 3), (67, 1), (130, 2))])
def test_cliquet_matches_same_stream_checker(anti, N, M, periods, kw):
    k = {**dict(local_cap=0.05, local_floor=-0.05, global_cap=0.30, global_floor=0.0), **kw}
    st = _hip.cliquet(100.0, 1.0, 0.05, 0.2, 0.01, k["local_cap"], k["local_floor"], k["global_cap"], k["global_floor"], periods, N, M, 5, anti)
    sx, sxx, n = po.cliquet_moments(100.0, 1.0, 0.05, 0.2, 0.01, k["local_cap"], k["local_floor"], k["global_cap"], k["global_floor"], periods, N, M, 5, anti)
    assert st.n == n
    assert st.sum == pytest.approx(sx, rel=REL) and st.sumsq == pytest.approx(sxx, rel=4 * REL)


def test_autocallable_and_cliquet_against_reference_golden(golden):
    for c in golden["autocallable"]:
        o = ol.AutocallableOption(S=100.0, K=100.0, T=1.0, r=0.05, sigma=0.2, seed=c["seed"], **c["kwargs"])
        price, se = o.price(c["n_paths"], c["n_steps"], c["observation_freq"], return_error=True)
        assert isinstance(price, np.float64) and 0.5 < price < 1.2
        assert abs(price - c["price"]) <= 3 * math.sqrt(2) * se, c
    for c in golden["cliquet"]:
        o = ol.CliquetOption(S=100.0, K=100.0, T=1.0, r=0.05, sigma=0.2, seed=c["seed"], **c["kwargs"])
        price, se = o.price(c["n_paths"], c["n_steps"], c["n_periods"], return_error=True)
        assert abs(price - c["price"]) <= 3 * math.sqrt(2) * se, c
    with pytest.raises(ol.AccelerationError):
        _hip.autocallable(100.0, 1.0, 0.05, 0.2, 0.0, 1.0, 0.8, 0.1, 0.6, 300, 100, 252, 1)     # no observation date
    with pytest.raises(ol.AccelerationError):
        _hip.cliquet(100.0, 1.0, 0.05, 0.2, 0.0, 0.05, -0.05, 0.3, 0.0, 300, 100, 252, 1)
    # bounds: a cliquet pays between global_floor and global_cap of spot, discounted
    p = ol.CliquetOption(100.0, 100.0, 1.0, 0.05, 0.2, seed=1).price(20000)
    assert 0.0 <= p <= 30.0 * math.exp(-0.05)


# ------------------------------------------------------------------ American (Longstaff-Schwartz)
@pytest.mark.parametrize("S,K,T,r,v,q,call,N,M,deg", [
    (100.0, 100.0, 1.0, 0.05, 0.2, 0.0, False, 50000, 50, 3), (90.0, 100.0, 0.5, 0.03, 0.3, 0.0, False, 20000, 25, 2),
    (100.0, 100.0, 1.0, 0.05, 0.2, 0.08, True, 20000, 40, 3), (100.0, 100.0, 1.0, 0.05, 0.2, 0.0, False, 5000, 1, 3),
    (100.0, 95.0, 1.0, 0.05, 0.25, 0.0, False, 7001, 13, 4), (100.0, 100.0, 1.0, 0.05, 0.2, 0.0, False, 3000, 2, 1),
    # the launch shapes of the larger sizes (one workgroup per compute unit, a thread's paths 2 / 4 / 8 at a time, ragged last trip)
    (100.0, 100.0, 1.0, 0.05, 0.2, 0.0, False, 100_003, 6, 3), (100.0, 100.0, 1.0, 0.05, 0.2, 0.0, False, 200_001, 5, 3),
    (100.0, 100.0, 1.0, 0.05, 0.2, 0.0, False, 400_003, 6, 3),
])
def test_american_matches_same_stream_checker(S, K, T, r, v, q, call, N, M, deg):
    """Same paths, same regression algebra: exercise decisions agree path by path, so the sums agree to rounding."""
    st = _hip.american_lsm(S, K, T, r, v, q, call, N, M, deg, 42)
    sx, sxx, n = po.american_lsm(S, K, T, r, v, q, call, N, M, deg, 42)
    assert st.n == n
    assert st.sum == pytest.approx(sx, rel=1e-7) and st.sumsq == pytest.approx(sxx, rel=1e-7)
    assert st.price == pytest.approx(sx / n, rel=1e-7)


def test_american_against_reference_golden_and_bounds(golden):
    for c in golden["american"]:
        S, K, T, r, v, q = c["params"]
        o = ol.AmericanOption(S=S, K=K, T=T, r=r, sigma=v, q=q, seed=c["seed"])
        price, se = o.price(c["n_paths"], c["n_steps"], c["option_type"], c["poly_degree"], return_error=True)
        assert isinstance(price, np.float64)
        assert abs(price - c["price"]) <= 3 * math.sqrt(2) * se, c
    put = ol.AmericanOption(*P, seed=3).price(100_000, 50, "put")
    assert put > ol.black_scholes(*P, "put") + 0.2                                   # early-exercise premium
    assert put >= 0 and put < 100.0
    call, se = ol.AmericanOption(*P, seed=3).price(100_000, 50, "call", return_error=True)
    assert abs(call - ol.black_scholes(*P, "call")) <= 4 * se + 0.15                 # no dividends: never exercised early (LSM is biased low)
    assert ol.price_american(100.0, 100.0, 1.0, 0.05, 0.2, "put", 20000, 7) == ol.AmericanOption(*P, seed=7).price(20000, 50, "put")
    with pytest.raises(ol.AccelerationError):
        _hip.american_lsm(*P, 0.0, False, 1000, 10, 7, 1)
    assert math.isnan(ol.AmericanOption(-1.0, 100.0, 1.0, 0.05, 0.2, seed=1).price(100, 4))


# ------------------------------------------------------------------ jump diffusion (Merton, Kou)
@pytest.mark.parametrize("kou,lam,a1,a2,a3,call,N,M", [
    (False, 0.5, -0.1, 0.2, 0.0, True, 20000, 50), (False, 40.0, 0.02, 0.1, 0.0, False, 5001, 13),      # lambda dt = 3: multi-jump steps
    (True, 1.0, 0.4, 10.0, 5.0, True, 20000, 50), (True, 60.0, 0.6, 25.0, 20.0, False, 3000, 7),         # up to ~20 jumps per step
    (False, 0.0, -0.1, 0.2, 0.0, True, 4000, 10),
    (False, 5.0, -0.1, 0.2, 0.0, True, 3000, 1), (True, 5.0, 0.4, 10.0, 5.0, False, 3000, 1), (False, 20.0, 0.05, 0.1, 0.0, True, 3000, 2),   # one block = two steps
])
def test_jump_diffusion_matches_same_stream_checker(kou, lam, a1, a2, a3, call, N, M):
    st = _hip.jump_diffusion(100.0, 100.0, 1.0, 0.05, 0.2, 0.01, call, kou, lam, a1, a2, a3, N, M, 17)
    sx, sxx, n = po.jump_moments(100.0, 100.0, 1.0, 0.05, 0.2, 0.01, call, kou, lam, a1, a2, a3, N, M, 17)
    assert st.n == n
    assert st.sum == pytest.approx(sx, rel=REL) and st.sumsq == pytest.approx(sxx, rel=4 * REL)


def test_jump_diffusion_against_reference_golden_and_series(golden):
    for c in golden["merton"]:
        S, K, T, r, v, q = c["args"]
        jd = ol.MertonJumpDiffusion(*c["model"])
        assert jd.kappa == pytest.approx(c["kappa"], rel=1e-14)
        assert jd.price(S, K, T, r, v, c["option_type"], q) == pytest.approx(c["series"], rel=1e-12)
        price, se = jd.price_monte_carlo(S, K, T, r, v, c["option_type"], q, c["n_paths"], c["n_steps"], c["seed"], return_error=True)
        assert isinstance(price, np.float64)
        assert abs(price - c["mc"]) <= 3 * math.sqrt(2) * se, c
        big, se_big = jd.price_monte_carlo(S, K, T, r, v, c["option_type"], q, 1_000_000, c["n_steps"], 3, return_error=True)
        assert abs(big - c["series"]) <= 3.5 * se_big, c                        # the MC converges to Merton's series
    for c in golden["kou"]:
        S, K, T, r, v, q = c["args"]
        kj = ol.KouJumpDiffusion(*c["model"])
        assert kj.kappa == pytest.approx(c["kappa"], rel=1e-14)
        price, se = kj.price_monte_carlo(S, K, T, r, v, c["option_type"], q, c["n_paths"], c["n_steps"], c["seed"], return_error=True)
        assert abs(price - c["mc"]) <= 3 * math.sqrt(2) * se, c
    # martingale check for Kou: E[S_T] = S e^{(r-q)T}  <=>  call(K -> 0) = S e^{-qT}
    kj = ol.KouJumpDiffusion(3.0, 0.4, 10.0, 5.0)
    deep, se = kj.price_monte_carlo(100.0, 1e-9, 1.0, 0.05, 0.2, "call", 0.02, 1_000_000, 32, 5, return_error=True)
    assert abs(deep - 100.0 * math.exp(-0.02)) <= 3.5 * se
    for bad in (dict(lambda_j=-1, mu_j=0, sigma_j=0.1), dict(lambda_j=1, mu_j=0, sigma_j=-0.1)):
        with pytest.raises(ValueError):
            ol.MertonJumpDiffusion(**bad)
    for bad in ((1, 1.5, 10, 5), (1, 0.5, 1.0, 5), (1, 0.5, 10, 0)):
        with pytest.raises(ValueError):
            ol.KouJumpDiffusion(*bad)


# ------------------------------------------------------------------ path generators and the exercise boundary
def test_heston_paths_against_checker_reference_and_pricer(golden):
    import warnings
    from oracle import numpy_reference as orc
    g = golden["heston_paths"]
    S, T, r, q, N, M, seed = g["args"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        hp = ol.HestonPricer(*g["model"])
    spot, var = hp.simulate_paths(S, T, r, q, N, M, seed)                       # heston.py:257-305
    assert list(spot.shape) == g["shape"] and list(var.shape) == g["shape"] and spot.flags["C_CONTIGUOUS"]
    assert np.all(spot[:, 0] == S) and np.all(var[:, 0] == hp.v0) and np.all(var >= 0)
    ws, wv = po.heston_paths(S, T, r, q, *g["model"], N, M, seed)
    assert np.allclose(spot, ws.T, rtol=4 * REL, atol=0) and np.allclose(var, wv.T, rtol=4 * REL, atol=2e-7)     # v near 0 is a difference of O(theta) terms: absolute bound
    # the reference's paths, statistically: column means of spot and variance within 4 standard errors
    se_s, se_v = spot.std(axis=0) / math.sqrt(N), var.std(axis=0) / math.sqrt(N)
    assert np.all(np.abs(spot.mean(axis=0) - np.array(g["spot_col_mean"])) <= 4 * math.sqrt(2) * se_s + 1e-12)
    assert np.all(np.abs(var.mean(axis=0) - np.array(g["var_col_mean"])) <= 4 * math.sqrt(2) * se_v + 1e-12)
    # the terminal column is what price_monte_carlo integrates for the same seed
    price = hp.price_monte_carlo(S, 95.0, T, r, q, "call", N, M, seed)
    assert price == pytest.approx(math.exp(-r * T) * np.maximum(spot[:, -1] - 95.0, 0).mean(), rel=1e-12)
    odd_s, odd_v = ol.HestonPricer(3.0, 0.02, 0.8, 0.3, 0.05).simulate_paths(S, T, r, q, 257, 7, 3)    # odd steps, truncation at 0
    cs, cv = po.heston_paths(S, T, r, q, 3.0, 0.02, 0.8, 0.3, 0.05, 257, 7, 3)
    assert odd_s.shape == (257, 8) and np.allclose(odd_s, cs.T, rtol=4 * REL) and np.allclose(odd_v, cv.T, rtol=4 * REL, atol=2e-7)
    assert orc.heston_simulate_paths(S, T, r, q, *g["model"], 5, M, seed)[0].shape == (5, M + 1)           # same layout as the restatement
    neg_s, neg_v = _hip.heston_paths(S, T, r, q, 2.0, 0.04, 0.3, -0.7, -0.002, 300, 5, 3)                   # v0 < 0 at the C ABI
    cs, cv = po.heston_paths(S, T, r, q, 2.0, 0.04, 0.3, -0.7, -0.002, 300, 5, 3)
    assert np.allclose(neg_s, cs, rtol=4 * REL) and np.allclose(neg_v, cv, rtol=4 * REL, atol=2e-7) and np.all(neg_v[0] == -0.002)


def test_jump_paths_against_checker_and_martingale(golden):
    g = golden["merton_path"]
    S, T, r, v, q, M, seed = g["args"]
    jd = ol.MertonJumpDiffusion(*g["model"])
    one = jd.simulate_path(S, T, r, v, q, M, seed)                             # jump_diffusion.py:227-272
    assert one.shape == (M + 1,) == np.array(g["path"]).shape and one[0] == S == g["path"][0]
    N = 40_000
    paths = jd.simulate_paths(S, T, r, v, q, N, M, seed)
    assert paths.shape == (N, M + 1) and np.array_equal(paths[0], one)
    assert np.allclose(paths, po.jump_paths(S, T, r, v, q, False, *g["model"], 0.0, N, M, seed).T, rtol=4 * REL)
    t = np.arange(M + 1) * (T / M)
    se = paths.std(axis=0) / math.sqrt(N)
    assert np.all(np.abs(paths.mean(axis=0) - S * np.exp((r - q) * t)) <= 4 * se + 1e-12)      # compensated drift (:207)
    price = jd.price_monte_carlo(S, 100.0, T, r, v, "put", q, N, M, seed)
    assert price == pytest.approx(math.exp(-r * T) * np.maximum(100.0 - paths[:, -1], 0).mean(), rel=1e-12)
    kou = ol.KouJumpDiffusion(3.0, 0.6, 25.0, 20.0)
    kp = kou.simulate_paths(S, T, r, v, q, 20_000, 10, 5)
    assert np.allclose(kp, po.jump_paths(S, T, r, v, q, True, 3.0, 0.6, 25.0, 20.0, 20_000, 10, 5).T, rtol=4 * REL)
    assert np.all(np.abs(kp.mean(axis=0) - S * np.exp((r - q) * np.arange(11) * (T / 10))) <= 4 * kp.std(axis=0) / math.sqrt(20_000) + 1e-12)


def test_early_exercise_boundary(golden):
    from oracle import numpy_reference as orc
    for c in golden["exercise_boundary"]:
        S, K, T, r, v, q = c["params"]
        N, M, typ = c["n_paths"], c["n_steps"], c["option_type"]
        times, b = ol.AmericanOption(S, K, T, r, v, q, seed=c["seed"]).early_exercise_boundary(N, M, typ)   # exotic_options.py:309-345
        assert np.array_equal(times, np.array(c["times"])) and b.shape == (M + 1,)
        # exact: NumPy's percentile of the very paths the device selected from (rows t >= 1 of the LSM path set)
        paths = _hip.gbm_paths(S, T, r, v, q, N, M, c["seed"]).T
        want = orc.exercise_boundary_from_paths(paths, K, typ)
        assert np.array_equal(b[1:], want[1:], equal_nan=True)
        assert np.isnan(b[0]) or b[0] == pytest.approx(S, rel=1e-14)             # t = 0: exp(log S) vs K, as the reference
        ref = np.array([np.nan if x is None else x for x in c["boundary"]])
        assert np.array_equal(np.isnan(b[1:]), np.isnan(ref[1:]))
        # the reference's own draw differs by sampling noise only: tight where many paths are in the money
        itm = ((paths[:, 1:] > K) if typ == "call" else (paths[:, 1:] < K)).sum(axis=0)
        rtol = np.where(itm >= 1500, 0.015, 0.06)
        assert np.all(np.isnan(ref[1:]) | (np.abs(b[1:] - ref[1:]) <= rtol * np.abs(ref[1:])))
    # many paths (several strides per workgroup), few in the money, and none at all
    times, b = ol.AmericanOption(100.0, 60.0, 0.25, 0.05, 0.2, seed=1).early_exercise_boundary(300_000, 6, "put")
    paths = _hip.gbm_paths(100.0, 0.25, 0.05, 0.2, 0.0, 300_000, 6, 1).T
    assert np.array_equal(b[1:], orc.exercise_boundary_from_paths(paths, 60.0, "put")[1:], equal_nan=True)
    assert np.isnan(b[0]) and np.isnan(b[1])
    assert np.all(np.isnan(ol.AmericanOption(100.0, 1.0, 0.1, 0.05, 0.1, seed=1).early_exercise_boundary(1000, 4, "put")[1]))


def test_path_matrix_layouts_agree_bitwise():
    """path_major: the reference's (n_paths, n_steps + 1) C order written by the kernel == the transpose of the time-major rows."""
    for N, M in ((1000, 12), (257, 7), (70_000, 3)):
        a = _hip.gbm_paths(100.0, 1.0, 0.05, 0.2, 0.01, N, M, 9, path_major=True)
        assert a.shape == (N, M + 1) and a.flags["C_CONTIGUOUS"] and np.array_equal(a, _hip.gbm_paths(100.0, 1.0, 0.05, 0.2, 0.01, N, M, 9).T)
        s1, v1 = _hip.heston_paths(100.0, 1.0, 0.05, 0.01, 2.0, 0.04, 0.3, -0.7, 0.04, N, M, 9, path_major=True)
        s0, v0 = _hip.heston_paths(100.0, 1.0, 0.05, 0.01, 2.0, 0.04, 0.3, -0.7, 0.04, N, M, 9)
        assert s1.shape == (N, M + 1) and np.array_equal(s1, s0.T) and np.array_equal(v1, v0.T)
        for kou, mdl in ((False, (3.0, -0.1, 0.2, 0.0)), (True, (3.0, 0.6, 25.0, 20.0))):
            j1 = _hip.jump_paths(100.0, 1.0, 0.05, 0.2, 0.01, kou, *mdl, N, M, 9, path_major=True)
            assert j1.shape == (N, M + 1) and np.array_equal(j1, _hip.jump_paths(100.0, 1.0, 0.05, 0.2, 0.01, kou, *mdl, N, M, 9).T)


def test_large_results_come_home_bit_identical_through_the_staged_copy():
    """Results of 32 MB and more leave the device in 16 MB chunks through two pinned staging buffers while host threads copy the
    previous chunk into the caller's buffer (olmc.hip copy_to_host; 41 GB/s end to end where one hipMemcpy into a fresh pageable
    buffer made 17).  Bytes are copied, never interpreted: the arrays must equal the single-copy form's (OLMC_TUNE_STAGED_COPY = -1)
    bit for bit -- sizes that are not a multiple of a chunk, of a page or of the thread count, both layouts, both Heston matrices,
    a terminal array, twice over (the staging buffers are reused)."""
    def grab():
        out = []
        for N, M, pm in ((30_011, 252, True), (17_000, 251, False), (262_144, 15, True)):          # 60.8 / 34.3 / 33.6 MB
            out.append(_hip.gbm_paths(100.0, 1.0, 0.05, 0.2, 0.01, N, M, 9, path_major=pm))
        out.extend(_hip.heston_paths(100.0, 1.0, 0.05, 0.01, 2.0, 0.04, 0.3, -0.7, 0.04, 45_001, 101, 9, path_major=True))
        out.append(_hip.european_terminal(100.0, 1.0, 0.05, 0.2, 0.0, 3_000_001, 16, 5, True))      # 48 MB [pos | neg]
        return out
    staged = grab()
    again = grab()
    _hip.tune(_hip.TUNE_STAGED_COPY, -1)
    try:
        direct = grab()
    finally:
        _hip.tune(_hip.TUNE_STAGED_COPY, 0)
    assert all(a.nbytes >= 32 << 20 for a in staged)
    for a, b, c in zip(staged, again, direct):
        assert a.shape == c.shape and a.tobytes() == c.tobytes() and b.tobytes() == c.tobytes()
        assert np.isfinite(a).all() and (a >= 0).all()          # (Heston's variance touches 0 under full truncation)


def _heston_call_characteristic_function(S, K, T, r, q, kappa, theta, sv, rho, v0):
    """Heston (1993) call by the P1/P2 integrals in the branch-cut-safe ("little trap") form -- a test-only analytic
    anchor, written from the published formula.  (The reference's own price_european, heston.py:131-182, is mirrored
    for API parity but is off by 13-80 % from its own Monte Carlo; this one agrees with both Monte Carlos.)"""
    from scipy.integrate import quad
    x = math.log(S)

    def cf(u, j):
        uj, bj = (0.5, kappa - rho * sv) if j == 1 else (-0.5, kappa)
        d = np.sqrt((rho * sv * 1j * u - bj) ** 2 - sv**2 * (2 * uj * 1j * u - u * u))
        g = (bj - rho * sv * 1j * u - d) / (bj - rho * sv * 1j * u + d)
        C = (r - q) * 1j * u * T + kappa * theta / sv**2 * ((bj - rho * sv * 1j * u - d) * T - 2 * np.log((1 - g * np.exp(-d * T)) / (1 - g)))
        D = (bj - rho * sv * 1j * u - d) / sv**2 * ((1 - np.exp(-d * T)) / (1 - g * np.exp(-d * T)))
        return np.exp(C + D * v0 + 1j * u * x)

    def P(j):
        return 0.5 + quad(lambda u: np.real(np.exp(-1j * u * math.log(K)) * cf(u, j) / (1j * u)), 1e-8, 200, limit=400)[0] / math.pi

    return S * math.exp(-q * T) * P(1) - K * math.exp(-r * T) * P(2)


def test_heston_monte_carlo_against_the_characteristic_function_price(golden):
    import warnings
    for c in golden["heston"][:3]:
        kappa, theta, sv, rho, v0 = c["model"]
        S, K, T, r, q = c["args"]
        call = _heston_call_characteristic_function(S, K, T, r, q, kappa, theta, sv, rho, v0)
        want = call if c["option_type"] == "call" else call - S * math.exp(-q * T) + K * math.exp(-r * T)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            hp = ol.HestonPricer(kappa, theta, sv, rho, v0)
        price, se = hp.price_monte_carlo(S, K, T, r, q, c["option_type"], 1 << 21, 512, 7, antithetic=True, return_error=True)
        assert abs(price - want) <= 3.5 * se + 0.01, (c["model"], price, want, se)      # + Euler bias of 512 full-truncation steps
        assert abs(c["mc"] - want) <= 0.12                                              # the reference's own Monte Carlo agrees too


@pytest.mark.parametrize("S,K,T,r,v,call,N,M,deg", [
    (80.0, 85.0, 1.0, 0.03, 0.1, False, 100_000, 12, 4),       # deep in the money, low vol: S_t / K of the in-the-money paths spans a few percent
    (100.0, 100.0, 1.0, 0.05, 0.2, False, 100_000, 50, 4), (120.0, 100.0, 0.5, 0.02, 0.15, True, 60_000, 20, 4),
    (100.0, 100.0, 1.0, 0.05, 0.2, False, 236, 12, 4),          # few paths, where the property hunt found device and checker apart before the regressor was standardised
])
def test_american_where_the_regression_is_ill_conditioned(S, K, T, r, v, call, N, M, deg):
    """Degree-4 fits over a narrow in-the-money range: in powers of S / K the normal equations have a condition number beyond 1e14;
    the device fits in the standardised regressor z = (S/K - c_t) / w_t (olmc_kernels.h, LsmScale).  Against (i) the checker on the
    same paths -- the same exercise decisions, sums to 2e-6 -- and (ii) the reference's own algorithm (exotic_options.py:237-305,
    np.linalg.lstsq on raw powers of S, restated in oracle/numpy_reference.py) on its own normals: 3 sigma of the two standard errors."""
    from oracle import numpy_reference as nr
    kind = "call" if call else "put"
    st = _hip.american_lsm(S, K, T, r, v, 0.01, call, N, M, deg, 11)
    sx, sxx, n = po.american_lsm(S, K, T, r, v, 0.01, call, N, M, deg, 11)
    assert st.n == n and st.sum == pytest.approx(sx, rel=2e-6) and st.sumsq == pytest.approx(sxx, rel=4e-6)
    ref = nr.american_price(S, K, T, r, v, 0.01, seed=5, n_paths=N, n_steps=M, option_type=kind, poly_degree=deg)
    assert abs(st.price - ref) <= 3.0 * math.sqrt(2.0) * st.std_error + 1e-12, (st.price, ref, st.std_error)
    assert st.price >= max(K - S, 0.0) - 3 * st.std_error if not call else st.price >= max(S - K, 0.0) - 3 * st.std_error


def test_american_where_the_moment_matrix_is_singular():
    """a numerically singular moment matrix -- a handful of in-the-money paths (count = degree + 2: the fit interpolates),
    in-the-money prices that all but coincide (vol 1e-7: the regressor's width is clamped) -- must not hand the next date
    coefficients that are rounding noise divided by rounding noise.  A pivot below 1e-11 of its own diagonal entry pins that unknown
    to zero (olmc_kernels.h LsmFit; restated in the checker).  Device = checker on the same paths wherever the decisions are robust,
    every price finite and between the European value's neighbourhood and the strike."""
    for N in (5, 6, 7, 9, 12):                                  # a few paths: some dates have exactly degree + 2 of them in the money
        for deg in (3, 4):
            for seed in range(12):
                st = _hip.american_lsm(100.0, 110.0, 1.0, 0.05, 0.2, 0.0, False, N, 6, deg, seed)
                sx, sxx, n = po.american_lsm(100.0, 110.0, 1.0, 0.05, 0.2, 0.0, False, N, 6, deg, seed)
                assert st.n == n == N and math.isfinite(st.price) and 0.0 <= st.price <= 110.0
                assert st.sum == pytest.approx(sx, rel=1e-5, abs=1e-9), (N, deg, seed, st.sum, sx)
    for v in (1e-7, 1e-5):                                      # the in-the-money prices of a date differ in the 8th / 6th digit
        st = _hip.american_lsm(90.0, 100.0, 1.0, 0.03, v, 0.0, False, 20_000, 10, 4, 3)
        sx, sxx, n = po.american_lsm(90.0, 100.0, 1.0, 0.03, v, 0.0, False, 20_000, 10, 4, 3)
        assert math.isfinite(st.price) and st.sum == pytest.approx(sx, rel=1e-6)
        # no uncertainty left: the first exercise date (t = dt; time 0 is not one, exotic_options.py:262-296) beats every later one --
        # K e^{-r k dt} - S falls with k -- whatever the (constant) continuation estimate the pinned fit makes of the later dates
        assert st.price == pytest.approx(100.0 * math.exp(-0.03 * 0.1) - 90.0, abs=2e-3)


_BIAS_CASES = {
    # name: (device price of seed s, the reference's algorithm -- oracle/numpy_reference.py, pinned bitwise to the reference -- on ITS normals of seed s)
    "asian arithmetic call": (lambda s: _hip.asian(100.0, 100.0, 1.0, 0.05, 0.2, 0.01, True, False, 40_000, 64, s).price,
                              lambda nr, s: nr.asian_price(100.0, 100.0, 1.0, 0.05, 0.2, 0.01, s, 40_000, 64, "arithmetic", "call")),
    "asian geometric put": (lambda s: _hip.asian(100.0, 105.0, 1.0, 0.05, 0.2, 0.01, False, True, 40_000, 64, s).price,
                            lambda nr, s: nr.asian_price(100.0, 105.0, 1.0, 0.05, 0.2, 0.01, s, 40_000, 64, "geometric", "put")),
    "barrier up-and-out call": (lambda s: _hip.barrier(100.0, 100.0, 1.0, 0.05, 0.2, 0.0, True, 125.0, 0, 40_000, 64, s).price,
                                lambda nr, s: nr.barrier_price(100.0, 100.0, 1.0, 0.05, 0.2, 125.0, 0.0, s, 40_000, 64, "up-and-out", "call")),
    "barrier down-and-in put": (lambda s: _hip.barrier(100.0, 100.0, 1.0, 0.05, 0.2, 0.0, False, 85.0, 3, 40_000, 64, s).price,
                                lambda nr, s: nr.barrier_price(100.0, 100.0, 1.0, 0.05, 0.2, 85.0, 0.0, s, 40_000, 64, "down-and-in", "put")),
    "lookback floating call": (lambda s: _hip.lookback(100.0, 100.0, 1.0, 0.05, 0.2, 0.0, True, False, 40_000, 64, s).price,
                               lambda nr, s: nr.lookback_price(100.0, 100.0, 1.0, 0.05, 0.2, 0.0, s, 40_000, 64, "floating", "call")),
    "lookback fixed put": (lambda s: _hip.lookback(100.0, 100.0, 1.0, 0.05, 0.2, 0.0, False, True, 40_000, 64, s).price,
                           lambda nr, s: nr.lookback_price(100.0, 100.0, 1.0, 0.05, 0.2, 0.0, s, 40_000, 64, "fixed", "put")),
    "autocallable": (lambda s: _hip.autocallable(100.0, 1.0, 0.05, 0.2, 0.0, 1.0, 0.8, 0.10, 0.6, 21, 40_000, 126, s).price,
                     lambda nr, s: nr.autocallable_price(100.0, 1.0, 0.05, 0.2, 0.0, s, 40_000, 126, 21)),
    "cliquet": (lambda s: _hip.cliquet(100.0, 1.0, 0.05, 0.2, 0.0, 0.05, -0.05, 0.30, 0.0, 12, 40_000, 120, s).price,
                lambda nr, s: nr.cliquet_price(100.0, 1.0, 0.05, 0.2, 0.0, s, 40_000, 120, 12)),
    "heston call": (lambda s: _hip.heston(100.0, 100.0, 1.0, 0.05, 0.0, True, 2.0, 0.04, 0.3, -0.7, 0.04, 40_000, 64, s).price,
                    lambda nr, s: nr.heston_price_mc(100.0, 100.0, 1.0, 0.05, 0.0, "call", 2.0, 0.04, 0.3, -0.7, 0.04, 40_000, 64, s)),
}


@pytest.mark.parametrize("name", sorted(_BIAS_CASES))
def test_no_bias_against_the_references_algorithm_over_many_seeds(name):
    """One price within 3 sigma of one reference price (the golden tests) resolves a bias of ~3 standard errors.  Sixteen device prices
    against sixteen prices of the reference's own algorithm on its own normals resolve a quarter of that: the two means differ by less
    than 3.5 standard errors of their difference, payoff by payoff."""
    from oracle import numpy_reference as nr
    dev_fn, ref_fn = _BIAS_CASES[name]
    dev = np.array([float(dev_fn(s)) for s in range(16)])
    ref = np.array([float(ref_fn(nr, 500 + s)) for s in range(16)])
    se = math.sqrt(dev.var(ddof=1) / len(dev) + ref.var(ddof=1) / len(ref))
    assert abs(dev.mean() - ref.mean()) <= 3.5 * se, (name, dev.mean(), ref.mean(), se)


def test_american_has_no_bias_against_the_references_algorithm_over_many_seeds():
    """Sharper than one 3-sigma comparison: the mean of 24 device prices (seeds 0..23, the reference's default 50,000 x 50, degree 3)
    against the mean of 24 prices of the reference's algorithm (np.linalg.lstsq on raw powers of S, its own normals; restated in
    oracle/numpy_reference.py): the two means differ by less than 3.5 standard errors of their difference -- a fifth of what a single
    pricing can resolve -- for an at-the-money put and an in-the-money put with dividends."""
    from oracle import numpy_reference as nr
    for (S, K, T, r, v, q) in ((100.0, 100.0, 1.0, 0.05, 0.2, 0.0), (92.0, 100.0, 0.75, 0.04, 0.3, 0.02)):
        dev = np.array([_hip.american_lsm(S, K, T, r, v, q, False, 50_000, 50, 3, seed).price for seed in range(24)])
        ref = np.array([nr.american_price(S, K, T, r, v, q, seed=1000 + seed, n_paths=50_000, n_steps=50, option_type="put", poly_degree=3) for seed in range(24)])
        se = math.sqrt(dev.var(ddof=1) / len(dev) + ref.var(ddof=1) / len(ref))
        assert abs(dev.mean() - ref.mean()) <= 3.5 * se, (dev.mean(), ref.mean(), se)


def test_american_lsm_against_a_bermudan_binomial_tree():
    """Independent anchor for Longstaff-Schwartz: a CRR tree that allows exercise on the same 50 dates (40 tree steps
    between dates).  LSM's two biases (sub-optimal fitted policy: low; in-sample fit: high) are ~1e-2 here."""
    S, K, T, r, v, dates, sub = 100.0, 100.0, 1.0, 0.05, 0.2, 50, 40
    n = dates * sub
    dt = T / n
    u = math.exp(v * math.sqrt(dt))
    p = (math.exp(r * dt) - 1 / u) / (u - 1 / u)
    disc = math.exp(-r * dt)
    j = np.arange(n + 1)
    val = np.maximum(K - S * u ** (2.0 * j - n), 0.0)
    for i in range(n - 1, -1, -1):
        val = disc * (p * val[1:] + (1 - p) * val[:-1])
        if i % sub == 0 and i > 0:
            val = np.maximum(val, K - S * u ** (2.0 * np.arange(i + 1) - i))
    tree = float(val[0])
    assert 6.07 < tree < 6.09
    price, se = ol.AmericanOption(S, K, T, r, v, seed=5).price(1_000_000, dates, "put", 3, return_error=True)
    assert abs(price - tree) <= 3 * se + 0.01, (price, tree, se)
    assert price > ol.black_scholes(S, K, T, r, v, "put") + 0.4            # the early-exercise premium is there (European put 5.57)


def test_autocallable_and_cliquet_against_degenerate_closed_forms():
    """Parameter corners where the structured payoffs collapse to something with a closed form."""
    S, K, T, r, v, q = 100.0, 100.0, 1.0, 0.05, 0.2, 0.01
    cdf = lambda x: 0.5 * math.erfc(-x / math.sqrt(2.0))
    N, M = 1 << 20, 252
    # never called, coupon always paid, never knocked in: a zero-coupon bond with one coupon -- deterministic
    p, se = ol.AutocallableOption(S, K, T, r, v, q, seed=1, autocall_barrier=1e9, coupon_barrier=0.0, coupon_rate=0.08,
                                  ki_barrier=0.0).price(N, M, 21, return_error=True)
    assert p == pytest.approx((1 + 0.08 * T) * math.exp(-r * T), rel=1e-12) and se < 1e-12
    # called at the first observation whatever happens: (1 + first coupon) discounted to that date
    p, se = ol.AutocallableOption(S, K, T, r, v, q, seed=1, autocall_barrier=0.0, coupon_rate=0.12).price(N, M, 21, return_error=True)
    assert p == pytest.approx((1 + 0.12 * T / 12) * math.exp(-r * T * 21 / 252), rel=1e-12) and se < 1e-12
    # never called, no coupon, knock-in certain: 1 - (ATM put on S_T / S_0)
    p, se = ol.AutocallableOption(S, K, T, r, v, q, seed=2, autocall_barrier=1e9, coupon_barrier=1e9, ki_barrier=1e9).price(
        N, M, 21, antithetic=True, return_error=True)
    want = math.exp(-r * T) - ol.black_scholes(S, S, T, r, v, "put", q) / S
    assert abs(p - want) <= 3.5 * se, (p, want, se)
    # cliquet, ONE period, local return clipped to [0, c]: an ATM call spread
    c = 0.10
    p, se = ol.CliquetOption(S, K, T, r, v, q, seed=3, local_cap=c, local_floor=0.0, global_cap=1e9, global_floor=0.0).price(
        N, M, 1, antithetic=True, return_error=True)
    want = ol.black_scholes(S, S, T, r, v, "call", q) - ol.black_scholes(S, S * (1 + c), T, r, v, "call", q)
    assert abs(p - want) <= 3.5 * se, (p, want, se)
    # cliquet, 12 periods, each return floored at 0 and uncapped: 12 forward-start ATM calls
    n, dt = 12, T / 12
    d1 = (r - q + 0.5 * v * v) * dt / (v * math.sqrt(dt))
    want = math.exp(-r * T) * S * n * (math.exp((r - q) * dt) * cdf(d1) - cdf(d1 - v * math.sqrt(dt)))
    p, se = ol.CliquetOption(S, K, T, r, v, q, seed=4, local_cap=1e9, local_floor=0.0, global_cap=1e9, global_floor=0.0).price(
        N, M, n, antithetic=True, return_error=True)
    assert abs(p - want) <= 3.5 * se, (p, want, se)


def test_barrier_and_lookback_against_degenerate_closed_forms():
    S, K, T, r, v, q = 100.0, 105.0, 1.0, 0.05, 0.2, 0.01
    N = 1 << 20
    bs_call, bs_put = ol.black_scholes(S, K, T, r, v, "call", q), ol.black_scholes(S, K, T, r, v, "put", q)
    # a barrier nobody reaches: knock-out = European, knock-in = 0; a barrier already touched at t = 0: the other way round
    for kind, level, typ, want in (("up-and-out", 1e9, "call", bs_call), ("down-and-out", 1e-9, "put", bs_put),
                                   ("up-and-in", 100.0, "call", bs_call), ("down-and-in", 100.0, "put", bs_put)):
        p, se = ol.BarrierOption(S, K, T, r, v, barrier=level, q=q, seed=3).price(N, 64, kind, typ, antithetic=True, return_error=True)
        assert abs(p - want) <= 3.5 * se, (kind, p, want, se)
    assert ol.BarrierOption(S, K, T, r, v, barrier=1e9, q=q, seed=3).price(10_000, 16, "up-and-in", "call") == 0.0
    assert ol.BarrierOption(S, K, T, r, v, barrier=100.0, q=q, seed=3).price(10_000, 16, "up-and-out", "call") == 0.0
    # ONE monitoring date: min / max over {S_0, S_T} turns both lookbacks into ATM-struck Europeans
    atm_call, atm_put = ol.black_scholes(S, S, T, r, v, "call", q), ol.black_scholes(S, S, T, r, v, "put", q)
    for kind, typ, strike, want in (("floating", "call", K, atm_call), ("floating", "put", K, atm_put),
                                    ("fixed", "call", S, atm_call), ("fixed", "put", S, atm_put)):
        p, se = ol.LookbackOption(S, strike, T, r, v, q=q, seed=4).price(N, 1, kind, typ, antithetic=True, return_error=True)
        assert abs(p - want) <= 3.5 * se, (kind, typ, p, want, se)


def test_non_positive_barrier_levels_and_negative_maturity_follow_the_reference():
    """the reference compares barrier LEVELS in price space (exotic_options.py:445-483), so a level <= 0 lies below
    every price: `S_t >= level` always holds, `S_t <= level` never does.  The device compares logs; log() of a negative
    level is NaN and every comparison with it false -- so the host maps levels <= 0 to -inf.  And T < 0 makes sqrt(dt) NaN in
    the reference (gbm_numpy.py:37): every price is NaN, which device fmax() would swallow."""
    S, T, r, v, q = 100.0, 1.0, 0.05, 0.2, 0.01
    N, M, f, rate = 50_000, 84, 21, 0.08
    first = (1 + rate * (1 / (M // f)) * T) * math.exp(-r * f * (T / M))          # everyone is redeemed at the first observation
    for level in (0.0, -1.0):
        st = _hip.autocallable(S, T, r, v, q, level, 0.8, rate, 0.6, f, N, M, 3)
        assert st.price == pytest.approx(first, rel=1e-13) and st.std_error == pytest.approx(0.0, abs=1e-9)
    # a knock-in level <= 0 is never touched: same sums as a level no path can reach
    ref = _hip.autocallable(S, T, r, v, q, 1.0, 0.8, rate, 1e-300, f, N, M, 3)
    for level in (0.0, -0.5):
        st = _hip.autocallable(S, T, r, v, q, 1.0, 0.8, rate, level, f, N, M, 3)
        assert (st.sum, st.sumsq) == (ref.sum, ref.sumsq)
    # a coupon level <= 0 always pays: same as a level every path clears
    ref = _hip.autocallable(S, T, r, v, q, 1e9, 1e-300, rate, 0.6, f, N, M, 3)
    st = _hip.autocallable(S, T, r, v, q, 1e9, -2.0, rate, 0.6, f, N, M, 3)
    assert (st.sum, st.sumsq) == (ref.sum, ref.sumsq)
    assert math.isnan(_hip.autocallable(S, T, r, v, q, float("nan"), 0.8, rate, 0.6, f, N, M, 3).price)
    for bad in (_hip.european(100.0, 100.0, -1.0, 0.05, 0.2, 0.0, True, 1000, 8, 1), _hip.asian(100.0, 100.0, -0.5, 0.05, 0.2, 0.0, True, False, 1000, 8, 1),
                _hip.barrier(100.0, 100.0, -1.0, 0.05, 0.2, 0.0, True, 120.0, 0, 1000, 8, 1), _hip.heston(100.0, 100.0, -1.0, 0.05, 0.0, True, 2.0, 0.04, 0.3, -0.7, 0.04, 1000, 8, 1)):
        assert math.isnan(bad.price) and math.isnan(bad.sum)


def test_jump_rates_beyond_the_inversion_samplers_range_are_refused():
    """the per-step jump count is drawn by inversion from exp(-lambda dt) with at most 64 jumps; above
    lambda dt = 20 the truncation would misprice silently (the reference's np.random.poisson has no limit), so such
    rates are refused with the remedy in the message.  At the edge (lambda dt = 20) the checker still agrees."""
    with pytest.raises(ol.AccelerationError, match="raise n_steps"):
        _hip.jump_diffusion(*P, 0.0, True, False, 50.0, -0.1, 0.2, 0.0, 1000, 2, 1)            # lambda dt = 25
    got = _hip.jump_diffusion(*P, 0.0, True, False, 40.0, -0.1, 0.05, 0.0, 2000, 2, 1)         # lambda dt = 20
    sx, sxx, n = po.jump_moments(*P, 0.0, True, False, 40.0, -0.1, 0.05, 0.0, 2000, 2, 1)
    assert got.n == n and got.sum == pytest.approx(sx, rel=1e-5)


# ------------------------------------------------------------------ fused Asian Greeks (round 4)
@pytest.mark.parametrize("second", [False, True])
@pytest.mark.parametrize("anti", [False, True])
@pytest.mark.parametrize("geometric", [False, True])
def test_fused_asian_greeks_equal_the_literal_bump_and_reprice(second, anti, geometric):
    """olmc_asian_greeks_fd: the 8 / 14 bumped contracts of compute_greeks_unified over ExoticAdapter(AsianOption)
    (unified_greeks.py:177-227, 295-358) as at most six path recursions in ONE launch, against the 8 / 14 launches of the default
    (fp64-exponent) arithmetic / the geometric Asian kernel on the same normals: every evaluation's own sums to 1e-13, every Greek to 1e-8."""
    S, K, T, r, v, q = 100.0, 95.0, 0.75, 0.04, 0.25, 0.01
    N, M, seed = 150_001, 67, 9
    vals, evals = _hip.asian_greeks_fd(S, K, T, r, v, q, True, N, M, seed, anti, second, geometric=geometric)
    h_S, h_v, h_r, h_T = max(1e-4, 0.01 * S), 0.01, 1e-4, 1 / 365.0
    bumps = [(S, T, r, v), (S + h_S, T, r, v), (S - h_S, T, r, v), (S, T, r, v + h_v), (S, T, r, v - h_v), (S, T - h_T, r, v), (S, T, r + h_r, v), (S, T, r - h_r, v)]
    if second:
        bumps += [(S + h_S, T, r, v + h_v), (S + h_S, T, r, v - h_v), (S - h_S, T, r, v + h_v), (S - h_S, T, r, v - h_v), (S + h_S, T - h_T, r, v), (S - h_S, T - h_T, r, v)]
    for (S_, T_, r_, v_), got in zip(bumps, evals):
        one = _hip.asian(S_, K, T_, r_, v_, q, True, geometric, N, M, seed, anti)
        assert got.n == one.n and got.sum == pytest.approx(one.sum, rel=1e-13) and got.sumsq == pytest.approx(one.sumsq, rel=1e-13)
        assert got.price == pytest.approx(one.price, rel=1e-13)
    ad = ol.ExoticAdapter(ol.AsianOption(S, K, T, r, v, q, seed=seed), n_paths=N, n_steps=M, antithetic=anti, avg_type="geometric" if geometric else "arithmetic")
    fused = ol.compute_greeks_unified(ad, S, K, T, r, v, "call", q, include_second_order=second)
    assert (ad.exotic.S, ad.exotic.T, ad.exotic.sigma) == (S, T, v)          # the fused form leaves the option at the base point
    literal = ol.compute_greeks_unified(ad, S, K, T, r, v, "call", q, include_second_order=second, fused=False)
    assert list(fused) == list(literal) and list(fused.values())[:len(vals[:9 if second else 6])] == [np.float64(x) for x in vals[:9 if second else 6]]
    for k in fused:
        assert fused[k] == pytest.approx(literal[k], rel=1e-8, abs=1e-8), k
    assert isinstance(fused["delta"], np.float64)


def test_geometric_asian_price_and_fused_greeks_against_the_discrete_closed_form():
    """An anchor outside this repository's own kernels: the geometric average over the M dates t_i = i T / M of a lognormal path is
    lognormal -- ln G ~ N(ln S + (r - q - sigma^2 / 2) T (M + 1) / (2 M), sigma^2 T (M + 1)(2 M + 1) / (6 M^2)) -- so the discretely
    monitored option has an exact Black-Scholes-type price, and compute_greeks_unified's own bump formulas (unified_greeks.py:295-358)
    applied to THAT function give the exact finite-difference Greeks.  Eight seeds of 2M paths x 64 dates each through the one-launch
    kernel: every Greek's mean over the seeds within 4.5 standard errors (of that mean) of the exact value."""
    S, K, T, r, v, q, M, N = 100.0, 100.0, 1.0, 0.05, 0.2, 0.01, 64, 1 << 21
    cdf = lambda x: 0.5 * math.erfc(-x / math.sqrt(2.0))

    class Exact:
        def price(self, S, K, T, r, sigma, option_type, q=0.0, **kw):
            mu = math.log(S) + (r - q - 0.5 * sigma * sigma) * T * (M + 1) / (2.0 * M)
            var = sigma * sigma * T * (M + 1) * (2 * M + 1) / (6.0 * M * M)
            sd = math.sqrt(var)
            d1 = (mu + var - math.log(K)) / sd
            fwd = math.exp(mu + 0.5 * var)
            call = math.exp(-r * T) * (fwd * cdf(d1) - K * cdf(d1 - sd))
            return call if option_type == "call" else call - math.exp(-r * T) * (fwd - K)

    exact = ol.compute_greeks_unified(Exact(), S, K, T, r, v, "call", q, include_second_order=True)
    runs = []
    for seed in range(8):
        ad = ol.ExoticAdapter(ol.AsianOption(S, K, T, r, v, q, seed=100 + seed), n_paths=N, n_steps=M, avg_type="geometric")
        runs.append(ol.compute_greeks_unified(ad, S, K, T, r, v, "call", q, include_second_order=True))
    for k, want in exact.items():
        xs = np.array([float(g[k]) for g in runs])
        se = xs.std(ddof=1) / math.sqrt(len(xs))
        assert abs(xs.mean() - want) <= 4.5 * se + 1e-12, (k, xs.mean(), want, se)
    assert exact["price"] == pytest.approx(5.4, abs=0.3) and 0.4 < exact["delta"] < 0.7       # the anchor itself is sane


@pytest.mark.parametrize("second", [False, True])
@pytest.mark.parametrize("kind", ["up-and-out", "up-and-in", "down-and-out", "down-and-in", "floating", "fixed"])
def test_fused_barrier_and_lookback_greeks_equal_the_literal_bump_and_reprice(kind, second):
    """olmc_extrema_greeks_fd (the Greeks streamlit_app/pages/7_Exotic_Options.py:266-284 asks for): the 8 / 14 bumped contracts as six
    recursions of (cumulative log-return, running max, running min) in ONE launch against the 8 / 14 launches of extrema_kernel on the
    same normals -- every evaluation's own sums to 1e-13 (a spot bump moves the barrier's RELATIVE level, not the path), every Greek to
    1e-8, through ExoticAdapter as the page calls it."""
    S, K, T, r, v, q = 100.0, 100.0, 1.0, 0.05, 0.2, 0.01
    N, M, seed, anti = 60_001, 50, 42, kind == "down-and-out"
    barrier = 125.0 if kind.startswith("up") else 82.0
    is_call = kind not in ("down-and-in", "fixed")
    if kind in ("floating", "fixed"):
        payoff, level = (4 if kind == "floating" else 5), 0.0
        one_launch = lambda S_, T_, r_, v_: _hip.lookback(S_, K, T_, r_, v_, q, is_call, kind == "fixed", N, M, seed, anti)
        opt, kw = ol.LookbackOption(S, K, T, r, v, q, seed=seed), dict(lookback_type=kind)
    else:
        payoff, level = _hip.BARRIER_KINDS[kind], barrier
        one_launch = lambda S_, T_, r_, v_: _hip.barrier(S_, K, T_, r_, v_, q, is_call, barrier, payoff, N, M, seed, anti)
        opt, kw = ol.BarrierOption(S, K, T, r, v, q, seed=seed, barrier=barrier), dict(barrier_type=kind)
    vals, evals = _hip.extrema_greeks_fd(S, K, T, r, v, q, is_call, payoff, level, N, M, seed, anti, second)
    h_S, h_v, h_r, h_T = max(1e-4, 0.01 * S), 0.01, 1e-4, 1 / 365.0
    bumps = [(S, T, r, v), (S + h_S, T, r, v), (S - h_S, T, r, v), (S, T, r, v + h_v), (S, T, r, v - h_v), (S, T - h_T, r, v), (S, T, r + h_r, v), (S, T, r - h_r, v)]
    if second:
        bumps += [(S + h_S, T, r, v + h_v), (S + h_S, T, r, v - h_v), (S - h_S, T, r, v + h_v), (S - h_S, T, r, v - h_v), (S + h_S, T - h_T, r, v), (S - h_S, T - h_T, r, v)]
    for bump, got in zip(bumps, evals):
        one = one_launch(*bump)
        assert got.n == one.n and got.sum == pytest.approx(one.sum, rel=1e-13) and got.sumsq == pytest.approx(one.sumsq, rel=1e-13), (kind, bump)
    ad = ol.ExoticAdapter(opt, n_paths=N, n_steps=M, antithetic=anti, **kw)
    typ = "call" if is_call else "put"
    fused = ol.compute_greeks_unified(ad, S, K, T, r, v, typ, q, include_second_order=second)
    assert (opt.S, opt.T, opt.sigma) == (S, T, v)
    literal = ol.compute_greeks_unified(ad, S, K, T, r, v, typ, q, include_second_order=second, fused=False)
    assert list(fused) == list(literal) and [float(x) for x in fused.values()] == vals[:len(fused)]
    for k in fused:
        assert fused[k] == pytest.approx(literal[k], rel=1e-8, abs=1e-8), (kind, k)
    assert fused["price"] > 0 and isinstance(fused["vega"], np.float64)


def test_fused_exotic_greeks_meet_the_reference_run(golden):
    """The fused Asian / barrier / lookback Greeks against the REFERENCE, not only against their own literal launches:
    compute_greeks_unified(ExoticAdapter(option, 20,000 x 64), ...) as the reference computes it (unified_greeks.py:177-358 over
    exotic_options.py:97-131, 163-224, 347-401), each Greek with the standard error of the reference run (from its own per-path
    payoffs under common random numbers, tests/golden/make_golden.py).  The device runs 100 x the paths in ONE launch per case: its
    error is a tenth of the fixture's, the gate 3 sqrt(1 + 1/100) sigma."""
    cases = golden["exotic_adapter_greeks"]
    assert {c["option"] for c in cases} == {"asian", "barrier", "lookback"} and any(c["include_second_order"] for c in cases)
    ratio = 100
    worst = 0.0
    for c in cases:
        S, K, T, r, v, q = 100.0, 100.0, 1.0, 0.05, 0.2, 0.0
        if c["option"] == "asian":
            opt = ol.AsianOption(S, K, T, r, v, seed=42)
        elif c["option"] == "barrier":
            opt = ol.BarrierOption(S, K, T, r, v, seed=42, barrier=c["barrier"])
        else:
            opt = ol.LookbackOption(S, K, T, r, v, seed=42)
        ad = ol.ExoticAdapter(opt, n_paths=ratio * c["n_paths"], n_steps=c["n_steps"], **c["kwargs"])
        assert ad._can_fuse({})                                    # one launch: the kernels under test
        g = ol.compute_greeks_unified(ad, S, K, T, r, v, c["option_type"], q, include_second_order=c["include_second_order"])
        assert list(g) == c["keys"]
        h_S, h_v, h_r, h_T = 1.0, 0.01, 1e-4, 1 / 365.0
        amp = dict(price=1.0, delta=1 / h_S, gamma=4 / h_S**2, vega=1 / h_v, theta=2 / h_T, rho=1 / h_r, vanna=1 / (h_S * h_v), charm=4 / (h_S * h_T), vomma=4 / h_v**2)
        for k in c["keys"]:
            se = c["std_errors"][k] * math.sqrt(1.0 + 1.0 / ratio)
            # a Greek that vanishes path by path (the floating lookback is homogeneous of degree one in S: its gamma is rounding noise
            # in both runs, "standard error" 1e-14) is held to the rounding of the prices it is a difference of, amplified by its divisor
            floor = 1e-11 * abs(c["values"]["price"]) * amp[k]
            z = abs(g[k] - c["values"][k]) / (se + floor / 3.0)
            worst = max(worst, z)
            assert z <= 3.0, (c["option"], c["kwargs"], c["option_type"], k, g[k], c["values"][k], c["std_errors"][k])
    assert 0.5 < worst <= 3.0                                      # the gate is neither vacuous nor crossed


def test_the_exotic_page_greeks_take_one_launch_each():
    """streamlit_app/pages/7_Exotic_Options.py:266-284 as it calls: ExoticAdapter(option, n_paths=10000, n_steps=50, <type kwarg>) and
    compute_greeks_unified(..., include_second_order=False), for the payoffs it offers Greeks of (both averages of the Asian)."""
    S, K, T, r, v = ATM if "ATM" in globals() else (100.0, 100.0, 1.0, 0.05, 0.2)
    _hip.profile_enable(True)
    try:
        for opt, kw in ((ol.AsianOption(S, K, T, r, v, seed=42), dict(avg_type="arithmetic")),
                        (ol.AsianOption(S, K, T, r, v, seed=42), dict(avg_type="geometric")),
                        (ol.BarrierOption(S, K, T, r, v, seed=42, barrier=120.0), dict(barrier_type="up-and-out")),
                        (ol.LookbackOption(S, K, T, r, v, seed=42), dict(lookback_type="floating"))):
            ad = ol.ExoticAdapter(opt, n_paths=10000, n_steps=50, **kw)
            _hip.profile_reset()
            g = ol.compute_greeks_unified(ad, S, K, T, r, v, "call", include_second_order=False)
            launches, _ms = _hip.kernel_time()
            assert launches == 1 and list(g) == ["price", "delta", "gamma", "vega", "theta", "rho"] and g["price"] > 0, (type(opt).__name__, launches)
    finally:
        _hip.profile_enable(False)


def test_the_fused_exotic_greeks_refuse_what_they_cannot_price():
    """olmc_asian_greeks_fd / olmc_extrema_greeks_fd argument checks: status OLMC_ERR_ARG through the binding (AccelerationError),
    a message in olmc_last_error, nothing launched."""
    from optionslab_amd._hip import lib
    import ctypes as C
    out9 = (C.c_double * 9)()
    P = (100.0, 100.0, 1.0, 0.05, 0.2, 0.0)
    assert lib().olmc_asian_greeks_fd(*P, 1, _hip.AVG_ARITHMETIC_FAST, 1000, 10, 1, 0, 0, out9, None) != 0        # the fp32-exponent form has no fused Greeks
    assert b"avg_kind" in lib().olmc_last_error()
    assert lib().olmc_asian_greeks_fd(*P, 1, _hip.AVG_ARITHMETIC, 1000, 10, 1, 0, 0, None, None) != 0                # no landing place
    assert lib().olmc_asian_greeks_fd(100.0, 100.0, 0.0, 0.05, 0.2, 0.0, 1, _hip.AVG_ARITHMETIC, 1000, 10, 1, 0, 0, out9, None) != 0     # T = 0 is the caller's branch
    assert lib().olmc_asian_greeks_fd(*P, 1, _hip.AVG_ARITHMETIC, 0, 10, 1, 0, 0, out9, None) != 0 and lib().olmc_asian_greeks_fd(*P, 1, _hip.AVG_GEOMETRIC, 10, 0, 1, 0, 0, out9, None) != 0
    with pytest.raises(ol.AccelerationError):
        _hip.extrema_greeks_fd(*P, True, 6, 120.0, 1000, 10, 1, False, False)                                       # not a payoff
    with pytest.raises(ol.AccelerationError):
        _hip.extrema_greeks_fd(*P, True, 0, 120.0, (1 << 26) + 1, 10, 1, False, False)                              # beyond one launch of the fused kernel
    with pytest.raises(ol.AccelerationError):
        _hip.asian_greeks_fd(*P, True, (1 << 26) + 1, 10, 1, False, False)
    vals, _ = _hip.asian_greeks_fd(*P, True, 1000, 10, 1, False, False)                                             # and the library is fine afterwards
    assert math.isfinite(vals[0]) and vals[0] > 0


def test_fused_asian_greeks_fall_back_where_there_is_no_fused_kernel():
    """An unseeded option (fresh normals per evaluation, as the reference's), the fp32 form of the arithmetic average and a keyword the
    fused entry point does not know take the literal path; asking for fused=True there is an error, as it is for any pricer without
    a fused form."""
    base = dict(n_paths=5_000, n_steps=12)
    odd = ol.ExoticAdapter(ol.AsianOption(100.0, 100.0, 1.0, 0.05, 0.2, seed=3), avg_type="geometric", precision="fp16", **base)
    assert not odd._can_fuse({})                              # price() rejects it; the literal path is the one that says so
    with pytest.raises(ol.GreeksError):
        ol.compute_greeks_unified(odd, 100.0, 100.0, 1.0, 0.05, 0.2, "call", fused=True)
    unseeded = ol.ExoticAdapter(ol.AsianOption(100.0, 100.0, 1.0, 0.05, 0.2), **base)
    assert not unseeded._can_fuse({}) and unseeded.price(100.0, 100.0, 1.0, 0.05, 0.2, "call") > 0
    fast = ol.ExoticAdapter(ol.AsianOption(100.0, 100.0, 1.0, 0.05, 0.2, seed=3), precision="fp32", **base)
    assert not fast._can_fuse({})
    # the degenerate corner the literal path handles by its own branch: T <= h_T
    short = ol.ExoticAdapter(ol.AsianOption(100.0, 100.0, 0.002, 0.05, 0.2, seed=3), **base)
    a = ol.compute_greeks_unified(short, 100.0, 100.0, 0.002, 0.05, 0.2, "call")
    b = ol.compute_greeks_unified(short, 100.0, 100.0, 0.002, 0.05, 0.2, "call", fused=False)
    assert a["charm"] == 0.0 and all(a[k] == pytest.approx(b[k], rel=1e-8, abs=1e-7) for k in a)
