"""Property-based parity: random sizes, offsets, seeds and contract parameters, device kernels against the
same-stream C checker (hypothesis; derandomized so the driver's run is reproducible).  The parametrized
tests pin chosen shapes; these sweep the ragged ones nobody thought of (path counts around wave / workgroup
boundaries, step counts around the Philox-block and fp32-group boundaries, 64-bit path offsets)."""
import math

import pytest
from hypothesis import HealthCheck, assume, given, settings
from hypothesis import strategies as st

from optionslab_amd import _hip
from optionslab_amd.exceptions import AccelerationError
from oracle import philox_oracle as po

pytestmark = pytest.mark.gpu
REL = 2e-6
import os

# OLMC_PROPERTY_SCALE=30 turns this into a long, randomised hunt (not the default: the driver's run must be reproducible)
SCALE = int(os.environ.get("OLMC_PROPERTY_SCALE", "1"))
COMMON = dict(deadline=None, derandomize=SCALE == 1, suppress_health_check=list(HealthCheck))

paths = st.one_of(st.integers(1, 130), st.sampled_from([255, 256, 257, 511, 513, 1023, 1025]), st.integers(131, 3000))
steps = st.one_of(st.integers(1, 40), st.sampled_from([63, 64, 65, 127, 129, 252]))
seeds = st.integers(0, 2**64 - 1)
offsets = st.one_of(st.just(0), st.integers(0, 10_000), st.sampled_from([2**32 - 3, 2**32, 2**40 + 17]))
spot = st.floats(50.0, 150.0)
strike = st.floats(60.0, 140.0)
vol = st.floats(0.05, 0.6)
rate = st.floats(0.0, 0.08)
div = st.floats(0.0, 0.05)
mat = st.floats(0.1, 2.0)
if os.environ.get("OLMC_PROPERTY_WIDE"):       # a hunt far outside what a desk would type: penny stocks to indices, 1 % to 150 % vol, negative rates, days to 8 years
    spot, strike = st.floats(0.5, 5000.0), st.floats(0.5, 5000.0)
    vol, rate, div, mat = st.floats(0.01, 1.5), st.floats(-0.03, 0.15), st.floats(0.0, 0.1), st.floats(0.02, 8.0)


def close(a, b, scale=1.0, n=0, level=0.0, power=1):
    """Relative 2e-6 of the sum -- or, when the sum is small because most payoffs sit near / below their kink, the
    absolute error that n prices of size `level`, each good to 2e-6 relative (hardware log2 / sin / cos), can leave."""
    return a == pytest.approx(b, rel=REL * scale, abs=1e-9 * scale + REL * scale * n * (3.0 * level) ** power)


@settings(max_examples=60 * SCALE, **COMMON)
@given(N=paths, M=steps, seed=seeds, off=offsets, S=spot, K=strike, v=vol, r=rate, q=div, T=mat, call=st.booleans(), anti=st.booleans())
def test_european_shard(N, M, seed, off, S, K, v, r, q, T, call, anti):
    got = _hip.european(S, K, T, r, v, q, call, N, M, seed, anti, path_offset=off)
    sx, sxx, *_rest, n = po.european_moments(S, K, T, r, v, q, call, N, M, seed, anti, off)
    assert got.n == n and close(got.sum, sx, 1, n, max(S, K)) and close(got.sumsq, sxx, 4, n, max(S, K), 2)


@settings(max_examples=40 * SCALE, **COMMON)
@given(N=paths, M=steps, seed=seeds, off=offsets, S=spot, K=strike, v=vol, r=rate, q=div, T=mat, call=st.booleans(), anti=st.booleans(),
       geo=st.booleans(), fast=st.booleans())
def test_asian(N, M, seed, off, S, K, v, r, q, T, call, anti, geo, fast):
    got = _hip.asian(S, K, T, r, v, q, call, geo, N, M, seed, anti, path_offset=off, fast=fast and not geo)
    sx, sxx, n = po.asian_moments(S, K, T, r, v, q, call, geo, N, M, seed, anti, off)
    assert got.n == n and close(got.sum, sx, 1, n, max(S, K)) and close(got.sumsq, sxx, 4, n, max(S, K), 2)


@settings(max_examples=40 * SCALE, **COMMON)
@given(N=paths, M=steps, seed=seeds, S=spot, K=strike, v=vol, r=rate, q=div, T=mat, call=st.booleans(), anti=st.booleans(),
       payoff=st.integers(4, 5))
def test_lookback(N, M, seed, S, K, v, r, q, T, call, anti, payoff):
    # (barriers: test_barrier below)
    got = _hip.lookback(S, K, T, r, v, q, call, payoff == 5, N, M, seed, anti)
    sx, sxx, n = po.extrema_moments(S, K, T, r, v, q, call, payoff, 0.0, N, M, seed, anti)
    assert got.n == n and close(got.sum, sx, 1, n, max(S, K)) and close(got.sumsq, sxx, 4, n, max(S, K), 2)


@settings(max_examples=40 * SCALE, **COMMON)
@given(N=paths, M=st.integers(1, 80), seed=seeds, S=spot, v=vol, r=rate, q=div, T=mat, anti=st.booleans(), periods=st.integers(1, 12),
       cap=st.floats(0.01, 0.2), floor=st.floats(-0.2, 0.0))
def test_cliquet(N, M, seed, S, v, r, q, T, anti, periods, cap, floor):
    if M // periods < 1:
        periods = M
    got = _hip.cliquet(S, T, r, v, q, cap, floor, 0.5, 0.0, periods, N, M, seed, anti)
    sx, sxx, n = po.cliquet_moments(S, T, r, v, q, cap, floor, 0.5, 0.0, periods, N, M, seed, anti)
    assert got.n == n and close(got.sum, sx, 1, n, S) and close(got.sumsq, sxx, 4, n, S, 2)


@settings(max_examples=40 * SCALE, **COMMON)
@given(N=paths, M=st.integers(1, 60), seed=seeds, S=spot, K=strike, v=vol, r=rate, q=div, T=mat, call=st.booleans(), kou=st.booleans(),
       lam=st.floats(0.0, 30.0))
def test_jump_diffusion(N, M, seed, S, K, v, r, q, T, call, kou, lam):
    a = (0.4, 10.0, 5.0) if kou else (-0.1, 0.2, 0.0)
    if lam * (T / M) > 20.0:        # more than 20 expected jumps per step: refused (the inversion sampler caps at 64 jumps), not mispriced
        with pytest.raises(AccelerationError, match="jumps per step"):
            _hip.jump_diffusion(S, K, T, r, v, q, call, kou, lam, *a, N, M, seed)
        return
    got = _hip.jump_diffusion(S, K, T, r, v, q, call, kou, lam, *a, N, M, seed)
    sx, sxx, n = po.jump_moments(S, K, T, r, v, q, call, kou, lam, *a, N, M, seed)
    assert got.n == n and close(got.sum, sx, 2, n, max(S, K)) and close(got.sumsq, sxx, 8, n, max(S, K), 2)


@settings(max_examples=40 * SCALE, **COMMON)
@given(N=paths, M=st.integers(1, 60), seed=seeds, S=spot, K=strike, r=rate, q=div, T=mat, call=st.booleans(), anti=st.booleans(),
       kappa=st.floats(0.5, 4.0), theta=st.floats(0.01, 0.1), sv=st.floats(0.05, 1.0), rho=st.floats(-0.95, 0.95), v0=st.floats(0.005, 0.15))
def test_heston(N, M, seed, S, K, r, q, T, call, anti, kappa, theta, sv, rho, v0):
    got = _hip.heston(S, K, T, r, q, call, kappa, theta, sv, rho, v0, N, M, seed, anti)
    sx, sxx, n = po.heston_moments(S, K, T, r, q, call, kappa, theta, sv, rho, v0, N, M, seed, anti)
    assert got.n == n and close(got.sum, sx, 2, n, max(S, K)) and close(got.sumsq, sxx, 8, n, max(S, K), 2)
    assert math.isfinite(got.price)


@settings(max_examples=20 * SCALE, **COMMON)
@given(N=st.integers(1, 700), M=st.integers(1, 48), seed=st.integers(0, 2**31 - 1), S=spot, v=vol, r=rate, q=div, T=mat,
       off=st.integers(0, 300))
def test_qmc_terminal_equals_the_scipy_based_oracle(N, M, seed, S, v, r, q, T, off):
    """Scrambled Sobol points expanded on the device from SciPy's own direction numbers: the terminal prices equal the
    NumPy/SciPy restatement of simulate_gbm_qmc (gbm_qmc.py:14-46) to rounding, at any point offset."""
    import warnings

    import numpy as np

    import optionslab_amd as ol
    from oracle import numpy_reference as orc
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")            # SciPy: n is not a power of two
        want = orc.terminal_sobol(S, T, r, v, q, N + off, M, seed)[off:]
    got = _hip.european_qmc_terminal(S, T, r, v, q, N, *ol.monte_carlo.sobol_tables(M, seed), point_offset=off)
    assert got.shape == want.shape and np.allclose(got, want, rtol=1e-11, atol=0)


# ---- sets of contracts on common normals (the fused Greeks machinery) ----------------------------------------------------------
# A set is laid out by the host (group_contracts): contracts with bit-identical sigma * sqrt(dt) form a group behind one base (a
# pair of exps), the others take scale * S_T(base); slot NSETS / 2 is always a base because the kernel walks the two halves as two
# streams.  Random group structures probe every such layout: groups that straddle the middle, sets of 2..16 contracts (8- and
# 16-slot kernels with padding), with and without antithetic legs, ragged path counts (dead lanes carry a NaN normal sum), split
# workgroups (M >= 64), many workgroups (wide rows summed by the whole workgroup, one and two reduction levels).
contracts = st.lists(st.tuples(st.integers(0, 5), st.floats(60.0, 140.0), st.floats(-5.0, 5.0), st.floats(-0.02, 0.02), st.booleans()),
                     min_size=2, max_size=16)
batch_paths = st.one_of(st.integers(1, 600), st.sampled_from([255, 256, 257, 70_000, 131_073]))
batch_steps = st.one_of(st.integers(1, 20), st.sampled_from([63, 64, 65, 130]))


@settings(max_examples=40 * SCALE, **COMMON)
@given(cs=contracts, N=batch_paths, M=batch_steps, seed=seeds, off=offsets, S=spot, r=rate, q=div, T=mat, anti=st.booleans())
def test_european_batch_prices_every_contract_as_its_own_launch_does(cs, N, M, seed, off, S, r, q, T, anti):
    """olmc_european_batch (one launch, one set of normals) against one olmc_european_shard launch per contract on the same
    stream.  A base contract evaluates the very same expressions as the single launch; a contract that shares its vol with a
    base takes scale * S_T(base) with scale = exp(a - a_base) -- a few ulp of S_T, i.e. ~1e-15 relative on a sum of prices;
    the reduction orders differ (1e-16 per addition).  Tolerance 1e-11 relative on sum and sum of squares."""
    vols = [0.1 + 0.07 * g for g in range(6)]
    assume(all(S + dS > 0.1 for (_g, _K, dS, _dr, _c) in cs))                                 # (the wide hunt's spots start at 0.5)
    opts = [(S + dS, K, T, r + dr, vols[g], q, call) for (g, K, dS, dr, call) in cs]        # same g => same vol bits: S- and r-bumps of one another
    got = _hip.european_batch(opts, N, M, seed, anti, path_offset=off)
    for o, g in zip(opts, got):
        one = _hip.european(*o[:6], o[6], N, M, seed, anti, path_offset=off)
        assert g.n == one.n
        assert g.sum == pytest.approx(one.sum, rel=1e-11, abs=1e-9) and g.sumsq == pytest.approx(one.sumsq, rel=1e-11, abs=1e-7)
        assert g.price == pytest.approx(one.price, rel=1e-11, abs=1e-12)


# ---------------------------------------------------------------------------------------------------- round 4
@settings(max_examples=25 * SCALE, **COMMON)
@given(N=st.integers(16, 200_000), M=steps, seed=seeds, ranks=st.integers(1, 12), S=spot, K=strike, v=vol, r=rate, q=div, T=mat, call=st.booleans(),
       anti=st.booleans())
def test_multi_rank_engine_equals_the_rank_ordered_sum_of_its_shards(N, M, seed, ranks, S, K, v, r, q, T, call, anti):
    """olmc_multi_gpu_european with 1..12 ranks REHEARSED on this one device (instrumented build): the triple is bit for bit what
    olmc_combine_stats makes of the shards [d N / P, (d + 1) N / P) priced one by one, for any path count, rank count and contract."""
    from tools.probe import binding as probe
    hip = probe.hip
    ranks = min(ranks, N)
    probe.tune(probe.TUNE_MULTI_REHEARSAL, 1)
    try:
        got = hip.multi_gpu_european(S, K, T, r, v, q, call, N, M, seed, anti, ranks)
    finally:
        probe.tune(probe.TUNE_MULTI_REHEARSAL, 0)
    parts = []
    for d in range(ranks):
        lo, hi = N * d // ranks, N * (d + 1) // ranks
        st_ = hip.european(S, K, T, r, v, q, call, hi - lo, M, seed, anti, path_offset=lo)
        parts.append((st_.sum, st_.sumsq, st_.n))
    want = hip.combine_stats(parts, r, T)
    assert (got.sum, got.sumsq, got.n, got.price, got.std_error) == (want.sum, want.sumsq, want.n, want.price, want.std_error)
    assert hip.device_info()["device"] == 0


@settings(max_examples=25 * SCALE, **COMMON)
@given(N=st.one_of(st.integers(1, 300), st.sampled_from([63, 64, 65, 4095, 4097]), st.integers(301, 20_000)), M=st.integers(1, 80), seed=st.integers(0, 2**31 - 1),
       off=st.one_of(st.just(0), st.integers(0, 100_000)), S=spot, v=vol, r=rate, q=div, T=mat)
def test_sobol_launch_shapes_return_the_same_terminal_prices(N, M, seed, off, S, v, r, q, T):
    """One point per thread, eight points per thread and split workgroups (a quarter of the dimensions per wave): the same Sobol points,
    the same canonical association of a point's normal sum -- terminal prices bit for bit, for any point count, dimension count and offset."""
    import numpy as np
    from optionslab_amd.monte_carlo import sobol_tables
    tables = sobol_tables(M, seed)
    try:
        out = []
        for knob in (-1, 0, 1):
            _hip.tune(_hip.TUNE_QMC_BLOCK, knob)
            out.append(_hip.european_qmc_terminal(S, T, r, v, q, N, *tables, point_offset=off))
    finally:
        _hip.tune(_hip.TUNE_QMC_BLOCK, 0)
    assert np.isfinite(out[0]).all() and np.array_equal(out[0], out[1]) and np.array_equal(out[0], out[2])


def _bumps(S, T, r, v, second):
    """unified_greeks.py:295-358: the 8 / 14 evaluation points in call order."""
    h_S, h_v, h_r, h_T = max(1e-4, 0.01 * S), 0.01, 1e-4, 1 / 365.0
    b = [(S, T, r, v), (S + h_S, T, r, v), (S - h_S, T, r, v), (S, T, r, v + h_v), (S, T, r, v - h_v), (S, T - h_T, r, v), (S, T, r + h_r, v), (S, T, r - h_r, v)]
    if second:
        b += [(S + h_S, T, r, v + h_v), (S + h_S, T, r, v - h_v), (S - h_S, T, r, v + h_v), (S - h_S, T, r, v - h_v), (S + h_S, T - h_T, r, v), (S - h_S, T - h_T, r, v)]
    return b


@settings(max_examples=30 * SCALE, **COMMON)
@given(N=paths, M=st.one_of(steps, st.sampled_from([255, 256, 257, 300])), seed=seeds, S=spot, K=strike, v=vol, r=rate, q=div, T=mat, call=st.booleans(),
       anti=st.booleans(), second=st.booleans(), geo=st.booleans())
def test_fused_asian_greeks_are_their_own_launches(N, M, seed, S, K, v, r, q, T, call, anti, second, geo):
    """olmc_asian_greeks_fd, any shape and contract: every one of the 8 / 14 evaluations is what its own launch of the one-contract kernel
    returns -- the bits for the recursions of their own, 2e-14 for the two r bumps of the arithmetic average, which ride on the mid
    recursion through a per-date growth term re-anchored every 256 dates (the date counts around 256 are there for that; the hunt
    took two less careful forms of that term apart -- see the kernel)."""
    _vals, evals = _hip.asian_greeks_fd(S, K, T, r, v, q, call, N, M, seed, anti, second, geometric=geo)
    for k, ((S_, T_, r_, v_), got) in enumerate(zip(_bumps(S, T, r, v, second), evals)):
        one = _hip.asian(S_, K, T_, r_, v_, q, call, geo, N, M, seed, anti)
        assert got.n == one.n
        if geo or k not in (6, 7):
            assert got.sum == pytest.approx(one.sum, rel=1e-14, abs=1e-300) and got.sumsq == pytest.approx(one.sumsq, rel=1e-14, abs=1e-300), k
        else:
            # the literal launch of an r bump and the mid recursion round their cumulative log-returns independently at every date (other
            # drift, other roundings: a random walk of a few ulp over some hundred dates, seen: 6 ulp on one path of 256 dates), and
            # K - avg keeps that absolute error.  An off-by-one date in the growth term would be 1e-7 of a payoff: eight orders above this.
            lvl = max(S, K)
            assert got.sum == pytest.approx(one.sum, rel=2e-14, abs=2e-14 * got.n * lvl) and got.sumsq == pytest.approx(one.sumsq, rel=4e-14, abs=2e-14 * got.n * lvl * lvl), k


@settings(max_examples=30 * SCALE, **COMMON)
@given(N=paths, M=steps, seed=seeds, S=spot, K=strike, v=vol, r=rate, q=div, T=mat, call=st.booleans(), anti=st.booleans(), second=st.booleans(),
       payoff=st.integers(0, 5), rel_level=st.floats(0.03, 0.4))
def test_fused_barrier_and_lookback_greeks_are_their_own_launches(N, M, seed, S, K, v, r, q, T, call, anti, second, payoff, rel_level):
    """olmc_extrema_greeks_fd, any shape, contract and payoff (four barriers, two lookbacks): each evaluation against its own launch."""
    level = S * (1 + rel_level) if payoff < 2 else S * (1 - rel_level)          # up barriers above the spot, down barriers below
    _vals, evals = _hip.extrema_greeks_fd(S, K, T, r, v, q, call, payoff, level if payoff < 4 else 0.0, N, M, seed, anti, second)
    for (S_, T_, r_, v_), got in zip(_bumps(S, T, r, v, second), evals):
        if payoff < 4:
            one = _hip.barrier(S_, K, T_, r_, v_, q, call, level, payoff, N, M, seed, anti)
        else:
            one = _hip.lookback(S_, K, T_, r_, v_, q, call, payoff == 5, N, M, seed, anti)
        assert got.n == one.n and got.sum == pytest.approx(one.sum, rel=1e-14, abs=1e-300) and got.sumsq == pytest.approx(one.sumsq, rel=1e-14, abs=1e-300)


@settings(max_examples=25 * SCALE, **COMMON)
@given(N=st.one_of(st.integers(8, 600), st.sampled_from([255, 256, 257, 1024, 4097]), st.integers(601, 30_000)), M=st.integers(1, 24), seed=st.integers(0, 2**40),
       S=st.floats(80.0, 120.0), K=st.floats(85.0, 115.0), v=st.floats(0.1, 0.5), r=rate, q=div, T=st.floats(0.25, 2.0), call=st.booleans(), deg=st.integers(1, 4))
def test_american_chain(N, M, seed, S, K, v, r, q, T, call, deg):
    """The Longstaff-Schwartz chain (one launch per exercise date, every workgroup of a launch summing the rows of the one before and
    solving the normal equations itself) against the checker's serial restatement of the same algebra on the same paths: same exercise
    decisions path by path, sums to the 2e-6 the hardware log2 / sin / cos leave on a path (a flipped decision would move a sum by one
    whole cash flow)."""
    got = _hip.american_lsm(S, K, T, r, v, q, call, N, M, deg, seed)
    sx, sxx, n = po.american_lsm(S, K, T, r, v, q, call, N, M, deg, seed)
    if SCALE > 1 and not close(got.sum, sx, 1, n, max(S, K)):
        # the randomised hunt only: the device's normals (hardware log2 / sin / cos) and the checker's (libm) differ by ~1e-7, so about
        # one pricing in several thousand has a path whose exercise value sits within 1e-6 of its continuation value and is decided
        # differently -- ONE cash flow moves (seen: 1 in 5,000 examples); more than two is a failure
        assert abs(got.sum - sx) <= 2.0 * max(S, K) * math.exp(4 * v * math.sqrt(T)), (got.sum, sx)
        return
    assert got.n == n and close(got.sum, sx, 1, n, max(S, K)) and close(got.sumsq, sxx, 4, n, max(S, K), 2)


@settings(max_examples=40 * SCALE, **COMMON)
@given(N=st.one_of(paths, st.integers(3001, 70_000)), M=steps, seed=seeds, off=offsets, S=spot, K=strike, v=vol, r=rate, q=div, T=mat, call=st.booleans(), anti=st.booleans())
def test_control_variate_moments(N, M, seed, off, S, K, v, r, q, T, call, anti):
    """olmc_european_cv_shard: the five moments of (discounted payoff, terminal price) the control-variate estimator is built from
    (monte_carlo.py:154-186), against the checker's on the same paths."""
    got = _hip.european_cv_shard(S, K, T, r, v, q, call, off, N, M, seed, anti)
    sx, sxx, ss, sss, sxs, n = po.european_moments(S, K, T, r, v, q, call, N, M, seed, anti, off)
    disc = math.exp(-r * T)
    level = max(S, K) * math.exp(3 * v * math.sqrt(T))
    assert got.n == n and close(got.sum_d, disc * sx, 1, n, level) and close(got.sum_dd, disc * disc * sxx, 4, n, level, 2)
    assert close(got.sum_s, ss, 1, n, level) and close(got.sum_ss, sss, 4, n, level, 2) and close(got.sum_ds, disc * sxs, 4, n, level, 2)


@settings(max_examples=40 * SCALE, **COMMON)
@given(N=st.one_of(paths, st.integers(3001, 300_000)), M=steps, seed=seeds, S=spot, K=strike, v=vol, r=rate, q=div, T=mat, call=st.booleans(), second=st.booleans())
def test_fused_european_greeks_are_their_own_launches(N, M, seed, S, K, v, r, q, T, call, second):
    """olmc_european_greeks_fd at any shape (whole workgroups, split workgroups, the ragged end): each of the 8 / 14 evaluations against
    the one-contract launch on the same normals, to the rounding the two payoff forms differ by (olmc.h, olmc_european_batch)."""
    _vals, evals = _hip.european_greeks_fd(S, K, T, r, v, q, call, N, M, seed, second)
    for (S_, T_, r_, v_), got in zip(_bumps(S, T, r, v, second), evals):
        one = _hip.european(S_, K, T_, r_, v_, q, call, N, M, seed, True)
        assert got.n == one.n and got.sum == pytest.approx(one.sum, rel=1e-12, abs=1e-9) and got.sumsq == pytest.approx(one.sumsq, rel=1e-12, abs=1e-9)


@settings(max_examples=20 * SCALE, **COMMON)
@given(N=st.one_of(st.integers(2, 300), st.sampled_from([1023, 1024, 1025, 4097]), st.integers(301, 20_000)), M=st.integers(1, 30), seed=seeds, S=spot, K=strike,
       v=vol, r=rate, q=div, T=mat, call=st.booleans())
def test_exercise_boundary_is_the_percentile_of_the_in_the_money_prices(N, M, seed, S, K, v, r, q, T, call):
    """olmc_exercise_boundary (exotic_options.py:309-345): per date the 10th (put) / 90th (call) percentile -- NumPy's linear
    interpolation -- of the in-the-money prices, NaN where nobody is in the money; against np.percentile on the device's own path matrix."""
    import numpy as np
    from oracle import numpy_reference as nr
    got = _hip.exercise_boundary(S, K, T, r, v, q, call, N, M, seed)
    paths_ = _hip.gbm_paths(math.exp(math.log(S)), T, r, v, q, N, M, seed, path_major=True)
    want = nr.exercise_boundary_from_paths(paths_, K, "call" if call else "put")
    assert got.shape == want.shape and np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want)
    assert np.allclose(got[ok], want[ok], rtol=1e-12, atol=0.0)


@settings(max_examples=25 * SCALE, **COMMON)
@given(N=st.one_of(st.integers(1, 300), st.sampled_from([63, 64, 65, 4095, 4097, 1 << 15]), st.integers(301, 40_000)), M=st.integers(1, 70), seed=st.integers(0, 2**31 - 1),
       S=spot, K=strike, v=vol, r=rate, q=div, T=mat, call=st.booleans(), second=st.booleans())
def test_fused_sobol_greeks_and_control_variate_are_their_own_launches(N, M, seed, S, K, v, r, q, T, call, second):
    """olmc_european_qmc_greeks_fd (one launch, 8 / 14 contracts on the same Sobol points; eight points per thread or split workgroups
    by size) against olmc_european_qmc per contract, and olmc_european_qmc_cv's payoff moments against the same launch."""
    from optionslab_amd.monte_carlo import sobol_tables
    sv, shift = sobol_tables(M, seed)
    _vals, evals = _hip.european_qmc_greeks_fd(S, K, T, r, v, q, call, N, sv, shift, second)
    for (S_, T_, r_, v_), got in zip(_bumps(S, T, r, v, second), evals):
        one = _hip.european_qmc(S_, K, T_, r_, v_, q, call, N, sv, shift)
        assert got.n == one.n and got.sum == pytest.approx(one.sum, rel=1e-12, abs=1e-9) and got.sumsq == pytest.approx(one.sumsq, rel=1e-12, abs=1e-9)
    base = _hip.european_qmc(S, K, T, r, v, q, call, N, sv, shift)
    cv = _hip.european_qmc_cv(S, K, T, r, v, q, call, N, sv, shift)
    disc = math.exp(-r * T)
    assert cv.n == base.n and cv.sum_d == pytest.approx(disc * base.sum, rel=1e-12, abs=1e-9) and cv.sum_dd == pytest.approx(disc * disc * base.sumsq, rel=1e-12, abs=1e-9)


@settings(max_examples=20 * SCALE, **COMMON)
@given(N=st.integers(16, 100_000), M=steps, seed=seeds, ranks=st.integers(1, 9), S=spot, K=strike, v=vol, r=rate, q=div, T=mat, call=st.booleans(), second=st.booleans())
def test_multi_rank_greeks_and_control_variate_equal_the_one_device_call(N, M, seed, ranks, S, K, v, r, q, T, call, second):
    """olmc_multi_gpu_greeks_fd / _european_cv with 1..9 ranks REHEARSED on this one device (instrumented build): every rank prices its
    contiguous range of the global paths, ONE grouped all-reduce of 2 k + 1 / 6 sums; against the one-device call to the rounding of
    another summation order."""
    from tools.probe import binding as probe
    hip = probe.hip
    ranks = min(ranks, N)
    probe.tune(probe.TUNE_MULTI_REHEARSAL, 1)
    try:
        vals, evals = hip.multi_gpu_greeks_fd(S, K, T, r, v, q, call, N, M, seed, second, ranks)
        cv = hip.multi_gpu_european_cv(S, K, T, r, v, q, call, N, M, seed, True, ranks)
    finally:
        probe.tune(probe.TUNE_MULTI_REHEARSAL, 0)
    vals1, evals1 = hip.european_greeks_fd(S, K, T, r, v, q, call, N, M, seed, second)
    for a, b in zip(evals, evals1):
        assert a.n == b.n and a.sum == pytest.approx(b.sum, rel=1e-12, abs=1e-9) and a.sumsq == pytest.approx(b.sumsq, rel=1e-12, abs=1e-9)
    cv1 = hip.european_cv(S, K, T, r, v, q, call, N, M, seed, True)
    for f in ("sum_d", "sum_s", "sum_dd", "sum_ss", "sum_ds"):
        assert getattr(cv, f) == pytest.approx(getattr(cv1, f), rel=1e-12, abs=1e-9), f
    assert cv.n == cv1.n and hip.device_info()["device"] == 0


@settings(max_examples=20 * SCALE, **COMMON)
@given(N=st.one_of(st.integers(1, 300), st.sampled_from([255, 256, 257, 1025])), M=st.integers(1, 40), seed=seeds, S=spot, v=vol, r=rate, q=div, T=mat,
       kappa=st.floats(0.5, 4.0), theta=st.floats(0.01, 0.1), sv=st.floats(0.05, 1.0), rho=st.floats(-0.95, 0.95), v0=st.floats(0.005, 0.15), path_major=st.booleans())
def test_path_matrices(N, M, seed, S, v, r, q, T, kappa, theta, sv, rho, v0, path_major):
    """simulate_gbm_paths (gbm_numpy.py:86-118) and HestonPricer.simulate_paths (heston.py:257-305) on the device: every state of every
    path, both layouts, against the checker's."""
    import numpy as np
    got = _hip.gbm_paths(S, T, r, v, q, N, M, seed, path_major=path_major)
    want = po.gbm_paths(S, T, r, v, q, N, M, seed)
    assert np.allclose(got.T if path_major else got, want, rtol=4e-6, atol=0)
    spot_, var_ = _hip.heston_paths(S, T, r, q, kappa, theta, sv, rho, v0, N, M, seed, path_major=path_major)
    ws, wv = po.heston_paths(S, T, r, q, kappa, theta, sv, rho, v0, N, M, seed)
    gs_, gv_ = (spot_.T, var_.T) if path_major else (spot_, var_)
    # state by state only on paths whose variance stays away from the truncation at 0: there sqrt(v) turns the 1e-7 of the hardware
    # normals into 3e-4 and the two paths part for good (a Hoelder-1/2 map; the PRICE property above averages over it)
    calm = wv.min(axis=0) > 2e-3
    assert np.allclose(gs_[:, calm], ws[:, calm], rtol=2e-5, atol=0) and np.allclose(gv_[:, calm], wv[:, calm], rtol=2e-4, atol=1e-7)
    assert np.isfinite(gs_).all() and (gs_ > 0).all() and (gv_ >= 0).all() and np.array_equal(gv_[0], wv[0]) and np.array_equal(gs_[0], ws[0])


@settings(max_examples=25 * SCALE, **COMMON)
@given(cs=st.lists(st.tuples(spot, strike, mat, rate, vol, div), min_size=1, max_size=40), N=st.one_of(st.integers(1, 600), st.sampled_from([255, 256, 257, 4097, 70_000])),
       M=batch_steps, seed=seeds, call=st.booleans(), anti=st.booleans())
def test_a_batch_of_independent_contracts_prices_each_as_its_own_launch(cs, N, M, seed, call, anti):
    """olmc_european_multi (MonteCarloPricerUni.price_batch, monte_carlo_unified.py:562-631: one launch, a grid of contracts x path
    blocks, per-contract finisher): with every contract on stream tag 0 each one is exactly the single-contract launch of the same
    contract -- any number of contracts, any ragged path count, with and without antithetic legs."""
    import numpy as np
    a = np.array(cs, dtype=np.float64)
    got = _hip.european_multi(a[:, 0], a[:, 1], a[:, 2], a[:, 3], a[:, 4], a[:, 5], call, N, M, seed, anti, tags=np.zeros(len(cs), dtype=np.uint32))
    for j, (S_, K_, T_, r_, v_, q_) in enumerate(cs):
        one = _hip.european(S_, K_, T_, r_, v_, q_, call, N, M, seed, anti)
        assert got["n"][j] == one.n and got["sum"][j] == pytest.approx(one.sum, rel=1e-12, abs=1e-9) and got["sumsq"][j] == pytest.approx(one.sumsq, rel=1e-12, abs=1e-9), j
        assert got["price"][j] == pytest.approx(one.price, rel=1e-12, abs=1e-12)


def _same_but_for_a_path_on_the_edge(got, want, scale, n, level, power=1, flips=2):
    """close(), or -- in the randomised hunt only -- off by at most `flips` payoffs: a path whose extremum sits within the 1e-7 the
    hardware log2 / sin / cos leave of a barrier lands on one side on the device and on the other in the checker (libm)."""
    if close(got, want, scale, n, level, power):
        return True
    return SCALE > 1 and abs(got - want) <= flips * (3.0 * level) ** power


@settings(max_examples=40 * SCALE, **COMMON)
@given(N=paths, M=steps, seed=seeds, off=offsets, S=spot, K=strike, v=vol, r=rate, q=div, T=mat, call=st.booleans(), anti=st.booleans(), payoff=st.integers(0, 3),
       rel_level=st.floats(0.02, 0.5))
def test_barrier(N, M, seed, off, S, K, v, r, q, T, call, anti, payoff, rel_level):
    """The four barrier payoffs (exotic_options.py:174-224) against the checker on the same paths."""
    level = S * (1 + rel_level) if payoff < 2 else S * (1 - rel_level)
    got = _hip.barrier(S, K, T, r, v, q, call, level, payoff, N, M, seed, anti, path_offset=off)
    sx, sxx, n = po.extrema_moments(S, K, T, r, v, q, call, payoff, level, N, M, seed, anti, off)
    lvl = max(S, K) * math.exp(3 * v * math.sqrt(T))
    assert got.n == n and _same_but_for_a_path_on_the_edge(got.sum, sx, 1, n, lvl) and _same_but_for_a_path_on_the_edge(got.sumsq, sxx, 4, n, lvl, 2)


@settings(max_examples=40 * SCALE, **COMMON)
@given(N=paths, M=st.integers(1, 130), seed=seeds, off=offsets, S=spot, v=vol, r=rate, q=div, T=mat, anti=st.booleans(), freq=st.integers(1, 40),
       autocall=st.floats(0.9, 1.3), coupon_b=st.floats(0.5, 1.0), coupon=st.floats(0.0, 0.2), ki=st.floats(0.4, 0.9))
def test_autocallable(N, M, seed, off, S, v, r, q, T, anti, freq, autocall, coupon_b, coupon, ki):
    """AutocallableOption.price (exotic_options.py:404-491) against the checker: observation dates, early redemption, coupon and
    knock-in barriers on the same paths."""
    if freq > M:
        with pytest.raises(AccelerationError):
            _hip.autocallable(S, T, r, v, q, autocall, coupon_b, coupon, ki, freq, N, M, seed, anti, path_offset=off)
        return
    got = _hip.autocallable(S, T, r, v, q, autocall, coupon_b, coupon, ki, freq, N, M, seed, anti, path_offset=off)
    sx, sxx, n = po.autocall_moments(S, T, r, v, q, autocall, coupon_b, coupon, ki, freq, N, M, seed, anti, off)
    assert got.n == n and _same_but_for_a_path_on_the_edge(got.sum, sx, 1, n, 1.0) and _same_but_for_a_path_on_the_edge(got.sumsq, sxx, 4, n, 1.0, 2)
