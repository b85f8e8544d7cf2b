"""GPU parity tests (run on the MI355X box with `-m gpu`).  Every call goes
through the C ABI of libolmc.so (ctypes).  Three gates:

  1. bit-exact Philox words vs the C checker (integer work);
  2. tight agreement with the C checker, which consumes the same counter stream
     (tolerances below are set by the hardware log2/sin/cos approximations);
  3. statistical agreement with the reference: golden vectors captured from the
     reference itself, 3 sigma of the Monte Carlo standard error (north_star's
     stated floating-point tolerance), plus the reference's own test assertions.
"""
import math
import os
import time

import numpy as np
import pytest

import optionslab_amd as ol
from optionslab_amd import _hip
from oracle import numpy_reference as orc
from oracle import philox_oracle as po

pytestmark = pytest.mark.gpu

ATM = (100.0, 100.0, 1.0, 0.05, 0.2)
BS_CALL = 10.450583572185565
Z_ABS_TOL = 2e-5        # |z_gpu - z_checker| per normal (fp32 normals, hardware transcendentals)
REL_STREAM_TOL = 2e-6   # price vs the same-stream C checker


@pytest.fixture(scope="module", autouse=True)
def _device():
    info = _hip.device_info()
    assert info["arch"].startswith("gfx950"), info
    yield
    _hip.shutdown()


# ------------------------------------------------------------------ RNG stream
def test_philox_words_bit_exact():
    for seed, path0, n_paths, block0, n_blocks, tag in [(42, 0, 300, 0, 5, 0), (2**63 + 12345, (1 << 32) - 7, 20, 1000, 3, 0),
                                                        (0, 0, 1, 0, 1, 0), (7, 5, 9, 62, 2, 3)]:
        got = _hip.philox_words(seed, path0, n_paths, block0, n_blocks, tag)
        assert np.array_equal(got, po.philox_words(seed, path0, n_paths, block0, n_blocks, tag))
    kat = _hip.philox_words(0, 0, 1, 0, 1, 0)[0, 0]
    assert [int(x) for x in kat] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]   # Random123 KAT


@pytest.mark.parametrize("n_steps", [1, 2, 3, 4, 5, 252])
def test_normals_match_checker(n_steps):
    z = _hip.normals(42, 1000, 513, n_steps)
    want = po.normals(42, 1000, 513, n_steps)
    assert z.shape == want.shape and np.isfinite(z).all()
    assert np.abs(z.astype(np.float64) - want.astype(np.float64)).max() < Z_ABS_TOL


def test_normals_distribution():
    from scipy import stats
    z = _hip.normals(123, 0, 200000, 10).astype(np.float64).ravel()
    n = z.size
    assert abs(z.mean()) < 4 / math.sqrt(n)
    assert abs(z.var() - 1) < 4 * math.sqrt(2 / n)
    assert abs(stats.skew(z)) < 4 * math.sqrt(6 / n)
    assert abs(stats.kurtosis(z)) < 4 * math.sqrt(24 / n)
    assert stats.kstest(z[:200000], "norm").pvalue > 1e-3


# ------------------------------------------------------------------ European
@pytest.mark.parametrize("N,M,seed,typ,q,anti", [
    (10000, 50, 42, "call", 0.0, True), (10000, 50, 42, "put", 0.02, True), (100000, 252, 42, "call", 0.0, True),
    (100000, 1, 42, "put", 0.0, True), (4097, 7, 3, "call", 0.0, True), (1, 1, 0, "put", 0.0, True),
    (255, 3, 9, "call", 0.0, False), (70001, 13, 5, "put", 0.01, False),
])
def test_european_matches_same_stream_checker(N, M, seed, typ, q, anti):
    S, K, T, r, v = ATM
    st = _hip.european(S, K, T, r, v, q, typ == "call", N, M, seed, anti)
    sx, sxx, *_m, n = po.european_moments(S, K, T, r, v, q, typ == "call", N, M, seed, anti)
    price, se = po.price_and_error(sx, sxx, n, r, T)
    assert st.n == n
    assert st.sum == pytest.approx(sx, rel=REL_STREAM_TOL, abs=1e-9)
    assert st.sumsq == pytest.approx(sxx, rel=4 * REL_STREAM_TOL, abs=1e-9)
    assert st.price == pytest.approx(price, rel=REL_STREAM_TOL, abs=1e-9)
    assert st.std_error == pytest.approx(se, rel=1e-4, abs=1e-9)


def test_golden_reference_prices_within_3_sigma(golden):
    for c in golden["price"]:
        N, M, seed, method = c["ctor"]
        if method == "qmc":
            continue            # near-exact gate of its own: test_qmc_matches_reference_sobol
        S, K, T, r, v, typ, q = c["args"]
        kw = {} if c["call_seed"] is None else {"seed": c["call_seed"]}
        res = ol.MonteCarloPricer(N, M, seed, ol.MCMethod(method)).price(S, K, T, r, v, typ, q, return_error=True, **kw)
        assert res.n_paths == c["n_paths"]
        if N < 100:      # tiny cases: only sanity (se of 2..14 samples is meaningless)
            assert res.price >= 0
            continue
        assert abs(res.price - c["price"]) <= 3 * math.hypot(res.std_error, c["std_error"]), c["ctor"]
        assert abs(res.std_error / c["std_error"] - 1) < 0.05, c["ctor"]
        bs = ol.black_scholes(S, K, T, r, v, typ, q)
        assert abs(res.price - bs) <= 3.5 * res.std_error, c["ctor"]


def test_headline_config_1m_x_252_within_3_sigma_of_bs_and_reference(golden):
    g4 = next(c for c in golden["price"] if c["ctor"][0] == 1000000)
    for seed in (42, 43, 44):
        res = ol.MonteCarloPricer(1_000_000, 252, seed).price(*ATM, "call", return_error=True)
        assert res.n_paths == 2_000_000
        assert abs(res.price - BS_CALL) <= 3 * res.std_error
        assert abs(res.price - g4["price"]) <= 3 * math.hypot(res.std_error, g4["std_error"])
        assert abs(res.std_error / g4["std_error"] - 1) < 0.02


def test_reference_test_suite_assertions():
    """tests/test_monte_carlo.py of the reference, re-run against the device pricer."""
    p = ol.MonteCarloPricer(num_simulations=10000, num_steps=50, seed=42)
    call = p.price(S=100, K=100, T=1.0, r=0.05, sigma=0.2, option_type="call", q=0.0)
    assert type(call) is float and 0 < call < 100 and abs(call - BS_CALL) < 1.0               # :119-131
    put = p.price(S=100, K=100, T=1.0, r=0.05, sigma=0.2, option_type="put", q=0.0)
    assert put > 0 and abs(put - ol.black_scholes(100, 100, 1.0, 0.05, 0.2, "put")) < 1.0        # :133-141
    assert p.price(100, 100, 1.0, 0.05, 0.2, "call", seed=42) == p.price(100, 100, 1.0, 0.05, 0.2, "call", seed=42)  # :153-158
    res = p.price(100, 100, 1.0, 0.05, 0.2, "call", return_error=True)
    assert res.price > 0 and 0 < res.std_error < res.price                                    # :160-168
    c, q_ = p.price(100, 100, 1.0, 0.05, 0.2, "call", 0.02), p.price(100, 100, 1.0, 0.05, 0.2, "put", 0.02)
    assert abs((c - q_) - (100 * math.exp(-0.02) - 100 * math.exp(-0.05))) < 2.0              # :509-521
    assert p.price(110, 100, 1.0, 0.05, 0.2, "call") > p.price(100, 100, 1.0, 0.05, 0.2, "call") > p.price(90, 100, 1.0, 0.05, 0.2, "call")
    assert p.price(100, 100, 1.0, 0.05, 0.4, "call") > p.price(100, 100, 1.0, 0.05, 0.1, "call")
    assert p.price(100, 100, 2.0, 0.05, 0.2, "call") > p.price(100, 100, 0.25, 0.05, 0.2, "call")
    assert p.price(100, 100, 1.0, 0.05, 0.2, "call", seed=1) != p.price(100, 100, 1.0, 0.05, 0.2, "call", seed=2)
    assert p.price(100, 100, 1.0, 0.05, 0.2, "banana") == p.price(100, 100, 1.0, 0.05, 0.2, "put")   # :140-143


def test_unseeded_pricer_is_self_consistent():
    p = ol.MonteCarloPricer(5000, 4)            # seed drawn once at construction
    assert p.price(*ATM, "call") == p.price(*ATM, "call")


def test_fast_method_forces_single_step():
    a = ol.MonteCarloPricer(20000, 50, 7, ol.MCMethod.FAST).price(*ATM, "call")
    b = ol.MonteCarloPricer(20000, 1, 7, ol.MCMethod.NUMPY).price(*ATM, "call")
    assert a == b


def test_argument_errors():
    with pytest.raises(ol.AccelerationError):
        _hip.european(*ATM, 0.0, True, 0, 4, 1)
    with pytest.raises(ol.AccelerationError):
        _hip.european(*ATM, 0.0, True, 10, 0, 1)
    with pytest.raises(ValueError):
        ol.MonteCarloPricer(10, 0, 1).price(*ATM, "call")


# ------------------------------------------------------------------ QMC (scrambled Sobol)
def test_qmc_matches_reference_sobol(golden):
    """The device expands SciPy's own scrambled direction matrix, so uniforms are bit-equal to the
    reference's and the only differences are the inverse normal (ndtri_w vs Cephes ndtri, ~2e-15)
    and the summation order over dims: prices agree to 1e-10 relative, not merely 3 sigma."""
    for c in golden["price"]:
        N, M, seed, method = c["ctor"]
        if method != "qmc":
            continue
        S, K, T, r, v, typ, q = c["args"]
        p = ol.MonteCarloPricer(N, M, seed, ol.MCMethod.QMC)
        res = p.price(S, K, T, r, v, typ, q, return_error=True)
        assert res.n_paths == c["n_paths"] == N                      # no antithetic mirror (gbm_qmc.py:44-46)
        assert res.price == pytest.approx(c["price"], rel=1e-10)
        assert res.std_error == pytest.approx(c["std_error"], rel=1e-9)
        st = p._simulate(S, T, r, v, q)
        assert st.shape == (c["terminal_len"],)
        assert np.allclose(st[:4], c["terminal_head"], rtol=1e-12, atol=0)
        assert np.allclose(st[N // 2:N // 2 + 4], c["terminal_mid"], rtol=1e-12, atol=0)
        assert type(p.price(S, K, T, r, v, typ, q)) is float


@pytest.mark.parametrize("N,M,seed", [(1000, 3, 7), (4096, 64, 1), (777, 252, 9)])
def test_qmc_terminal_array_vs_oracle(N, M, seed):
    got = ol.MonteCarloPricer(N, M, seed, ol.MCMethod.QMC)._simulate(100.0, 1.0, 0.05, 0.2, 0.01)
    want = orc.terminal_sobol(100.0, 1.0, 0.05, 0.2, 0.01, N, M, seed)
    assert np.allclose(got, want, rtol=1e-11, atol=0)
    a = _hip.european_qmc_terminal(100.0, 1.0, 0.05, 0.2, 0.01, 300, *ol.monte_carlo.sobol_tables(M, seed), point_offset=400)
    assert np.array_equal(a, got[400:700])                           # point index = global path index


@pytest.mark.parametrize("off,M", [((1 << 16) - 3, 16), ((1 << 20) + 5, 9), ((1 << 22) - 130, 12), ((1 << 24) + 1, 5), ((1 << 26) - 64, 3)])
def test_qmc_points_far_into_the_sequence_equal_scipys(off, M):
    """A shard of a large Sobol launch (multi-GPU or chunked: point index = global path index) starts millions of points in, where the
    high bits of the Gray code select the late columns of the direction matrix: 257 points from index `off` on (crossing a power of two
    or ending on one) against SciPy's own engine fast-forwarded there, through the reference's pipeline (gbm_qmc.py:32-46)."""
    from scipy.stats import norm
    from scipy.stats.qmc import Sobol
    S, T, r, v, q, seed, N = 100.0, 1.0, 0.05, 0.2, 0.01, 13, 257
    eng = Sobol(d=M, scramble=True, seed=seed)
    eng.fast_forward(off)
    z = norm.ppf(np.clip(eng.random(N), 1e-10, 1 - 1e-10))
    want = np.exp(np.log(S) + (r - q - 0.5 * v * v) * (T / M) * M + v * np.sqrt(T / M) * np.sum(z, axis=1))
    got = _hip.european_qmc_terminal(S, T, r, v, q, N, *ol.monte_carlo.sobol_tables(M, seed), point_offset=off)
    assert np.allclose(got, want, rtol=1e-11, atol=0)


@pytest.mark.parametrize("knob,off,M", [(0, (1 << 24) + 64 * 5, 40), (0, (1 << 26) - 512, 33), (0, (1 << 29) + 64, 130), (1, (1 << 25) + 512 * 3, 9),
                                        (1, (1 << 28) - 1024, 70), (2, (1 << 22) + 128, 300)])
def test_aligned_qmc_kernels_far_into_the_sequence_equal_scipys(knob, off, M):
    """The ALIGNED Sobol kernels (round 5: the wave's high Gray bits folded lane-per-dimension, olmc_kernels.h qmc_point_sum<true> /
    qmc_block_sums<2, true>) where the high bits are not zero: 321 points from a 64- / 512-aligned index far into the sequence --
    split workgroups (the default shape at this size, and forced: knob 2) and eight points per thread (knob 1), quarters of 8 .. 75
    dimensions, a ragged last wave -- against SciPy's own engine fast-forwarded there, through the reference's pipeline
    (gbm_qmc.py:32-46)."""
    from scipy.stats import norm
    from scipy.stats.qmc import Sobol
    S, T, r, v, q, seed, N = 100.0, 1.0, 0.05, 0.2, 0.01, 13, 321
    eng = Sobol(d=M, scramble=True, seed=seed)
    eng.fast_forward(off)
    z = norm.ppf(np.clip(eng.random(N), 1e-10, 1 - 1e-10))
    want = np.exp(np.log(S) + (r - q - 0.5 * v * v) * (T / M) * M + v * np.sqrt(T / M) * np.sum(z, axis=1))
    try:
        _hip.tune(_hip.TUNE_QMC_BLOCK, knob)
        got = _hip.european_qmc_terminal(S, T, r, v, q, N, *ol.monte_carlo.sobol_tables(M, seed), point_offset=off)
    finally:
        _hip.tune(_hip.TUNE_QMC_BLOCK, 0)
    assert np.allclose(got, want, rtol=1e-11, atol=0)


def test_qmc_prices_the_same_bits_from_tables_derived_without_scipys_privates(monkeypatch):
    """optionslab_amd.monte_carlo._derive_sobol_tables: a SciPy without `_sv` / `_shift` still prices on the same Sobol points -- the
    price, its Greeks and the terminal array from the derived tables are the bits the private tables give."""
    from scipy.stats import qmc
    from optionslab_amd import monte_carlo as mc

    real = qmc.Sobol
    p = ol.MonteCarloPricer(5000, 24, 11, ol.MCMethod.QMC)
    mc._sobol_cache.clear()
    want = (p.price(*ATM, "call", return_error=True), tuple(p.greeks(*ATM, "put").items()), p._simulate(100.0, 1.0, 0.05, 0.2, 0.0).tobytes())

    class PublicOnly:
        bits = 30

        def __init__(self, d, scramble=True, seed=None):
            self._eng = real(d=d, scramble=scramble, seed=seed)

        def random(self, n):
            return self._eng.random(n)

        def fast_forward(self, n):
            self._eng.fast_forward(n)
            return self

    monkeypatch.setattr(qmc, "Sobol", PublicOnly)
    mc._sobol_cache.clear()
    try:
        got = (p.price(*ATM, "call", return_error=True), tuple(p.greeks(*ATM, "put").items()), p._simulate(100.0, 1.0, 0.05, 0.2, 0.0).tobytes())
        assert not mc._sobol_cache[(24, 11)][0][:, 13:].any()          # 5000 points select 13 columns: these tables WERE derived
    finally:
        monkeypatch.setattr(qmc, "Sobol", real)
        mc._sobol_cache.clear()
    assert got == want


@pytest.mark.parametrize("N,M,seed", [(1024, 12, 42), (333, 40, 5)])
def test_qmc_standalone_backends_vs_oracle(N, M, seed, golden):
    """simulate_gbm_qmc / simulate_gbm_qmc_antithetic as exported functions (src/simulation/__init__.py)."""
    plain = ol.simulate_gbm_qmc_hip(100.0, 1.0, 0.05, 0.2, 0.01, N, M, seed)
    assert np.allclose(plain, orc.terminal_sobol(100.0, 1.0, 0.05, 0.2, 0.01, N, M, seed), rtol=1e-11, atol=0)
    both = ol.simulate_gbm_qmc_antithetic_hip(100.0, 1.0, 0.05, 0.2, 0.01, N, M, seed)
    want = orc.terminal_sobol_antithetic(100.0, 1.0, 0.05, 0.2, 0.01, N, M, seed)
    assert both.shape == (2 * N,) and np.allclose(both, want, rtol=1e-11, atol=0)
    # pos * neg = (S exp(drift T))^2 for every point: the mirror really is -z of the same point
    assert np.allclose(both[:N] * both[N:], (100.0 * np.exp((0.05 - 0.01 - 0.02) * 1.0)) ** 2, rtol=1e-12)
    if (N, M, seed) == (1024, 12, 42):
        g = golden["qmc_antithetic"]
        assert np.allclose(both[:4], g["head"], rtol=1e-11) and np.allclose(both[N:N + 4], g["mid"], rtol=1e-11)


def test_qmc_greeks_and_control_variate_run_on_the_same_points():
    p = ol.MonteCarloPricer(2**14, 16, 42, ol.MCMethod.QMC)
    g = p.greeks(*ATM, "call", include_second_order=False)
    want = orc.fd_greeks(orc.OraclePricer(2**14, 16, 42, "qmc").price, *ATM, "call", 0.0, include_second_order=False)
    for k in g:
        assert g[k] == pytest.approx(want[k], rel=1e-6, abs=1e-7), k
    cv = p.price_with_control_variate(*ATM, "call")
    assert cv == pytest.approx(orc.OraclePricer(2**14, 16, 42, "qmc").price_with_control_variate(*ATM, "call"), rel=1e-9)
    with pytest.raises(ol.AccelerationError):
        _hip.european_qmc(*ATM, 0.0, True, 10, np.zeros((4, 31), np.uint32), np.zeros(4, np.uint32))   # bits != 30


# Finite-difference steps the Greeks are quotients over (unified_greeks.py:274-277): an error eps on every price shows as about
# eps x this factor in the Greek (sum of |coefficients| / step)
def _fd_amplification(S=100.0):
    h_S, h_v, h_r, h_T = max(1e-4, 0.01 * S), 0.01, 1e-4, 1 / 365.0
    return dict(price=1.0, delta=1 / h_S, gamma=4 / h_S**2, vega=1 / h_v, theta=2 / h_T, rho=1 / h_r, vanna=1 / (h_S * h_v),
                charm=2 / (h_S * h_T), vomma=4 / h_v**2)


def test_qmc_fused_greeks_equal_the_reference_and_the_literal_form(golden):
    """olmc_european_qmc_greeks_fd: the 8 / 14 bumped contracts of compute_greeks_unified on the Sobol points in ONE launch.
    Against the reference's own numbers (golden qmc_greeks: every device QMC price is within 1e-10 relative of the reference's,
    test_qmc_matches_reference_sobol, so each Greek is within that error times its finite-difference amplification) and against
    the literal 8 / 14 single-contract launches on the device (contracts that share a vol take scale x S_T(base): a few ulp)."""
    amp = _fd_amplification()
    for c in golden["qmc_greeks"]:
        N, M, seed, _ = c["ctor"]
        S, K, T, r, v, typ, q = c["args"]
        p = ol.MonteCarloPricer(N, M, seed, ol.MCMethod.QMC)
        fused = p.greeks(S, K, T, r, v, typ, q, include_second_order=c["include_second_order"])
        literal = ol.compute_greeks_unified(p, S, K, T, r, v, typ, q, include_second_order=c["include_second_order"], fused=False)
        assert list(fused) == list(literal) == c["keys"]
        price = c["values"]["price"]
        for k in c["keys"]:
            assert fused[k] == pytest.approx(c["values"][k], abs=2e-10 * price * amp[k]), (c["ctor"], k, "vs the reference")
            assert fused[k] == pytest.approx(literal[k], abs=1e-13 * price * amp[k] + 1e-12), (c["ctor"], k, "vs literal launches")
        assert fused == ol.compute_greeks_unified(p, S, K, T, r, v, typ, q, include_second_order=c["include_second_order"])


def test_qmc_batch_prices_each_contract_as_its_own_launch_does():
    """olmc_european_qmc_batch: arbitrary contracts (nothing shared but the points), both launch shapes, ragged ranges."""
    rng = np.random.default_rng(11)
    for N, M, off, k in [(5000, 12, 0, 16), (777, 33, 1234, 3), (4096, 64, 0, 9), (2**20 + 5, 4, 3, 5)]:
        tables = ol.monte_carlo.sobol_tables(M, 9)
        opts = [(float(rng.uniform(80, 120)), float(rng.uniform(80, 120)), float(rng.uniform(0.3, 2.0)), float(rng.uniform(0.0, 0.08)),
                 float(rng.uniform(0.1, 0.5)), float(rng.uniform(0.0, 0.03)), bool(i % 2)) for i in range(k)]
        opts[1] = opts[0][:3] + (opts[0][3] + 1e-3,) + opts[0][4:]            # an r-bump of contract 0: same vol, takes the scaled price
        got = _hip.european_qmc_batch(opts, N, *tables, point_offset=off)
        for o, g in zip(opts, got):
            one = _hip.european_qmc(*o[:6], o[6], N, *tables, point_offset=off)
            assert g.n == one.n == N
            assert g.price == pytest.approx(one.price, rel=1e-13, abs=1e-13) and g.std_error == pytest.approx(one.std_error, rel=1e-10)
    with pytest.raises(ol.AccelerationError):
        _hip.european_qmc_batch([opts[0]] * 17, 100, *tables)


# ------------------------------------------------------------------ terminal array (backend contract)
@pytest.mark.parametrize("N,M", [(1000, 12), (257, 1), (5000, 6)])
def test_terminal_array_layout_and_values(N, M):
    st = ol.simulate_gbm_hip(100.0, 1.0, 0.05, 0.2, 0.01, N, M, 3)
    want = po.european_terminal(100.0, 1.0, 0.05, 0.2, 0.01, N, M, 3)
    assert st.dtype == np.float64 and st.shape == (2 * N,)
    assert np.allclose(st, want, rtol=2e-5, atol=0)
    one_leg = ol.simulate_gbm_hip(100.0, 1.0, 0.05, 0.2, 0.01, N, M, 3, antithetic=False)
    assert np.array_equal(one_leg, st[:N])
    p = ol.MonteCarloPricer(N, M, 3)
    assert np.array_equal(p._simulate(100.0, 1.0, 0.05, 0.2, 0.01), st)
    # price() == the reference tail applied to the returned array (monte_carlo.py:140-150)
    x = np.maximum(st - 100.0, 0.0)
    res = p.price(100.0, 100.0, 1.0, 0.05, 0.2, "call", 0.01, return_error=True)
    assert res.price == pytest.approx(float(np.exp(-0.05) * np.mean(x)), rel=1e-12)
    assert res.std_error == pytest.approx(float(np.exp(-0.05) * np.std(x) / np.sqrt(len(x))), rel=1e-9)
    assert np.array_equal(ol.simulate_gbm_hip_fast(100.0, 1.0, 0.05, 0.2, 0.0, N, 9),
                          ol.simulate_gbm_hip(100.0, 1.0, 0.05, 0.2, 0.0, N, 1, 9))


# ------------------------------------------------------------------ sharding (size-independent property)
def test_shards_add_up_to_the_whole():
    S, K, T, r, v = ATM
    N, M, seed = 300_001, 21, 11
    whole = _hip.european(S, K, T, r, v, 0.0, True, N, M, seed)
    for world in (2, 3, 8):
        parts = []
        for k in range(world):
            lo, hi = ol.sharding.shard_bounds(N, k, world)
            st = _hip.european(S, K, T, r, v, 0.0, True, hi - lo, M, seed, True, path_offset=lo)
            parts.append((st.sum, st.sumsq, st.n))
        comb = _hip.combine_stats(parts, r, T)
        assert comb.n == whole.n
        assert comb.price == pytest.approx(whole.price, rel=1e-12)
        assert comb.std_error == pytest.approx(whole.std_error, rel=1e-9)


def test_config5_eight_shards_of_8m_x_252_equal_the_64m_launch():
    """BASELINE configs[4] at size on one GPU: 64M paths x 252 steps as the 8 contiguous 8M-path shards the 8 ranks
    would own (path_offset = k * 8M; gbm_numpy.py:43-51 is the shape being sharded, monte_carlo.py:140-150 the
    reduction), combined in rank order, against the single 64M-path launch."""
    S, K, T, r, v = ATM
    M, per, world, seed = 252, 8_000_000, 8, 42
    parts = []
    for k in range(world):
        lo, hi = ol.sharding.shard_bounds(per * world, k, world)
        assert (lo, hi) == (k * per, (k + 1) * per)
        st = _hip.european(S, K, T, r, v, 0.0, True, hi - lo, M, seed, True, path_offset=lo)
        assert st.n == 2 * per and abs(st.price - BS_CALL) <= 4 * st.std_error        # each shard is itself a valid 8M-path pricing
        parts.append((st.sum, st.sumsq, st.n))
    comb = _hip.combine_stats(parts, r, T)
    whole = _hip.european(S, K, T, r, v, 0.0, True, per * world, M, seed, True)
    assert comb.n == whole.n == 128_000_000
    assert comb.price == pytest.approx(whole.price, rel=1e-12)
    assert comb.std_error == pytest.approx(whole.std_error, rel=1e-9)
    assert 1.25e-3 < comb.std_error < 1.35e-3                       # se ~ 1.3e-3 at 64M paths
    assert abs(comb.price - BS_CALL) <= 3 * comb.std_error
    # the shards are disjoint streams: no two of them may return the same sums
    assert len({p[0] for p in parts}) == world


@pytest.mark.parametrize("N,M", [(1_000_000, 252), (70_001, 64), (65_536, 100), (300, 1024), (63, 67), (1, 64), (262_144 + 77, 129), (5_000, 63)])
def test_split_workgroups_return_the_bits_of_whole_path_workgroups(N, M):
    """The paths beyond a whole number of workgroups per compute unit are priced by SPLIT workgroups (64 paths, each wave a
    quarter of the steps, meeting in LDS: olmc_kernels.h european_path_kernel).  The fp64 sum over the fp32 groups has one
    canonical association, so the launch shape does not show in a single bit of any PATH: the terminal array with the knob
    off (one shape throughout) equals the default byte for byte.  The reduced sums group the same per-path values into
    different workgroup rows (256 against 64 paths), so they agree to reduction-order rounding, exactly like the shards of
    test_shards_add_up_to_the_whole (1e-13 here).  M = 63 has fewer than four groups: never split."""
    S, K, T, r, v = ATM
    def everything():
        st = _hip.european(S, K, T, r, v, 0.01, True, N, M, 9, True, path_offset=12345)
        na = _hip.european(S, K, T, r, v, 0.01, False, N, M, 9, False)
        cv = _hip.european_cv(S, K, T, r, v, 0.0, True, N, M, 9)
        out = [st.sum, st.sumsq, st.n, na.sum, na.sumsq, cv.sum_d, cv.sum_s, cv.sum_ds, cv.value]
        if N <= 300_000:
            _g, evals = _hip.european_greeks_fd(S, K, T, r, v, 0.0, True, N, M, 9, True)
            out += [e.sum for e in evals]                 # the 14 bumped contracts on the common normals (NSETS = 16 kernel)
            out.append(_hip.european_terminal(S, T, r, v, 0.0, N, M, 9, True).tobytes())
        return out
    split = everything()
    _hip.tune(_hip.TUNE_SPLIT_TAIL, -1)
    try:
        whole = everything()
    finally:
        _hip.tune(_hip.TUNE_SPLIT_TAIL, 0)
    assert len(split) == len(whole)
    for a_, b_ in zip(split, whole):
        if isinstance(a_, bytes):
            assert a_ == b_                              # every terminal price, bit for bit
        else:
            assert a_ == pytest.approx(b_, rel=1e-13, abs=1e-300)
    sx, sxx, *_m, n = po.european_moments(S, K, T, r, v, 0.01, True, min(N, 70_001), M, 9, True, 12345)
    if N <= 70_001:
        assert split[0] == pytest.approx(sx, rel=REL_STREAM_TOL) and split[2] == n


def test_completion_by_polled_flag_equals_completion_by_stream_synchronize():
    """Blocking entry points return as soon as the kernel's last wave has raised a flag in pinned host memory behind its
    results (system-scope release; olmc.hip sync_or_recover), not when the runtime has seen the kernel retire.  Same
    numbers either way -- for one-launch calls, for the many-launch LSM call (never armed) and across a change of
    workspace size -- and a tiny launch right behind a large one must not read the large one's results."""
    S, K, T, r, v = ATM

    def everything():
        big = _hip.european(S, K, T, r, v, 0.0, True, 2_000_000, 64, 3, True)
        tiny = _hip.european(S, K, T, r, v, 0.0, False, 7, 5, 4, True)           # right behind the big one
        out = [big.sum, big.sumsq, big.n, tiny.sum, tiny.sumsq, tiny.n]
        out += [_hip.asian(S, K, T, r, v, 0.0, True, False, 30_000, 64, 5).sum, _hip.barrier(S, K, T, r, v, 0.0, True, 120.0, 0, 30_000, 64, 6).sum,
                _hip.heston(S, K, T, r, 0.0, True, 2.0, 0.04, 0.3, -0.7, 0.04, 30_000, 32, 7).sum, _hip.european_cv(S, K, T, r, v, 0.0, True, 30_000, 64, 8).value,
                _hip.american_lsm(S, K, T, r, v, 0.0, False, 20_000, 20, 3, 9).sum]
        out += [_hip.european(S, K, T, r, v, 0.0, True, n, 64, 10 + i, True).sum for i, n in enumerate((1, 255, 100_000, 65, 300_000, 2))]
        return out

    polled = everything()
    _hip.tune(_hip.TUNE_POLL, -1)
    try:
        synced = everything()
    finally:
        _hip.tune(_hip.TUNE_POLL, 0)
    assert polled == synced
    assert polled[2] == 4_000_000 and polled[5] == 14 and polled[3] != polled[0]


def test_large_path_offsets_use_the_high_counter_word():
    S, K, T, r, v = ATM
    off = (1 << 32) - 100
    st = _hip.european(S, K, T, r, v, 0.0, True, 1000, 5, 1, True, path_offset=off)
    sx, sxx, *_m, n = po.european_moments(S, K, T, r, v, 0.0, True, 1000, 5, 1, True, off)
    assert st.sum == pytest.approx(sx, rel=REL_STREAM_TOL)


def test_single_process_multi_gpu_entry_point_on_one_gpu():
    S, K, T, r, v = ATM
    a = _hip.multi_gpu_european(S, K, T, r, v, 0.0, True, 200_000, 16, 5, True, 1)    # RCCL all-reduce, 1 rank
    b = _hip.european(S, K, T, r, v, 0.0, True, 200_000, 16, 5, True)
    assert (a.sum, a.sumsq, a.n, a.price, a.std_error) == (b.sum, b.sumsq, b.n, b.price, b.std_error)
    with pytest.raises(ol.AccelerationError):
        _hip.multi_gpu_european(S, K, T, r, v, 0.0, True, 1000, 4, 5, True, 64)
    # the other two payloads of the single-process form (count 17 / 33 and 6), one rank through the real all-reduce;
    # several ranks are rehearsed on this one device in tests/test_gpu_instrumented.py
    for second in (False, True):
        one, e1 = _hip.european_greeks_fd(S, K, T, r, v, 0.0, True, 200_000, 16, 5, second)
        many, e2 = _hip.multi_gpu_greeks_fd(S, K, T, r, v, 0.0, True, 200_000, 16, 5, second, 1)
        assert many == one and [(x.sum, x.sumsq, x.n) for x in e1] == [(x.sum, x.sumsq, x.n) for x in e2]
    m1 = _hip.european_cv(S, K, T, r, v, 0.0, True, 200_000, 16, 5, True)
    m2 = _hip.multi_gpu_european_cv(S, K, T, r, v, 0.0, True, 200_000, 16, 5, True, 1)
    assert (m1.sum_d, m1.sum_s, m1.sum_dd, m1.sum_ss, m1.sum_ds, m1.n, m1.value) == (m2.sum_d, m2.sum_s, m2.sum_dd, m2.sum_ss, m2.sum_ds, m2.n, m2.value)
    assert _hip.device_info()["device"] == 0


# ------------------------------------------------------------------ Greeks
def test_batch_equals_separate_launches():
    opts = [(100.0, 100.0, 1.0, 0.05, 0.2, 0.0, True), (101.0, 100.0, 1.0, 0.05, 0.2, 0.0, True),
            (99.0, 95.0, 0.5, 0.03, 0.25, 0.01, False)]
    for k in (1, 2, 3):
        got = _hip.european_batch(opts[:k], 50_000, 10, 42)
        for o, st in zip(opts[:k], got):
            S, K, T, r, v, q, c = o
            one = _hip.european(S, K, T, r, v, q, c, 50_000, 10, 42)
            assert st.price == pytest.approx(one.price, rel=1e-13) and st.n == one.n
    many = [(100.0 + i, 100.0, 1.0, 0.05, 0.2, 0.0, True) for i in range(16)]
    got = _hip.european_batch(many, 20_000, 6, 1)
    assert all(a.price < b.price for a, b in zip(got, got[1:]))
    with pytest.raises(ol.AccelerationError):
        _hip.european_batch(many + many[:1], 100, 2, 1)


@pytest.mark.parametrize("second", [False, True])
def test_fused_greeks_equal_literal_bump_and_reprice(second):
    p = ol.MonteCarloPricer(200_000, 52, 42)
    fused = ol.compute_greeks_unified(p, *ATM, "call", 0.0, include_second_order=second)
    literal = ol.compute_greeks_unified(p, *ATM, "call", 0.0, include_second_order=second, fused=False)
    assert list(fused) == list(literal)
    assert list(fused) == ["price", "delta", "gamma", "vega", "theta", "rho"] + (["vanna", "charm", "vomma"] if second else [])
    for k in fused:
        assert fused[k] == pytest.approx(literal[k], rel=1e-8, abs=1e-8), k
    assert fused == p.greeks(*ATM, "call", 0.0, include_second_order=second)


@pytest.mark.parametrize("second", [False, True])
@pytest.mark.parametrize("N,M", [(200_000, 52), (777, 7), (1_000_000, 252)])
def test_prices_only_greeks_equal_the_greeks_with_their_evaluations(second, N, M):
    """olmc_european_greeks_fd with evals == NULL (what MonteCarloPricer.greeks() calls) launches the sum-only form of the fused kernel
    (kSumOnly: no sums of squares, half the exchanges); with evals it launches the full one.  Same paths, same payoffs, another
    association of the lane sums: every Greek within 1e-13 x price x its finite-difference amplification of the other form's, and
    the evaluations the full form returns carry the standard errors the lean form never computed."""
    amp = _fd_amplification()
    lean, none = _hip.european_greeks_fd(*ATM, 0.0, True, N, M, 42, second, want_evals=False)
    full, evals = _hip.european_greeks_fd(*ATM, 0.0, True, N, M, 42, second, want_evals=True)
    assert none == [] and len(evals) == 14 and evals[0].std_error > 0 and evals[0].n == 2 * N
    keys = ol.monte_carlo.GREEK_KEYS[:9 if second else 6]
    for k, a, b in zip(keys, lean, full):
        assert a == pytest.approx(b, abs=1e-13 * full[0] * amp[k] + 1e-13), k
    assert lean[0] == pytest.approx(evals[0].price, rel=1e-13)


def test_greeks_against_reference_golden_and_black_scholes(golden):
    exact = orc.bs_greeks(*ATM, "call")
    g7 = next(c for c in golden["greeks"] if c["ctor"][:2] == [100000, 252] and not c["include_second_order"])
    # tolerances: SURVEY §8c -- the BS-vs-reference gaps at N=1e5 scaled by sqrt(1e5/N), x3
    N = 1_000_000
    scale = 3 * math.sqrt(1e5 / N)
    tol = dict(delta=3e-4 * scale * 3, gamma=4e-4 * scale * 3, vega=0.07 * scale * 3, theta=0.012 * scale * 3, rho=0.03 * scale * 3)
    g = ol.MonteCarloPricer(N, 252, 42).greeks(*ATM, "call", include_second_order=False)
    for k, t in tol.items():
        assert abs(g[k] - exact[k]) <= t, (k, g[k], exact[k])
        assert abs(g[k] - g7["values"][k]) <= 4 * abs(g7["values"][k] - exact[k]) + t, k
    put = ol.MonteCarloPricer(N, 252, 42).greeks(*ATM, "put", include_second_order=False)
    assert put["delta"] == pytest.approx(g["delta"] - 1.0, abs=2e-3)      # put-call parity of delta (q = 0)
    assert put["gamma"] == pytest.approx(g["gamma"], abs=1e-3)


def test_greeks_short_dated_branch():
    p = ol.MonteCarloPricer(50_000, 4, 42)
    g = p.greeks(100.0, 100.0, 0.002, 0.05, 0.2, "call")
    lit = ol.compute_greeks_unified(p, 100.0, 100.0, 0.002, 0.05, 0.2, "call", fused=False)
    assert g["charm"] == 0.0 and g["theta"] == pytest.approx(-g["price"] / 0.002, rel=1e-12)
    for k in g:
        assert g[k] == pytest.approx(lit[k], rel=1e-8, abs=1e-8)


# ------------------------------------------------------------------ control variate
def test_control_variate(golden):
    for c in golden["control_variate"]:
        N, M, seed, _ = c["ctor"]
        S, K, T, r, v, typ, q = c["args"]
        p = ol.MonteCarloPricer(N, M, seed)
        got = p.price_with_control_variate(S, K, T, r, v, typ, q)
        assert type(got) is float
        # same-stream checker: rebuild the estimator from its five moments
        sx, sxx, ss, sss, sxs, n = po.european_moments(S, K, T, r, v, q, typ == "call", N, M, seed, True)
        disc = math.exp(-r * T)
        md, ms = disc * sx / n, ss / n
        beta = ((disc * sxs - n * md * ms) / (n - 1)) / ((sss - n * ms * ms) / (n - 1))
        assert got == pytest.approx(md - beta * (ms - S * math.exp((r - q) * T)), rel=5e-6)
        plain = p.price(S, K, T, r, v, typ, q, return_error=True)
        bs = ol.black_scholes(S, K, T, r, v, typ, q)
        assert abs(got - bs) <= 3 * plain.std_error and abs(got - c["value"]) <= 3 * plain.std_error


# ------------------------------------------------------------------ Asian
@pytest.mark.parametrize("geometric,typ,anti,N,M", [(False, "call", False, 20000, 64), (True, "put", False, 20000, 64),
                                                    (False, "put", True, 5001, 13), (True, "call", True, 777, 252),
                                                    # BASELINE C4's 1024 dates (64 fp64 groups of 16), and every shape of the trailing group
                                                    (False, "call", False, 3000, 1024), (False, "call", True, 2000, 1023), (True, "call", False, 3000, 1021)]
                         + [(g, "call", a, 300, m) for m in (1, 3, 4, 15, 16, 17, 19, 20, 29, 31, 32, 33) for g, a in ((False, True), (True, False))])
def test_asian_matches_same_stream_checker(geometric, typ, anti, N, M):
    sx, sxx, n = po.asian_moments(100.0, 100.0, 1.0, 0.05, 0.2, 0.02, typ == "call", geometric, N, M, 7, anti)
    # default = the reference's arithmetic (fp64 cumsum + fp64 exp per date); fast = the opt-in fp32-exponent kernel
    for fast in ((False,) if geometric else (False, True)):
        st = _hip.asian(100.0, 100.0, 1.0, 0.05, 0.2, 0.02, typ == "call", geometric, N, M, 7, anti, fast=fast)
        assert st.n == n
        assert st.sum == pytest.approx(sx, rel=REL_STREAM_TOL) and st.sumsq == pytest.approx(sxx, rel=4 * REL_STREAM_TOL)


def test_asian_fp64_exponent_tracks_the_checker_more_closely_than_the_fp32_one():
    """Same device normals in both kernels, so what separates them from the fp64 checker beyond the normals' own
    2e-5 hardware-transcendental error is the per-date exponential: the reference-precision kernel must not be the
    worse of the two, and the two must agree to the fp32 form's stated 2e-6."""
    args = (100.0, 100.0, 1.0, 0.05, 0.2, 0.0, True, False, 40_000, 1024, 11, False)
    sx, _sxx, _n = po.asian_moments(*args)
    f64, f32 = _hip.asian(*args), _hip.asian(*args, fast=True)
    assert abs(f64.sum - sx) <= abs(f32.sum - sx) + 1e-9 * sx
    assert f32.price == pytest.approx(f64.price, rel=2e-6)


def test_asian_fp32_exponent_keeps_crn_finite_differences_smooth_at_c4_size():
    """BASELINE configs[3] (1M paths x 1024 dates).  Common random numbers, the steps of
    unified_greeks.py:274-277 (h_S = 0.01 S, h_sigma = 0.01): delta, gamma and vega of the fp32-exponent kernel against
    the fp64 kernel on the SAME normals.  This is the licence under which OLMC_AVG_ARITHMETIC_FAST stays in the library;
    the default everywhere (AsianOption.price, ExoticAdapter Greeks) is the fp64 kernel regardless."""
    S, K, T, r, v, q, N, M, seed = 100.0, 100.0, 1.0, 0.05, 0.2, 0.0, 1_000_000, 1024, 42
    h_s, h_v = 0.01 * S, 0.01

    def greeks(fast):
        P = lambda s_, v_: _hip.asian(s_, K, T, r, v_, q, True, False, N, M, seed, False, fast=fast).price
        mid, up, dn, vu, vd = P(S, v), P(S + h_s, v), P(S - h_s, v), P(S, v + h_v), P(S, v - h_v)
        return dict(price=mid, delta=(up - dn) / (2 * h_s), gamma=(up - 2 * mid + dn) / (h_s * h_s), vega=(vu - vd) / (2 * h_v))

    g64, g32 = greeks(False), greeks(True)
    assert 5.6 < g64["price"] < 5.9 and 0.5 < g64["delta"] < 0.65 and g64["gamma"] > 0 and 15 < g64["vega"] < 30
    for name in ("price", "delta", "gamma", "vega"):
        assert g32[name] == pytest.approx(g64[name], rel=1e-6), (name, g32[name], g64[name])


def test_geometric_asian_against_the_exact_discrete_closed_form():
    """ln of the geometric average over t_i = iT/M, i = 1..M is exactly normal, so the discretely monitored geometric
    Asian has a closed form: an analytic anchor for the Asian step loop (the reference's closed form, exotic_options.py:
    133-160, is the continuous limit: 5e-3 lower at 1024 dates, 2e-2 at 252)."""
    S, K, T, r, v, q = 100.0, 100.0, 1.0, 0.05, 0.2, 0.01
    cdf = lambda x: 0.5 * math.erfc(-x / math.sqrt(2.0))
    for M, seed in ((1024, 3), (252, 4), (13, 5), (1, 6)):
        mu = (r - q - 0.5 * v * v) * T * (M + 1) / (2 * M)
        s2 = v * v * T * (M + 1) * (2 * M + 1) / (6 * M * M)
        m, s = math.log(S) + mu, math.sqrt(s2)
        d1 = (m - math.log(K) + s2) / s
        call = math.exp(-r * T) * (math.exp(m + 0.5 * s2) * cdf(d1) - K * cdf(d1 - s))
        put = call - math.exp(-r * T) * (math.exp(m + 0.5 * s2) - K)
        for typ, want in (("call", call), ("put", put)):
            st = _hip.asian(S, K, T, r, v, q, typ == "call", True, 1 << 21, M, seed, True)
            assert abs(st.price - want) <= 3.5 * st.std_error, (M, typ, st.price, want, st.std_error)
        if M == 1:      # one date: the European option
            assert call == pytest.approx(ol.black_scholes(S, K, T, r, v, "call", q), rel=1e-12)


def test_asian_against_reference_golden_and_reference_tests(golden):
    for c in golden["asian"]:
        S, K, T, r, v, q = c["params"]
        o = ol.AsianOption(S, K, T, r, v, q, seed=c["seed"])
        price, se = o.price(c["n_paths"], c["n_steps"], c["avg_type"], c["option_type"], return_error=True)
        assert isinstance(price, np.float64) and price > 0                                   # test_exotic_options.py:56-76
        assert abs(price - c["price"]) <= 3 * math.sqrt(2) * se, c
        if c["avg_type"] == "geometric":                                                    # :92-101 (closed form is continuous-monitoring)
            assert abs(price - c["geometric_closed_form"]) / c["geometric_closed_form"] < 0.05
    a = ol.AsianOption(100, 100, 1.0, 0.05, 0.2, seed=42)
    assert a.price(50000, avg_type="arithmetic", option_type="call") < BS_CALL               # :78-90
    assert a.price(n_paths=1000) == ol.AsianOption(100, 100, 1.0, 0.05, 0.2, seed=42).price(n_paths=1000)  # :103-114
    assert ol.price_asian(100, 100, 1.0, 0.05, 0.2, n_paths=1000, seed=42) == a.price(n_paths=1000)


def test_asian_1m_x_1024_against_the_reference_run_at_that_shape(golden):
    """BASELINE configs[3] at its own shape: the default (fp64-exponent) kernel at 1,000,000 x 1024 against the reference's run of
    exotic_options.py:97-131 at 100,000 x 1024 (`asian_c4_shape` in the fixture; its standard error was taken from the reference's
    own paths).  3 sigma of the two independent errors, the tolerance north_star states.  The geometric average goes through
    `asian_kernel<., true>`, the arithmetic ones through `asian_exp64_kernel`."""
    for c in golden["asian_c4_shape"]:
        S, K, T, r, v, q = c["params"]
        o = ol.AsianOption(S, K, T, r, v, q, seed=c["seed"])
        price, se = o.price(1_000_000, c["n_steps"], c["avg_type"], c["option_type"], return_error=True)
        se_ref = c["std_error_from_reference_paths"]
        assert isinstance(price, np.float64)
        assert se == pytest.approx(se_ref / math.sqrt(10.0), rel=0.02), c          # ten times the paths, no antithetic in either
        assert abs(price - c["price"]) <= 3 * math.sqrt(se * se + se_ref * se_ref), (c, price, se)
        # the opt-in fp32-exponent form prices the same paths: 2e-6 of the default, so it sits inside the same bound
        if c["avg_type"] == "arithmetic":
            fast = _hip.asian(S, K, T, r, v, q, c["option_type"] == "call", False, 1_000_000, c["n_steps"], c["seed"], False, fast=True)
            assert fast.price == pytest.approx(float(price), rel=2e-6)
    call, put = (next(c for c in golden["asian_c4_shape"] if c["avg_type"] == "arithmetic" and c["option_type"] == t) for t in ("call", "put"))
    # arithmetic put-call parity at 1024 dates on the device's own paths: C - P = disc * (mean(A) - K), mean(A) known in closed form
    S, K, T, r, v, q = call["params"]
    o = ol.AsianOption(S, K, T, r, v, q, seed=42)
    c1, e1 = o.price(1_000_000, 1024, "arithmetic", "call", return_error=True)
    p1, e2 = o.price(1_000_000, 1024, "arithmetic", "put", return_error=True)
    dt, g = T / 1024, (r - q) * T / 1024
    mean_avg = S * math.exp(g) * (1 - math.exp(g * 1024)) / (1 - math.exp(g)) / 1024
    # a path pays the call or the put, never both: cov(x_c, x_p) = -E[x_c] E[x_p], so var(C - P) = var C + var P + 2 C P / n
    se_cp = math.sqrt(e1 * e1 + e2 * e2 + 2.0 * float(c1) * float(p1) / 1_000_000)
    assert abs((c1 - p1) - math.exp(-r * T) * (mean_avg - K)) <= 3.5 * se_cp


def test_literal_eight_launch_greeks_equal_the_fused_kernel_at_1m_x_252():
    """BASELINE configs[2] in its literal form -- the 8 (14) `price()` calls of unified_greeks.py:295-358, each its own launch of
    the one-contract kernel -- against the fused kernel at the size the config names.  Same normals, same payoffs, another
    association of the sums: 1e-8 on every Greek."""
    p = ol.MonteCarloPricer(1_000_000, 252, 42)
    for second in (False, True):
        fused = ol.compute_greeks_unified(p, *ATM, "call", 0.0, include_second_order=second)
        literal = ol.compute_greeks_unified(p, *ATM, "call", 0.0, include_second_order=second, fused=False)
        assert list(fused) == list(literal)
        for k in fused:
            assert fused[k] == pytest.approx(literal[k], rel=1e-8, abs=1e-8), (second, k)
    exact = orc.bs_greeks(*ATM, "call")
    assert abs(literal["delta"] - exact["delta"]) < 1e-3 and abs(literal["vega"] - exact["vega"]) < 0.2


def test_asian_greeks_through_exotic_adapter(golden):
    """compute_greeks_unified over ExoticAdapter(AsianOption) against the reference's own run of the same call (20,000 paths x 64 dates,
    unified_greeks.py:177-227) -- every Greek within 3 sigma of the two runs' standard errors.  The reference run's are in the fixture
    (tests/golden/make_golden.py: from the reference's own per-path payoffs under common random numbers); the device runs 100 x the
    paths, so its own error is a tenth of that (1 / sqrt(100)) and the gate is 3 sqrt(1 + 1/100) of the fixture's."""
    c = golden["exotic_adapter_greeks"][0]
    assert c["option"] == "asian" and c["values"] == golden["asian_greeks"]["values"]       # the case the fixture has held since round 1
    ratio = 100
    ad = ol.ExoticAdapter(ol.AsianOption(100.0, 100.0, 1.0, 0.05, 0.2, seed=42), n_paths=ratio * c["n_paths"], n_steps=c["n_steps"], **c["kwargs"])
    g = ol.compute_greeks_unified(ad, 100.0, 100.0, 1.0, 0.05, 0.2, c["option_type"], 0.0, include_second_order=c["include_second_order"])
    assert list(g) == c["keys"]
    for k in c["keys"]:
        assert abs(g[k] - c["values"][k]) <= 3.0 * c["std_errors"][k] * math.sqrt(1.0 + 1.0 / ratio), (k, g[k], c["values"][k], c["std_errors"][k])


# ------------------------------------------------------------------ re-entrancy (Streamlit sessions are threads)
def test_concurrent_callers_get_bit_identical_results_on_contexts_of_their_own():
    """Round 4: a call leases one of up to 8 contexts of the device (stream, workspace, landing buffer, completion word of its own)
    instead of holding one mutex from launch to result, so the pricings of N Streamlit sessions (threads) overlap.  Twelve threads
    -- more than there are contexts: the ninth waits for a lease -- hammer the interactive sizes of the one live UI caller
    (streamlit_app/pages/1_MonteCarlo_Basic.py:111-126: 10k - 200k paths x 1 - 100 steps) with every kind of entry point that owns
    per-context scratch (reduction workspace, terminal array, Sobol table, batch workspace, LSM path matrix); every result must be
    the bits the same call returns single-threaded."""
    import threading
    from optionslab_amd.monte_carlo_unified import MonteCarloPricerUni

    sizes = [(10_000, 50), (100_000, 100), (200_000, 1), (50_000, 16)]
    def calls(k):
        N, M = sizes[k % len(sizes)]
        p = ol.MonteCarloPricer(N, M, 100 + k)
        qp = ol.MonteCarloPricer(1 << 12, 8 + k % 3, 7, ol.MCMethod.QMC)
        return [
            lambda: p.price(100.0 + k, 100.0, 1.0, 0.05, 0.2, "call", return_error=True),
            lambda: tuple(p.greeks(*ATM, "put", include_second_order=bool(k & 1)).items()),
            lambda: p.price_with_control_variate(*ATM, "call"),
            lambda: p._simulate(100.0, 1.0, 0.05, 0.2, 0.0).tobytes(),
            lambda: qp.price(*ATM, "call", return_error=True),
            lambda: ol.AsianOption(*ATM, seed=k).price(20_000, 32, return_error=True),
            lambda: ol.AmericanOption(*ATM, seed=k).price(5_000 + 100 * k, 10),
            lambda: tuple(ol.compute_greeks_unified(ol.ExoticAdapter(ol.AsianOption(*ATM, seed=k), n_paths=10_000, n_steps=50, avg_type=("arithmetic", "geometric")[k & 1]),
                                                    *ATM, "call", include_second_order=False).items()),
            lambda: tuple(ol.compute_greeks_unified(ol.ExoticAdapter(ol.BarrierOption(*ATM, barrier=120.0, seed=k), n_paths=10_000, n_steps=50, barrier_type="up-and-out"),
                                                    *ATM, "call", include_second_order=False).items()),
            lambda: tuple(MonteCarloPricerUni(5_000, 10, seed=k).price_batch(np.array([90.0, 100.0 + k]), np.array([100.0, 100.0]), np.array([1.0, 0.5]),
                                                                              np.array([0.05, 0.05]), np.array([0.2, 0.3]), "call")),
        ]
    n_threads = 12
    want = {k: [f() for f in calls(k)] for k in range(n_threads)}
    got, errs = {}, []
    start = threading.Barrier(n_threads)

    def work(k):
        try:
            fs = calls(k)
            start.wait()
            rounds, t_end = 0, time.monotonic() + float(os.environ.get("OLMC_CONCURRENCY_SECONDS", "0"))     # a long hunt on request
            while rounds < 15 or time.monotonic() < t_end:
                rounds += 1
                got[k] = [f() for f in fs]
                if got[k] != want[k]:
                    raise AssertionError(f"thread {k}: a concurrent call returned other bits than the single-threaded one")
        except Exception as e:  # noqa: BLE001
            errs.append(e)
            start.abort()

    ts = [threading.Thread(target=work, args=(k,)) for k in range(n_threads)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs[:2]
    assert got == want
    assert [f() for f in calls(3)] == want[3]               # and single-threaded again afterwards


def test_one_seed_gives_one_pair_of_sums_thousands_of_times_under_load():
    """`price1 == price2` for equal seeds (the reference's tests/test_monte_carlo.py:153-158) rests on the grid reduction: every
    workgroup's row stored write-through and drained, a ticket, and the LAST workgroup reading ALL rows (agent-scope acquire since round
    5; olmc_kernels.h acquire_rows).  A stale 16-byte row at 1M paths would move the price by 0.25 sigma -- invisible to every 3-sigma
    gate.  So: the same pricing thousands of times (the headline 1M x 252: 3,907 rows in two levels; 8M x 252: 31,250 rows; the
    14-contract Greeks: 32-wide rows summed by a whole workgroup; 10k x 50: one level), while seven other threads keep seven other
    contexts of the device busy with launches of other sizes -- ONE distinct (sum, sumsq) each.  OLMC_REPRO_SCALE multiplies the counts
    (10 = the 20,000 pricings VERDICT r4 asked for)."""
    import threading
    scale = float(os.environ.get("OLMC_REPRO_SCALE", "1"))
    stop, errs = threading.Event(), []

    def load(k):
        try:
            sizes = [(10_000, 50), (100_000, 100), (300_001, 7), (65_537, 33), (1_000_000, 16), (257, 1), (2_000_000, 4)]
            N, M = sizes[k % len(sizes)]
            first = None
            while not stop.is_set():
                st = _hip.european(*ATM, 0.0, True, N, M, 1000 + k, True)
                first = first or (st.sum, st.sumsq)
                if (st.sum, st.sumsq) != first:
                    raise AssertionError(f"load thread {k}: other bits on a repeat")
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=load, args=(k,)) for k in range(7)]
    [t.start() for t in ts]
    try:
        for what, reps, fn in (
                ("1M x 252", int(2000 * scale), lambda: (lambda st: (st.sum, st.sumsq))(_hip.european(*ATM, 0.0, True, 1_000_000, 252, 42, True))),
                ("8M x 252", int(150 * scale), lambda: (lambda st: (st.sum, st.sumsq))(_hip.european(*ATM, 0.0, True, 8_000_000, 252, 42, True))),
                ("10k x 50", int(3000 * scale), lambda: (lambda st: (st.sum, st.sumsq))(_hip.european(*ATM, 0.0, True, 10_000, 50, 42, True))),
                ("greeks14 1M x 64", int(500 * scale), lambda: tuple(_hip.european_greeks_fd(*ATM, 0.0, True, 1_000_000, 64, 42, True, want_evals=False)[0]))):
            seen = {fn() for _ in range(reps)}
            assert len(seen) == 1, (what, len(seen), sorted(seen)[:3])
    finally:
        stop.set()
        [t.join() for t in ts]
    assert not errs, errs[:2]


def test_concurrent_calls_from_threads_are_serialised_correctly():
    import threading
    p = ol.MonteCarloPricer(50_000, 16, 7)
    want = {k: p.price(100.0 + k, 100.0, 1.0, 0.05, 0.2, "call", seed=k) for k in range(8)}
    wantg = p.greeks(*ATM, "call", include_second_order=False)
    got, errs = {}, []

    def work(k):
        try:
            for _ in range(20):
                got[k] = p.price(100.0 + k, 100.0, 1.0, 0.05, 0.2, "call", seed=k)
                assert p.greeks(*ATM, "call", include_second_order=False) == wantg
                assert ol.AsianOption(*ATM, seed=k).price(2000, 8) > 0
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(k,)) for k in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs and got == want


def test_shutdown_and_reinitialise():
    a = ol.MonteCarloPricer(10_000, 8, 3).price(*ATM, "put")
    _hip.shutdown()
    assert ol.MonteCarloPricer(10_000, 8, 3).price(*ATM, "put") == a


# ------------------------------------------------------------------ unvalidated inputs behave like the reference
def test_invalid_inputs_propagate_nan_like_numpy():
    """monte_carlo.py validates nothing at call time: S < 0 -> np.log -> nan -> np.maximum propagates -> nan price."""
    p = ol.MonteCarloPricer(1000, 4, 1)
    assert math.isnan(p.price(-100.0, 100.0, 1.0, 0.05, 0.2, "call"))
    res = p.price(100.0, float("nan"), 1.0, 0.05, 0.2, "put", return_error=True)
    assert math.isnan(res.price) and math.isnan(res.std_error) and res.n_paths == 2000
    assert np.isnan(p._simulate(-1.0, 1.0, 0.05, 0.2, 0.0)).all()
    # S = 0 is not NaN in NumPy either: ln 0 = -inf, S_T = 0 -> call 0, put K e^{-rT}
    assert p.price(0.0, 100.0, 1.0, 0.05, 0.2, "call") == 0.0
    assert p.price(0.0, 100.0, 1.0, 0.05, 0.2, "put") == pytest.approx(100.0 * math.exp(-0.05), rel=1e-12)
    # sigma < 0 only flips the antithetic legs
    assert p.price(*ATM[:4], -0.2, "call") == pytest.approx(p.price(*ATM, "call"), rel=1e-12)
    assert math.isnan(ol.AsianOption(-1.0, 100.0, 1.0, 0.05, 0.2, seed=1).price(100, 4))
    with pytest.raises(ol.GreeksError):       # NaN greeks are still returned by the reference; a GreeksError only on exceptions
        ol.compute_greeks_unified(ol.MonteCarloPricer(100, 0, 1), *ATM)


# ------------------------------------------------------------------ RNG stream quality at scale
def test_normal_stream_quality_large_sample():
    from scipy import stats
    z = _hip.normals(2024, 0, 40_000, 252).astype(np.float64)          # 10.08M normals
    flat = z.ravel()
    n = flat.size
    m = [np.mean(flat ** k) for k in range(1, 9)]
    want = [0, 1, 0, 3, 0, 15, 0, 105]
    sd = [1, math.sqrt(2), math.sqrt(15), math.sqrt(96), math.sqrt(945), math.sqrt(10170), math.sqrt(135135), math.sqrt(2016000)]
    for k in range(8):
        assert abs(m[k] - want[k]) < 5 * sd[k] / math.sqrt(n), (k + 1, m[k])
    assert stats.kstest(flat[:2_000_000], "norm").pvalue > 1e-4
    # the sum over steps is what the European payoff consumes: N(0, 252)
    rows = z.sum(axis=1)
    assert abs(rows.mean()) < 5 * math.sqrt(252 / 40_000) and abs(rows.var() / 252 - 1) < 5 * math.sqrt(2 / 40_000)
    assert stats.kstest(rows / math.sqrt(252), "norm").pvalue > 1e-4
    # serial structure: lag-1..4 autocorrelation along steps, and across neighbouring paths
    for lag in (1, 2, 3, 4):
        c = np.mean(z[:, lag:] * z[:, :-lag])
        assert abs(c) < 5 / math.sqrt(z[:, lag:].size), (lag, c)
    assert abs(np.mean(z[1:] * z[:-1])) < 5 / math.sqrt(z[1:].size)
    # cos/sin legs of one Box-Muller pair are independent: chi-square on a 16x16 grid of their CDF values
    u0, u1 = stats.norm.cdf(z[:, 0::4].ravel()), stats.norm.cdf(z[:, 1::4].ravel())
    h, _, _ = np.histogram2d(u0, u1, bins=16, range=[[0, 1], [0, 1]])
    chi2 = ((h - u0.size / 256) ** 2 / (u0.size / 256)).sum()
    assert stats.chi2.sf(chi2, 255) > 1e-4, chi2
    # tail: P(|z| > 4) = 6.33e-5
    tail = np.mean(np.abs(flat) > 4.0)
    assert abs(tail - 6.334e-5) < 5 * math.sqrt(6.334e-5 / n)


# ------------------------------------------------------------------ one-process-per-GPU entry (torch = plumbing)
def test_sharded_price_through_process_group_single_rank():
    import socket

    import torch
    import torch.distributed as dist

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        price, se, n = ol.sharding.price_european_sharded(*ATM, "call", 0.0, 300_000, 21, 11)
        whole = _hip.european(*ATM, 0.0, True, 300_000, 21, 11)
        assert (price, n) == (whole.price, whole.n) and se == pytest.approx(whole.std_error, rel=1e-12)
        # on a non-default torch stream too: the kernel is ordered on the stream it is given
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            buf = torch.empty(3, dtype=torch.float64, device="cuda")
            p2, _, _ = ol.sharding.price_european_sharded(*ATM, "call", 0.0, 300_000, 21, 11, device_buffer=buf)
        assert p2 == price
        # Greeks, control variate and a path-dependent payoff through the same process group (SURVEY §8e payloads)
        pricer = ol.MonteCarloPricer(300_000, 21, 11)
        fused = pricer.greeks(*ATM, "call")
        sharded = ol.sharding.greeks_sharded(*ATM, "call", 0.0, 300_000, 21, 11)
        assert list(sharded) == list(fused)
        for k in fused:
            assert sharded[k] == pytest.approx(fused[k], rel=1e-9, abs=1e-9), k
        assert ol.sharding.control_variate_sharded(*ATM, "call", 0.0, 300_000, 21, 11) == pytest.approx(
            pricer.price_with_control_variate(*ATM, "call"), rel=1e-12)
        ap, ase, an = ol.sharding.price_sharded(lambda lo, n: _hip.asian(*ATM, 0.0, True, False, n, 21, 11, False, path_offset=lo),
                                                300_000, ATM[3], ATM[2])
        want = _hip.asian(*ATM, 0.0, True, False, 300_000, 21, 11, False)
        assert (ap, an) == (want.price, want.n) and ase == pytest.approx(want.std_error, rel=1e-12)
        qp, qse, qn = ol.sharding.qmc_sharded(*ATM, "call", 0.0, 1 << 15, 16, 42)          # Sobol points through the same helpers (device kernel)
        qwant = ol.MonteCarloPricer(1 << 15, 16, 42, ol.MCMethod.QMC).price(*ATM, "call", return_error=True)
        assert (qp, qn) == (qwant.price, qwant.n_paths) and qse == pytest.approx(qwant.std_error, rel=1e-12)
        qpricer = ol.MonteCarloPricer(1 << 15, 16, 42, ol.MCMethod.QMC)
        for second in (False, True):                                                    # Sobol Greeks and control variate through the same helpers
            got = ol.sharding.qmc_greeks_sharded(*ATM, "call", 0.0, 1 << 15, 16, 42, include_second_order=second)
            want = qpricer.greeks(*ATM, "call", include_second_order=second)
            assert list(got) == list(want)
            for key in want:
                assert got[key] == pytest.approx(want[key], rel=1e-9, abs=1e-9), key
        assert ol.sharding.qmc_control_variate_sharded(*ATM, "call", 0.0, 1 << 15, 16, 42) == pytest.approx(
            qpricer.price_with_control_variate(*ATM, "call"), rel=1e-12)
    finally:
        dist.destroy_process_group()


def test_control_variate_shards_combine_to_the_whole():
    whole = _hip.european_cv(*ATM, 0.01, True, 100_001, 13, 5, True)
    parts = [_hip.european_cv_shard(*ATM, 0.01, True, lo, hi - lo, 13, 5, True)
             for lo, hi in (ol.sharding.shard_bounds(100_001, k, 3) for k in range(3))]
    both = _hip.combine_cv(parts, ATM[0], ATM[2], ATM[3], 0.01)
    assert both.n == whole.n == 200_002
    for f in ("sum_d", "sum_s", "sum_dd", "sum_ss", "sum_ds", "value"):
        assert getattr(both, f) == pytest.approx(getattr(whole, f), rel=1e-12), f


# ------------------------------------------------------------------ large sizes (size-independent properties)
def test_large_path_count_grid_stride_and_long_paths():
    S, K, T, r, v = ATM
    # 2^28 paths: beyond 2^18 workgroups the kernel grid-strides; se ~ 6e-4, so this is a sharp accuracy check
    st = _hip.european(S, K, T, r, v, 0.0, True, 1 << 28, 8, 5)
    assert st.n == 1 << 29
    assert abs(st.price - BS_CALL) <= 3.5 * st.std_error and st.std_error < 1e-3
    halves = [_hip.european(S, K, T, r, v, 0.0, True, 1 << 27, 8, 5, True, path_offset=k << 27) for k in range(2)]
    comb = _hip.combine_stats([(h.sum, h.sumsq, h.n) for h in halves], r, T)
    assert comb.price == pytest.approx(st.price, rel=1e-12)
    # very long paths: 100,000 steps (25,000 Philox blocks per path)
    lp = _hip.european(S, K, T, r, v, 0.0, False, 20_000, 100_000, 3)
    assert abs(lp.price - ol.black_scholes(S, K, T, r, v, "put")) <= 3.5 * lp.std_error
    # the fused 14-contract Greeks at the 64M-path scale of config 5 on one GPU stay within tight bands of Black-Scholes
    g = ol.MonteCarloPricer(16_000_000, 16, 9).greeks(*ATM, "call", include_second_order=False)
    ex = orc.bs_greeks(*ATM, "call")
    assert g["delta"] == pytest.approx(ex["delta"], abs=2e-4) and g["vega"] == pytest.approx(ex["vega"], rel=2e-3)
    assert g["rho"] == pytest.approx(ex["rho"], rel=2e-3) and g["gamma"] == pytest.approx(ex["gamma"], abs=2e-4)


def test_monte_carlo_error_shrinks_like_one_over_sqrt_n():
    """The harness of src/pricing_models/validation.py:202-239 applied to the device pricer: the spread of
    independent estimates must fall like 1/sqrt(N) and bracket Black-Scholes."""
    stds, means = [], []
    for N in (10_000, 40_000, 160_000, 640_000):
        prices = [ol.MonteCarloPricer(N, 16, seed).price(*ATM, "call") for seed in range(100, 148)]
        stds.append(float(np.std(prices)))
        means.append(float(np.mean(prices)))
    for a, b in zip(stds, stds[1:]):
        assert 1.45 < a / b < 2.75, stds                      # expected 2.0; 48 trials => ~10 % noise on each std
    for m, s in zip(means, stds):
        assert abs(m - BS_CALL) <= 4 * s / math.sqrt(48)
    # the reported (naive, reference-formula) std_error overstates the true spread of the antithetic estimator
    rep = ol.MonteCarloPricer(640_000, 16, 1).price(*ATM, "call", return_error=True).std_error
    assert 0.4 * rep < stds[-1] < 1.05 * rep


# ------------------------------------------------------------------ full paths (simulate_gbm_paths counterpart)
def test_full_paths_against_checker_and_reference_shape(golden):
    g = golden["full_paths"]
    S, T, r, v, q, N, M, seed = g["args"]
    fp = ol.simulate_gbm_paths_hip(S, T, r, v, q, N, M, seed)
    assert list(fp.shape) == g["shape"] and fp.flags["C_CONTIGUOUS"] == g["c_contiguous"] and fp.dtype == np.float64
    assert np.all(fp[:, 0] == S)                                                # column 0 is the spot (gbm_numpy.py:115)
    want = po.gbm_paths(S, T, r, v, q, N, M, seed).T
    assert np.allclose(fp, want, rtol=2e-6, atol=0)
    # statistically the reference's paths: E[S_t] = S e^{(r-q)t}, column means within 4 standard errors of the golden's
    t = np.arange(M + 1) * (T / M)
    se = fp.std(axis=0) / math.sqrt(N)
    assert np.all(np.abs(fp.mean(axis=0) - S * np.exp((r - q) * t)) <= 4 * se + 1e-12)
    assert np.all(np.abs(fp.mean(axis=0) - np.array(g["col_mean"])) <= 4 * math.sqrt(2) * se + 1e-12)
    # a path's terminal value is what the terminal-only backend returns for the same stream (non-antithetic leg)
    term = ol.simulate_gbm_hip(S, T, r, v, q, N, M, seed, antithetic=False)
    assert np.allclose(fp[:, -1], term, rtol=1e-6)
    odd = ol.simulate_gbm_paths_hip(S, T, r, v, q, 257, 7, 3)                  # ragged sizes, step remainder
    assert odd.shape == (257, 8) and np.allclose(odd, po.gbm_paths(S, T, r, v, q, 257, 7, 3).T, rtol=2e-6)


def test_repeated_calls_do_not_leak_device_memory():
    """Scratch is grown lazily and reused: hundreds of pricings of every kind leave the device's free memory where it was."""
    import torch

    def spin(n):
        for i in range(n):
            _hip.european(*ATM, 0.0, True, 50_000 + (i % 7) * 1000, 16, i)
            if i % 10 == 0:
                _hip.asian(*ATM, 0.0, True, False, 20_000, 16, i)
                _hip.heston(100.0, 100.0, 1.0, 0.05, 0.0, True, 2.0, 0.04, 0.3, -0.7, 0.04, 20_000, 16, i)
                _hip.american_lsm(*ATM, 0.0, False, 5_000, 10, 3, i)
                _hip.european_terminal(100.0, 1.0, 0.05, 0.2, 0.0, 30_000, 8, i)
                _hip.gbm_paths(100.0, 1.0, 0.05, 0.2, 0.0, 2_000, 8, i, path_major=True)
                ol.MonteCarloPricer(30_000, 8, i).greeks(*ATM, "call")

    spin(50)                                   # every scratch buffer reaches its working size
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    spin(600)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 8 << 20, (free0, free1)


def test_qmc_eight_point_blocks_give_the_same_points():
    """european_qmc_block_kernel (eight consecutive Sobol points per thread, Gray-code increments) against the one-point
    kernel: identical terminal prices bit for bit at ragged offsets and sizes, prices equal to reduction-order rounding.  Round 5: at
    point offsets that are multiples of 512 the ALIGNED form runs (Gray bits 9 .. 29 of a wave's 64 blocks are uniform: their direction
    numbers fold into one word per wave and dimension, lane l serving dimension c0 + l: a wave stays whole while its first lane has a
    block) -- offsets 0, 512, 1 << 21, ragged counts, waves with one live lane and quarters longer than 64 dimensions included."""
    try:
        for N, M, off in ((1, 1, 0), (7, 3, 0), (8, 5, 8), (9, 2, 7), (1000, 16, 3), (4097, 64, 12345), (70_000, 7, 1),
                          (513, 5, 512), (4097, 64, 1 << 21), (70_001, 17, 1024), (300_000, 3, 0), (1, 9, 0), (4097, 300, 512), (520, 260, 0)):
            tables = ol.monte_carlo.sobol_tables(M, 11)
            _hip.tune(_hip.TUNE_QMC_BLOCK, -1)
            one = _hip.european_qmc_terminal(100.0, 1.0, 0.05, 0.2, 0.01, N, *tables, point_offset=off)
            mir = _hip.european_qmc_terminal(100.0, 1.0, 0.05, 0.2, 0.01, N, *tables, point_offset=off, antithetic=True)
            _hip.tune(_hip.TUNE_QMC_BLOCK, 1)
            assert np.array_equal(_hip.european_qmc_terminal(100.0, 1.0, 0.05, 0.2, 0.01, N, *tables, point_offset=off), one), (N, M, off)
            assert np.array_equal(_hip.european_qmc_terminal(100.0, 1.0, 0.05, 0.2, 0.01, N, *tables, point_offset=off, antithetic=True), mir)
        for N, M in ((2**14, 16), (100_003, 33)):
            _hip.tune(_hip.TUNE_QMC_BLOCK, -1)
            a = ol.MonteCarloPricer(N, M, 42, ol.MCMethod.QMC).price(*ATM, "call", return_error=True)
            _hip.tune(_hip.TUNE_QMC_BLOCK, 1)
            b = ol.MonteCarloPricer(N, M, 42, ol.MCMethod.QMC).price(*ATM, "call", return_error=True)
            assert b.price == pytest.approx(a.price, rel=1e-13) and b.std_error == pytest.approx(a.std_error, rel=1e-10) and b.n_paths == a.n_paths
        _hip.tune(_hip.TUNE_QMC_BLOCK, 0)
        big = ol.MonteCarloPricer(2**21, 8, 42, ol.MCMethod.QMC).price(*ATM, "call", return_error=True)     # a size that switches by itself (2^19 below 32 dimensions ... 2^22 from 128)
        assert abs(big.price - BS_CALL) < 2e-3
    finally:
        _hip.tune(_hip.TUNE_QMC_BLOCK, 0)


def test_qmc_split_workgroups_return_the_bits_of_one_point_threads():
    """Launches of fewer than 2^22 Sobol points (2^21 below 128 dimensions, 2^20 below 64, 2^19 below 32; round 4: up to 2^18) with >= 16 dimensions give 64 points to a workgroup and a quarter
    of the dimensions to each of its four waves (european_qmc_kernel<., true>, european_qmc_batch_kernel<., false, true>).  Every Sobol
    kernel adds a point's inverse normals in the same association (quarters), so the split form must return the one-point form's
    terminal prices bit for bit -- ragged point counts around the 64-point workgroup, dimension counts that do not divide by four,
    point offsets -- and the same sums to reduction-order rounding, for the price, the control variate and the fused Greeks.
    Round 5: where the point offset is a multiple of 64 and there are >= 32 dimensions, the ALIGNED form runs (the direction numbers of
    Gray bits 6 .. 29 folded lane-per-dimension for 64 dimensions at a time and broadcast by ds_bpermute, two dimensions in lockstep,
    the inverse normal's coefficients in registers): the cases with offsets 0, 64, 128, 640 and 1 << 20 and 32+ dimensions -- ragged
    last workgroups (dead lanes keep their index), odd quarters, quarters longer than 64 dimensions (a second chunk) and a launch
    above 2^20 points included -- hold it to the same bits."""
    S, K, T, r, v = ATM
    try:
        for N, M, off in ((1, 16, 0), (63, 17, 0), (64, 18, 5), (65, 19, 64), (1000, 33, 3), (4097, 252, 12345), (70_001, 63, 1), (1 << 18, 16, 0),
                          (1, 64, 0), (1000, 64, 0), (4097, 252, 0), (70_001, 65, 640), (300_000, 64, 64), ((1 << 19) + 17, 70, 1 << 20), (65, 100, 63),
                          (1000, 32, 0), (70_001, 41, 128), (4097, 300, 64), (130, 1021, 0), ((1 << 20) + 3, 130, 0), (1000, 260, 0), (77, 257, 64)):
            tables = ol.monte_carlo.sobol_tables(M, 11)
            _hip.tune(_hip.TUNE_QMC_BLOCK, -1)                       # one point per thread, never split
            one = _hip.european_qmc_terminal(S, T, r, v, 0.01, N, *tables, point_offset=off)
            mir = _hip.european_qmc_terminal(S, T, r, v, 0.01, N, *tables, point_offset=off, antithetic=True)
            p1 = _hip.european_qmc(S, K, T, r, v, 0.01, True, N, *tables, point_offset=off)
            c1 = _hip.european_qmc_cv(S, K, T, r, v, 0.01, False, N, *tables, point_offset=off)
            g1, e1 = _hip.european_qmc_greeks_fd(S, K, T, r, v, 0.01, True, N, *tables, True) if off == 0 else (None, None)
            _hip.tune(_hip.TUNE_QMC_BLOCK, 0)                        # the default shape: split at these sizes
            assert np.array_equal(_hip.european_qmc_terminal(S, T, r, v, 0.01, N, *tables, point_offset=off), one), (N, M, off)
            assert np.array_equal(_hip.european_qmc_terminal(S, T, r, v, 0.01, N, *tables, point_offset=off, antithetic=True), mir), (N, M, off)
            p2 = _hip.european_qmc(S, K, T, r, v, 0.01, True, N, *tables, point_offset=off)
            assert p2.n == p1.n == N and p2.sum == pytest.approx(p1.sum, rel=1e-13) and p2.sumsq == pytest.approx(p1.sumsq, rel=1e-13)
            c2 = _hip.european_qmc_cv(S, K, T, r, v, 0.01, False, N, *tables, point_offset=off)
            for f in ("sum_d", "sum_s", "sum_dd", "sum_ss", "sum_ds"):
                assert getattr(c2, f) == pytest.approx(getattr(c1, f), rel=1e-13), (N, M, f)
            if off == 0:
                g2, e2 = _hip.european_qmc_greeks_fd(S, K, T, r, v, 0.01, True, N, *tables, True)
                for a, b in zip(e1, e2):
                    assert b.sum == pytest.approx(a.sum, rel=1e-13, abs=1e-300) and b.n == a.n
        # below 16 dimensions nothing is split: the same launches, the same bits in every output
        tables = ol.monte_carlo.sobol_tables(15, 3)
        _hip.tune(_hip.TUNE_QMC_BLOCK, -1)
        a = _hip.european_qmc(S, K, T, r, v, 0.0, True, 5000, *tables)
        _hip.tune(_hip.TUNE_QMC_BLOCK, 0)
        b = _hip.european_qmc(S, K, T, r, v, 0.0, True, 5000, *tables)
        assert (a.sum, a.sumsq) == (b.sum, b.sumsq)
    finally:
        _hip.tune(_hip.TUNE_QMC_BLOCK, 0)


def test_fetch_dev_hands_over_what_earlier_work_on_the_stream_left():
    """olmc_fetch_dev: the blocking hand-over of a shard's (all-reduced) triple from a device buffer -- queued behind the
    caller's work on the caller's stream, polled like a blocking pricing.  Checked with a torch stream and buffer as in
    bench.py / sharding.price_european_sharded, against the same triple fetched by a plain D2H copy, polled and not."""
    torch = pytest.importorskip("torch")
    S, K, T, r, v = ATM
    st = torch.cuda.Stream()
    buf = torch.zeros(3, dtype=torch.float64, device="cuda")
    for poll in (0, -1):
        _hip.tune(_hip.TUNE_POLL, poll)
        try:
            with torch.cuda.stream(st):
                for k in range(4):
                    _hip.european_shard_dev(S, K, T, r, v, 0.0, True, 1000 * k, 300_000, 32, 5 + k, True, buf.data_ptr(), st.cuda_stream)
                    got = _hip.fetch_dev(buf.data_ptr(), 3, st.cuda_stream)
                    assert got == buf.cpu().tolist() and got[2] == 600_000.0
                    want = _hip.european(S, K, T, r, v, 0.0, True, 300_000, 32, 5 + k, True, path_offset=1000 * k)
                    assert (got[0], got[1]) == (want.sum, want.sumsq)
        finally:
            _hip.tune(_hip.TUNE_POLL, 0)
    with pytest.raises(ol.AccelerationError):
        _hip.fetch_dev(buf.data_ptr(), 64, st.cuda_stream)


def test_caller_streams_get_their_own_workspace_and_more_streams_than_slots_share_safely():
    """Launches on caller streams (olmc_european_shard_dev) take a reduction workspace per stream -- no event needed, the stream
    orders its own launches -- for up to eight streams; a ninth and later streams share slots, event-guarded.  Twelve streams
    (the NULL stream among them) with several launches in flight each must all deliver the blocking call's triple."""
    torch = pytest.importorskip("torch")
    S, K, T, r, v = ATM
    streams = [torch.cuda.Stream() for _ in range(11)]
    ptrs = [st.cuda_stream for st in streams] + [0]                  # 0 = the NULL stream
    rounds, n = 5, 200_000
    out = torch.zeros((rounds, len(ptrs), 3), dtype=torch.float64, device="cuda")
    for k in range(rounds):
        for j, ptr in enumerate(ptrs):
            _hip.european_shard_dev(S, K, T, r, v, 0.0, True, 0, n + 257 * j, 48, 100 * k + j, True, out[k, j].data_ptr(), ptr)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for k in range(rounds):
        for j in range(len(ptrs)):
            want = _hip.european(S, K, T, r, v, 0.0, True, n + 257 * j, 48, 100 * k + j, True)
            assert (got[k, j, 0], got[k, j, 1], got[k, j, 2]) == (want.sum, want.sumsq, float(want.n)), (k, j)
