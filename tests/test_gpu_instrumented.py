"""GPU tests that need the INSTRUMENTED build (include/olmc_probe.h -> tools/probe/libolmc_probe.so): the product's own translation
unit compiled with its test seams in, plus the validation taps libolmc.so no longer carries.  `probe.hip` is the product's ctypes
binding bound to that library, so `probe.hip.european(...)` runs the same code `_hip.european(...)` runs in libolmc.so.

  * the multi-GPU engine REHEARSED with several ranks on the one GPU of the box (hand-over order, drain of ranks >= 1,
    restoration of the thread's device, every payload: price 3, Greeks 17 / 33, control variate 6 doubles);
  * the error paths behind fault injection (failing rank, device-side row-capacity guard);
  * the exp2, inverse-normal and normal-moment taps.
"""
import math

import numpy as np
import pytest

import optionslab_amd as ol
from optionslab_amd import _hip
from tools.probe import binding as probe

pytestmark = pytest.mark.gpu

ATM = (100.0, 100.0, 1.0, 0.05, 0.2)
hip = probe.hip


@pytest.fixture(scope="module", autouse=True)
def _device():
    assert hip.device_info()["arch"].startswith("gfx950")
    yield
    for knob in (probe.TUNE_FAULT_SHARD, probe.TUNE_FORCE_NV, probe.TUNE_MULTI_REHEARSAL):
        probe.tune(knob, 0)
    hip.shutdown()


@pytest.fixture
def rehearsal():
    probe.tune(probe.TUNE_MULTI_REHEARSAL, 1)
    yield
    probe.tune(probe.TUNE_MULTI_REHEARSAL, 0)


def test_the_instrumented_build_prices_what_the_product_prices():
    a = hip.european(*ATM, 0.0, True, 300_000, 52, 42, True)
    b = _hip.european(*ATM, 0.0, True, 300_000, 52, 42, True)
    assert (a.sum, a.sumsq, a.n, a.price, a.std_error) == (b.sum, b.sumsq, b.n, b.price, b.std_error)


# ------------------------------------------------------------------ multi-GPU engine, n ranks rehearsed on one device
@pytest.mark.parametrize("n_ranks", [2, 3, 8, 11])
def test_n_rank_price_equals_the_rank_ordered_sum_of_its_shards(rehearsal, n_ranks):
    """olmc_multi_gpu_european with n ranks (each its own stream and buffers on this one device, the all-reduce emulated by a
    rank-ordered sum behind every rank's kernel): the triple must be EXACTLY what olmc_combine_stats makes of the n shard triples
    priced one by one -- same partition [d N / P, (d + 1) N / P), same paths, same association -- and the call itself checks that
    ranks 1..n-1 ended with rank 0's bits.  11 ranks share the 8 workspace slots a context has for caller streams."""
    S, K, T, r, v = ATM
    N, M, seed = 1_000_003, 40, 9
    got = hip.multi_gpu_european(S, K, T, r, v, 0.0, True, N, M, seed, True, n_ranks)
    parts = []
    for d in range(n_ranks):
        lo, hi = N * d // n_ranks, N * (d + 1) // n_ranks
        st = hip.european(S, K, T, r, v, 0.0, True, hi - lo, M, seed, True, path_offset=lo)
        parts.append((st.sum, st.sumsq, st.n))
    want = hip.combine_stats(parts, r, T)
    assert (got.sum, got.sumsq, got.n, got.price, got.std_error) == (want.sum, want.sumsq, want.n, want.price, want.std_error)
    whole = hip.european(S, K, T, r, v, 0.0, True, N, M, seed, True)
    assert got.n == whole.n == 2 * N and got.sum == pytest.approx(whole.sum, rel=1e-13) and got.sumsq == pytest.approx(whole.sumsq, rel=1e-13)
    assert hip.device_info()["device"] == 0                      # the thread's library device came back


@pytest.mark.parametrize("second", [False, True])
@pytest.mark.parametrize("n_ranks", [1, 4])
def test_n_rank_greeks_equal_the_one_device_greeks(rehearsal, n_ranks, second):
    """olmc_multi_gpu_greeks_fd: the 8 / 14 bumped contracts on the same normals, one launch per rank, ONE all-reduce of 17 / 33
    doubles (count 2k + 1).  Against olmc_european_greeks_fd on one device: same paths, another association of the
    sums -- 1e-12 x price x the finite-difference amplification on every Greek, and every evaluation's own sums to 1e-13."""
    N, M, seed = 400_000, 20, 42
    one, evals1 = hip.european_greeks_fd(*ATM, 0.0, True, N, M, seed, second, want_evals=True)
    many, evalsn = hip.multi_gpu_greeks_fd(*ATM, 0.0, True, N, M, seed, second, n_ranks, want_evals=True)
    k = 14 if second else 8
    for a, b in zip(evals1[:k], evalsn[:k]):
        assert b.n == a.n == 2 * N
        assert b.sum == pytest.approx(a.sum, rel=1e-13) and b.sumsq == pytest.approx(a.sumsq, rel=1e-13)
        assert b.price == pytest.approx(a.price, rel=1e-13) and b.std_error == pytest.approx(a.std_error, rel=1e-9)
    h_s, h_v, h_r, h_t = 1.0, 0.01, 1e-4, 1 / 365.0
    amp = [1, 1 / h_s, 4 / h_s**2, 1 / h_v, 2 / h_t, 1 / h_r, 1 / (h_s * h_v), 2 / (h_s * h_t), 4 / h_v**2]
    for i in range(9 if second else 6):
        assert many[i] == pytest.approx(one[i], abs=1e-12 * one[0] * amp[i] + 1e-13), i
    lean, none = hip.multi_gpu_greeks_fd(*ATM, 0.0, True, N, M, seed, second, n_ranks, want_evals=False)
    assert none == [] and lean[:6] == many[:6]


@pytest.mark.parametrize("n_ranks", [1, 3])
def test_n_rank_control_variate_equals_the_combined_shards(rehearsal, n_ranks):
    """olmc_multi_gpu_european_cv: five moments + n in one all-reduce (count 6) against olmc_combine_cv of the shard moments."""
    S, K, T, r, v = ATM
    N, M, seed = 300_001, 12, 5
    got = hip.multi_gpu_european_cv(S, K, T, r, v, 0.01, False, N, M, seed, True, n_ranks)
    parts = []
    for d in range(n_ranks):
        lo, hi = N * d // n_ranks, N * (d + 1) // n_ranks
        parts.append(hip.european_cv_shard(S, K, T, r, v, 0.01, False, lo, hi - lo, M, seed, True))
    want = hip.combine_cv(parts, S, T, r, 0.01)
    assert got.n == want.n == 2 * N
    for f in ("sum_d", "sum_s", "sum_dd", "sum_ss", "sum_ds", "value"):
        assert getattr(got, f) == pytest.approx(getattr(want, f), rel=1e-13), f
    whole = hip.european_cv(S, K, T, r, v, 0.01, False, N, M, seed, True)
    assert got.value == pytest.approx(whole.value, rel=1e-11)


def test_the_pricer_with_n_gpus_prices_what_the_one_gpu_pricer_prices(rehearsal, monkeypatch):
    """MonteCarloPricer(..., n_gpus=P): price(), greeks() and price_with_control_variate() through the single-process engine.  The
    pricer is pointed at the instrumented build (same code) so that P = 5 ranks can be rehearsed on this one device: same paths as
    the one-GPU pricer, sums in another association."""
    monkeypatch.setattr(ol.monte_carlo, "_hip", hip)
    one = ol.MonteCarloPricer(300_000, 40, 42)
    many = ol.MonteCarloPricer(300_000, 40, 42, n_gpus=5)
    a, b = one.price(*ATM, "call", return_error=True), many.price(*ATM, "call", return_error=True)
    assert b.n_paths == a.n_paths == 600_000 and b.price == pytest.approx(a.price, rel=1e-13) and b.std_error == pytest.approx(a.std_error, rel=1e-10)
    assert many.price(*ATM, "put", q=0.01, seed=7) == pytest.approx(one.price(*ATM, "put", q=0.01, seed=7), rel=1e-13)
    assert many.price_with_control_variate(*ATM, "call") == pytest.approx(one.price_with_control_variate(*ATM, "call"), rel=1e-11)
    for second in (False, True):
        g1, g5 = one.greeks(*ATM, "call", include_second_order=second), many.greeks(*ATM, "call", include_second_order=second)
        assert list(g1) == list(g5)
        for k in g1:
            assert g5[k] == pytest.approx(g1[k], rel=1e-7, abs=1e-7), k
        assert ol.compute_greeks_unified(many, *ATM, "call", include_second_order=second) == g5


@pytest.mark.parametrize("n_ranks", [2, 8])
def test_launcher_threads_and_the_serial_form_give_the_same_bits(rehearsal, n_ranks):
    """Round 5: each rank's kernel is queued by a launcher thread bound to the rank's device (all at once); OLMC_TUNE_MULTI_LAUNCH = -1
    keeps round 4's form (the calling thread queues them one after the other).  Same partition, same kernels, same rank-ordered
    collective: every payload must come back bit for bit the same, and the spans of the call are reported."""
    S, K, T, r, v = ATM
    N, M, seed = 700_001, 24, 11
    sv, shift = ol.monte_carlo.sobol_tables(16, 42, 1 << 18)
    def everything():
        a = hip.multi_gpu_european(S, K, T, r, v, 0.0, True, N, M, seed, True, n_ranks)
        spans = hip.multi_gpu_spans()
        g, ev = hip.multi_gpu_greeks_fd(S, K, T, r, v, 0.0, True, N, M, seed, True, n_ranks, want_evals=True)
        c = hip.multi_gpu_european_cv(S, K, T, r, v, 0.01, False, N, M, seed, True, n_ranks)
        q = hip.multi_gpu_european_qmc(S, K, T, r, v, 0.0, True, 1 << 18, sv, shift, n_ranks)
        return ((a.sum, a.sumsq, a.n), tuple(g), tuple((e.sum, e.sumsq) for e in ev), (c.sum_d, c.sum_s, c.sum_dd, c.sum_ss, c.sum_ds, c.value),
                (q.sum, q.sumsq, q.n)), spans
    threaded, spans_t = everything()
    hip.tune(hip.TUNE_MULTI_LAUNCH, -1)
    try:
        serial, spans_s = everything()
    finally:
        hip.tune(hip.TUNE_MULTI_LAUNCH, 0)
    assert threaded == serial
    again, _ = everything()
    assert again == threaded
    for spans in (spans_t, spans_s):
        assert 0.0 < spans["launch_us"] < 5e4 and spans["total_us"] >= spans["launch_us"] + spans["collective_us"]


@pytest.mark.parametrize("n_ranks", [1, 2, 5, 8])
def test_n_rank_sobol_price_is_the_one_device_sobol_price(rehearsal, n_ranks):
    """olmc_multi_gpu_european_qmc (gbm_qmc.py:14-46 over several devices): rank d prices POINTS [d N / P, (d + 1) N / P) of the one
    scrambled sequence (boundaries on multiples of 512: qmc_shard_bounds, every rank runs the aligned kernels) through the kernels' point offset -- the same points as the one-device call, the sums in another association
    (1e-13), and exactly the rank-ordered sum of the shards priced one by one through olmc_european_qmc(point_offset=...)."""
    S, K, T, r, v = ATM
    for n, dims in ((1 << 14, 16), (300_001, 64), (1 << 21, 16)):            # split workgroups (plain, aligned) / eight points per thread
        sv, shift = ol.monte_carlo.sobol_tables(dims, 42, n)
        whole = hip.european_qmc(S, K, T, r, v, 0.0, True, n, sv, shift)
        got = hip.multi_gpu_european_qmc(S, K, T, r, v, 0.0, True, n, sv, shift, n_ranks)
        assert got.n == whole.n == n
        assert got.sum == pytest.approx(whole.sum, rel=1e-13) and got.sumsq == pytest.approx(whole.sumsq, rel=1e-13)
        assert got.price == pytest.approx(whole.price, rel=1e-13)
        parts = []
        for d in range(n_ranks):
            lo, hi = ol.sharding.qmc_shard_bounds(n, d, n_ranks)                    # inner boundaries on multiples of 512 points
            st = hip.european_qmc(S, K, T, r, v, 0.0, True, hi - lo, sv, shift, point_offset=lo)
            parts.append((st.sum, st.sumsq, st.n))
        want = hip.combine_stats(parts, r, T)
        assert (got.sum, got.sumsq, got.n, got.price, got.std_error) == (want.sum, want.sumsq, want.n, want.price, want.std_error)


@pytest.mark.parametrize("n_ranks", [1, 3, 8])
def test_n_rank_sobol_greeks_and_control_variate_are_the_one_device_ones(rehearsal, n_ranks):
    """olmc_multi_gpu_european_qmc_greeks_fd / _cv: every rank prices all 8 / 14 bumped contracts (the five control-variate moments)
    on ITS block of the Sobol points -- the same points as the one-device call, the sums in another association (1e-13), and the
    finite differences evaluated from them (differences of nearly equal prices: 1e-8 of the price scale) -- through split
    workgroups, their aligned form and eight points per thread."""
    S, K, T, r, v = ATM
    for n, dims in ((1 << 14, 16), (300_001, 64), (1 << 19, 16)):
        sv, shift = ol.monte_carlo.sobol_tables(dims, 42, n)
        for second in (False, True):
            g1, e1 = hip.european_qmc_greeks_fd(S, K, T, r, v, 0.01, True, n, sv, shift, second)
            gn, en = hip.multi_gpu_european_qmc_greeks_fd(S, K, T, r, v, 0.01, True, n, sv, shift, second, n_ranks)
            for a, b in zip(e1[:14 if second else 8], en):
                assert b.n == a.n == n and b.sum == pytest.approx(a.sum, rel=1e-13, abs=1e-300) and b.sumsq == pytest.approx(a.sumsq, rel=1e-13, abs=1e-300)
            assert gn[0] == pytest.approx(g1[0], rel=1e-13)
            for a, b in zip(g1[1:], gn[1:]):
                assert b == pytest.approx(a, rel=1e-7, abs=1e-7)
        c1 = hip.european_qmc_cv(S, K, T, r, v, 0.01, False, n, sv, shift)
        cn = hip.multi_gpu_european_qmc_cv(S, K, T, r, v, 0.01, False, n, sv, shift, n_ranks)
        assert cn.n == c1.n == n
        for f in ("sum_d", "sum_s", "sum_dd", "sum_ss", "sum_ds"):
            assert getattr(cn, f) == pytest.approx(getattr(c1, f), rel=1e-13), f
        assert cn.value == pytest.approx(c1.value, rel=1e-10)
        parts = []
        for d in range(n_ranks):                                     # the rank-ordered combination of the shards priced one by one (the discount
            lo, hi = ol.sharding.qmc_shard_bounds(n, d, n_ranks)    # applied per shard there, once to the sums here: an ulp)
            parts.append(hip.european_qmc_cv(S, K, T, r, v, 0.01, False, hi - lo, sv, shift, point_offset=lo))
        want = hip.combine_cv(parts, S, T, r, 0.01)
        for f in ("sum_d", "sum_s", "sum_dd", "sum_ss", "sum_ds"):
            assert getattr(cn, f) == pytest.approx(getattr(want, f), rel=4e-16), f
        assert cn.value == pytest.approx(want.value, rel=1e-11)


def test_the_qmc_pricer_with_n_gpus_prices_what_the_one_gpu_qmc_pricer_prices(rehearsal, monkeypatch):
    monkeypatch.setattr(ol.monte_carlo, "_hip", hip)
    one = ol.MonteCarloPricer(1 << 15, 32, 42, ol.MCMethod.QMC)
    many = ol.MonteCarloPricer(1 << 15, 32, 42, ol.MCMethod.QMC, n_gpus=3)
    a, b = one.price(*ATM, "call", return_error=True), many.price(*ATM, "call", return_error=True)
    assert b.n_paths == a.n_paths == 1 << 15 and b.price == pytest.approx(a.price, rel=1e-13) and b.std_error == pytest.approx(a.std_error, rel=1e-10)
    g1, gn = one.greeks(*ATM, "call"), many.greeks(*ATM, "call")
    assert list(g1) == list(gn) and all(gn[k] == pytest.approx(g1[k], rel=1e-7, abs=1e-7) for k in g1)
    assert many.price_with_control_variate(*ATM, "call") == pytest.approx(one.price_with_control_variate(*ATM, "call"), rel=1e-10)


def test_engines_come_and_go_without_touching_dead_streams(rehearsal):
    """The host fault of round 4 (gpurun_out/r04d: SIGSEGV inside olmc_multi_gpu_greeks_fd).  Cause: the engine was
    rebuilt whenever the rank count changed, which DESTROYED the old rank streams while the workspace slots they had claimed in the
    device's contexts still named them as owners; once all eight slots of a context belonged to dead streams, the next rank stream
    went down the slot-sharing path, which drained the slot's owner with hipStreamSynchronize(dead handle) -- a use-after-free inside
    the HIP runtime, not an error code.  Fixed in b0df7f4: slots are handed back before a rank stream dies (pool_forget_stream), and a
    shared slot's previous owner is drained with hipDeviceSynchronize, never through its handle.  This test walks that road on
    purpose: more rank counts than the engine cache keeps (engines are evicted and their streams destroyed), more ranks than a
    context has slots, twice over; every price must stay the one-device price."""
    S, K, T, r, v = ATM
    N, M, seed = 200_003, 12, 3
    whole = hip.european(S, K, T, r, v, 0.0, True, N, M, seed, True)
    for _ in range(2):
        for n_ranks in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 9, 2, 12):
            got = hip.multi_gpu_european(S, K, T, r, v, 0.0, True, N, M, seed, True, n_ranks)
            assert got.n == whole.n and got.sum == pytest.approx(whole.sum, rel=1e-13)
            g, _ = hip.multi_gpu_greeks_fd(S, K, T, r, v, 0.0, True, N, M, seed, bool(n_ranks & 1), n_ranks, want_evals=False)
            assert g[0] == pytest.approx(whole.price, rel=1e-12)
    assert hip.device_info()["device"] == 0


def test_a_failing_rank_in_the_middle_leaves_the_thread_and_the_library_usable(rehearsal):
    """Error returns of the multi-GPU calls go through a scope guard: the ranks already queued are drained, the thread's library
    device and HIP device restored.  Rank 2 of 4 is made to fail by the fault-injection knob: ranks 0 and 1 have kernels in flight."""
    S, K, T, r, v = ATM
    before = hip.european(S, K, T, r, v, 0.0, True, 50_000, 16, 5, True)
    for bad in (1, 3, 4):
        probe.tune(probe.TUNE_FAULT_SHARD, bad)
        try:
            with pytest.raises(ol.AccelerationError, match="injected shard failure"):
                hip.multi_gpu_european(S, K, T, r, v, 0.0, True, 4_000_000, 64, 5, True, 4)
            with pytest.raises(ol.AccelerationError, match="injected shard failure"):
                hip.multi_gpu_greeks_fd(S, K, T, r, v, 0.0, True, 400_000, 16, 5, True, 4)
        finally:
            probe.tune(probe.TUNE_FAULT_SHARD, 0)
        assert hip.device_info()["device"] == 0
        after = hip.european(S, K, T, r, v, 0.0, True, 50_000, 16, 5, True)
        assert (before.sum, before.sumsq) == (after.sum, after.sumsq)
    again = hip.multi_gpu_european(S, K, T, r, v, 0.0, True, 50_000, 16, 5, True, 1)
    assert (again.sum, again.sumsq) == (before.sum, before.sumsq)
    with pytest.raises(ol.AccelerationError):               # a bad shard argument (n_steps = 0) is refused before any launch
        hip.multi_gpu_european(S, K, T, r, v, 0.0, True, 1000, 0, 5, True, 2)
    with pytest.raises(ol.AccelerationError):               # more ranks than paths
        hip.multi_gpu_european(S, K, T, r, v, 0.0, True, 3, 4, 5, True, 4)
    assert hip.european(S, K, T, r, v, 0.0, True, 50_000, 16, 5, True).sum == before.sum


def test_one_rank_goes_through_rccl_in_the_instrumented_build_too():
    S, K, T, r, v = ATM
    a = hip.multi_gpu_european(S, K, T, r, v, 0.0, True, 200_000, 16, 5, True, 1)      # rehearsal off: the real all-reduce, 1 rank
    b = hip.european(S, K, T, r, v, 0.0, True, 200_000, 16, 5, True)
    assert (a.sum, a.sumsq, a.n, a.price, a.std_error) == (b.sum, b.sumsq, b.n, b.price, b.std_error)


# ------------------------------------------------------------------ device-side guard
def test_device_side_row_capacity_guard_answers_nan_and_keeps_the_library_usable():
    """A launch whose rows would not fit the workspace it was given (the r1 fault: NV = 4 rows on an NV = 2 workspace)
    must not store: olmc_kernels.h grid_reduce answers NaN instead.  The knob makes the workspace REPORT one value per
    row while really holding enough, so tripping the guard is safe."""
    S, K, T, r, v = ATM
    ok = hip.european(S, K, T, r, v, 0.0, True, 300_000, 8, 3, True)
    probe.tune(probe.TUNE_FORCE_NV, 1)
    try:
        bad = hip.european(S, K, T, r, v, 0.0, True, 300_000, 8, 3, True)
        assert math.isnan(bad.sum) and math.isnan(bad.price)
        bad4 = probe.normal_moments(3, 50_000, 8)             # NV = 4
        assert math.isnan(bad4[0])
        g, _ = hip.european_greeks_fd(*ATM, 0.0, True, 50_000, 8, 3, True, want_evals=False)     # NV = 16 through the workgroup-wide reduction
        assert math.isnan(g[0])
    finally:
        probe.tune(probe.TUNE_FORCE_NV, 0)
    again = hip.european(S, K, T, r, v, 0.0, True, 300_000, 8, 3, True)
    assert (again.sum, again.sumsq, again.n) == (ok.sum, ok.sumsq, ok.n)
    assert all(math.isfinite(m) for m in probe.normal_moments(3, 50_000, 8))


# ------------------------------------------------------------------ taps
@pytest.mark.parametrize("form", [None, 0, 1])
def test_device_exp2_f64_is_within_two_ulp_of_libm_everywhere(form):
    """The per-date exponential of the reference-precision Asian kernel (olmc_kernels.h), in both forms the sources carry:
    exp2_f64 (form 0: rint + degree-11 polynomial on [-1/2, 1/2] + v_ldexp_f64) and exp2_f64_tab (form 1: 256-entry table of
    2^(k/256) + degree-4 correction 1 + r q(r), q cubic; form None = whichever the kernel is built with, i.e. the table).  Against
    the host libm (numpy.exp2, < 1 ulp): <= 2 ulp on a dense sweep of the range a cumulative log-return can take, on the reduction's
    seams (half-integers and, for the table, the 128ths and 512ths where the table index rounds), and correct limits (0, inf, NaN)."""
    rng = np.random.default_rng(7)
    x = np.concatenate([rng.uniform(-60.0, 60.0, 400_000), rng.uniform(-1.0, 1.0, 400_000), rng.normal(0.0, 1e-3, 100_000),
                        np.arange(-80, 81) + 0.5, np.nextafter(np.arange(-80, 81) + 0.5, np.inf), np.nextafter(np.arange(-80, 81) + 0.5, -np.inf),
                        np.arange(-1000, 1001, 7.0), [0.0, -0.0, 1.0, -1.0, 1e-300, -1e-300, 1023.999, -1021.5],
                        (np.arange(-4096, 4097) + 0.5) / 64.0, np.nextafter((np.arange(-4096, 4097) + 0.5) / 64.0, np.inf),
                        (np.arange(-8192, 8193) + 0.5) / 256.0, np.nextafter((np.arange(-8192, 8193) + 0.5) / 256.0, -np.inf)])
    y, want = probe.exp2_probe(x, form), np.exp2(x)
    ulp = np.abs(y - want) / np.spacing(want)
    assert np.isfinite(y).all() and ulp.max() <= 2.0, (ulp.max(), x[np.argmax(ulp)])
    assert (y[np.isin(x, np.arange(-1000, 1001, 7.0))] == want[np.isin(x, np.arange(-1000, 1001, 7.0))]).all()     # exact powers of two
    special = probe.exp2_probe(np.array([np.nan, 1025.0, 5000.0, 1e300, -1100.0, -5000.0, -1e300, np.inf, -np.inf]), form)
    assert np.isnan(special[0]) and (special[1:4] == np.inf).all()          # overflow like exp2()
    assert (special[4:7] == 0.0).all()                                       # below the subnormal range: zero like exp2()
    # documented limit of the domain: an INFINITE argument answers NaN (inf - rint(inf)), where exp2() says inf / 0.  A
    # cumulative log-return is infinite only for infinite parameters, for which the reference's own path is NaN as well.
    assert np.isnan(special[7:]).all()


def test_the_sobol_kernels_inverse_normal_is_scipy_class_in_every_form():
    """ndtri_w and its three other forms (coefficients in registers, two in lockstep, eight in lockstep: olmc_kernels.h) through the
    instrumented build's tap, on Sobol-shaped probabilities k 2^-30: the clip's ends, the centre, both sides of the seam between
    the two fits (w = -ln 4p(1-p) = 6.25), the 2,000 smallest and largest uniforms and 200,000 random ones.

      * the four forms agree BIT FOR BIT (why a Sobol price does not depend on the launch shape);
      * against mpmath (40 digits) on 5,000 of them: within 2 x 2^-52 of the value (measured: 1.5, mean 0.34; before round 5's shorter
        arithmetic: 1.8, 0.49) -- SciPy's own ndtri, the reference's inverse normal (gbm_qmc.py:37), measures 1.56e-15 absolute on the
        same sample, ours 1.40e-15;
      * against SciPy on all of them: 2e-15 max(1, |z|)."""
    mp = pytest.importorskip("mpmath")
    from scipy.special import ndtri
    mp.mp.dps = 40
    rng = np.random.default_rng(7)
    seam = 0.5 * (1 - np.sqrt(1 - np.exp(-6.25)))
    edges = np.array([1e-10, 2.0 ** -30, 2.0 ** -29, 3 * 2.0 ** -30, 0.5, 0.5 - 2.0 ** -30, 0.5 + 2.0 ** -30, 1 - 2.0 ** -30, 1 - 2.0 ** -29, 0.25, 0.75])
    near = np.round((seam * (1 + np.linspace(-1e-3, 1e-3, 401))) * 2.0 ** 30) * 2.0 ** -30
    tails = np.arange(1, 2001) * 2.0 ** -30
    p = np.concatenate([edges, near, 1 - near, tails, 1 - tails, rng.integers(1, 1 << 30, size=200_000) * 2.0 ** -30])
    z = probe.ndtri_probe(p, 0)
    for form in (1, 2, 3):
        assert np.array_equal(z, probe.ndtri_probe(p, form)), form
    assert z[4] == 0.0 and np.array_equal(z[:4] < 0, np.ones(4, bool))
    want = ndtri(p)
    assert (np.abs(z - want) <= 2e-15 * np.maximum(1.0, np.abs(want))).all()
    fixed = edges.size + 2 * near.size
    sample = np.concatenate([np.arange(fixed), fixed + np.arange(0, 4000, 20), rng.integers(0, p.size, 4000)])
    worst = 0.0
    for i in sample:
        exact = mp.sqrt(2) * mp.erfinv(2 * mp.mpf(float(p[i])) - 1)
        if exact != 0:
            worst = max(worst, float(abs(mp.mpf(float(z[i])) - exact) / (abs(exact) * mp.mpf(2) ** -52)))
    assert worst <= 2.0, worst
    with pytest.raises(ol.AccelerationError):
        probe.ndtri_probe(np.array([0.0, 0.5]), 0)


def test_normal_moments_tap_agrees_with_the_normals_tap():
    z = hip.normals(5, 100, 3000, 7).astype(np.float64)
    assert np.array_equal(z, _hip.normals(5, 100, 3000, 7).astype(np.float64))          # both builds, one stream
    s1, s2, s3, s4 = probe.normal_moments(5, 3000, 7, path_offset=100)
    assert s1 == pytest.approx(z.sum(), rel=1e-5, abs=1e-3) and s2 == pytest.approx((z**2).sum(), rel=1e-6)
    assert s3 == pytest.approx((z**3).sum(), rel=1e-5, abs=1e-2) and s4 == pytest.approx((z**4).sum(), rel=1e-6)
    big = probe.normal_moments(9, 1 << 22, 64)              # 2.7e8 normals: variance to 9e-5
    n = (1 << 22) * 64
    assert abs(big[0] / n) < 5 / math.sqrt(n) and abs(big[1] / n - 1) < 5 * math.sqrt(2 / n) and abs(big[3] / n - 3) < 5 * math.sqrt(96 / n)


# ------------------------------------------------------------------ measurement entry points of the instrumented build
def test_fp64_normals_kernel_prices_the_same_option():
    """olmc_european_f64_normals (bench.py's c2_f64_normals): the European call with fp64 normals -- two per Philox block from 53-bit
    uniforms, library log / sincospi, fp64 sum.  Other normals than the product's (same counter stream, another mapping), so the two
    prices differ by sampling noise only; both within 3.5 sigma of Black-Scholes, equal standard errors to 1 %."""
    S, K, T, r, v = ATM
    bs = ol.black_scholes(S, K, T, r, v, "call")
    for N, M, seed in ((1_000_000, 252, 42), (300_000, 7, 3), (100_001, 1, 5)):
        a = probe.european_f64_normals(S, K, T, r, v, 0.0, True, N, M, seed)
        b = hip.european(S, K, T, r, v, 0.0, True, N, M, seed, True)
        assert a.n == b.n == 2 * N
        assert abs(a.price - bs) <= 3.5 * a.std_error and abs(b.price - bs) <= 3.5 * b.std_error
        assert a.std_error == pytest.approx(b.std_error, rel=0.01)
        assert abs(a.price - b.price) <= 4 * math.hypot(a.std_error, b.std_error)
        assert a.price != b.price


def test_launch_gap_probe_returns_a_few_microseconds():
    gap = probe.launch_gap_us(500)
    assert 0.5 < gap < 50.0, gap
