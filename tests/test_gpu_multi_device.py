"""The PRODUCT library's multi-GPU entry points on real devices (the multi-rank RCCL branch of multi_gpu_run had never
run with more than one rank; the rehearsed ranks of tests/test_gpu_instrumented.py replace the collective by a kernel, in another
.so).  On a box with N >= 2 visible GPUs these tests compare olmc_multi_gpu_european / _greeks_fd / _european_cv / _european_qmc
at n_gpus = 2 .. N with the one-GPU results (same paths, another association of the sums) and SKIP where only one GPU is visible
-- the one-GPU boxes this build is developed on.  The n_gpus = 1 cases always run: one rank through the real ncclAllReduce.
The multi-rank bodies run in a child process under a time limit (a hang or fault on hardware nobody has run them on must not end
the whole GPU suite); `python tests/test_gpu_multi_device.py <body> <n,n,...>` runs one by hand.

Also here: the regression test of round 4's host fault through the product's C ABI (caller streams that are destroyed)."""
import ctypes as C
import os
import subprocess
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import optionslab_amd as ol  # noqa: E402
from optionslab_amd import _hip  # noqa: E402

pytestmark = pytest.mark.gpu

ATM = (100.0, 100.0, 1.0, 0.05, 0.2)


def _hip_runtime():
    _hip.lib()                                   # the library (and with it the HIP runtime) is loaded and initialised
    for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
        try:
            return C.CDLL(name)
        except OSError:
            continue
    pytest.skip("HIP runtime not loadable through ctypes")


def _device_count() -> int:
    n = C.c_int(0)
    rt = _hip_runtime()
    assert rt.hipGetDeviceCount(C.byref(n)) == 0
    return n.value


def _rank_counts():
    n = _device_count()
    if n < 2:
        pytest.skip(f"{n} GPU visible: the multi-rank RCCL branch needs at least two (UNVERIFIED ON HARDWARE, see include/olmc.h)")
    return sorted({2, n} | ({4} if n >= 4 else set()))


def test_one_rank_goes_through_the_real_all_reduce_for_every_payload():
    """n_gpus = 1: engine, rank stream, ncclCommInitAll on one device, ncclAllReduce of 3 / 17 / 33 / 6 / 3 doubles, polled fetch."""
    S, K, T, r, v = ATM
    N, M, seed = 400_000, 20, 42
    a, b = _hip.multi_gpu_european(S, K, T, r, v, 0.0, True, N, M, seed, True, 1), _hip.european(S, K, T, r, v, 0.0, True, N, M, seed, True)
    assert (a.sum, a.sumsq, a.n, a.price, a.std_error) == (b.sum, b.sumsq, b.n, b.price, b.std_error)
    spans = _hip.multi_gpu_spans()
    assert spans["total_us"] > 0 and spans["launch_us"] > 0
    for second in (False, True):
        g1, e1 = _hip.multi_gpu_greeks_fd(S, K, T, r, v, 0.0, True, N, M, seed, second, 1, want_evals=True)
        g0, e0 = _hip.european_greeks_fd(S, K, T, r, v, 0.0, True, N, M, seed, second, want_evals=True)
        assert g1 == g0 and [(x.sum, x.sumsq) for x in e1] == [(x.sum, x.sumsq) for x in e0]
    c1, c0 = _hip.multi_gpu_european_cv(S, K, T, r, v, 0.01, False, N, M, seed, True, 1), _hip.european_cv(S, K, T, r, v, 0.01, False, N, M, seed, True)
    assert (c1.sum_d, c1.sum_s, c1.sum_dd, c1.sum_ss, c1.sum_ds, c1.value) == (c0.sum_d, c0.sum_s, c0.sum_dd, c0.sum_ss, c0.sum_ds, c0.value)
    sv, shift = ol.monte_carlo.sobol_tables(16, 42, 1 << 16)
    q1, q0 = _hip.multi_gpu_european_qmc(S, K, T, r, v, 0.0, True, 1 << 16, sv, shift, 1), _hip.european_qmc(S, K, T, r, v, 0.0, True, 1 << 16, sv, shift)
    assert (q1.sum, q1.sumsq, q1.n, q1.price) == (q0.sum, q0.sumsq, q0.n, q0.price)
    p = ol.MonteCarloPricer(1 << 16, 16, 42, ol.MCMethod.QMC, n_gpus=1).price(*ATM, "call")
    assert p == q0.price
    for second in (False, True):
        g1, e1 = _hip.multi_gpu_european_qmc_greeks_fd(S, K, T, r, v, 0.0, True, 1 << 16, sv, shift, second, 1)
        g0, e0 = _hip.european_qmc_greeks_fd(S, K, T, r, v, 0.0, True, 1 << 16, sv, shift, second)
        assert g1 == g0 and [(x.sum, x.sumsq) for x in e1] == [(x.sum, x.sumsq) for x in e0]
    k1, k0 = _hip.multi_gpu_european_qmc_cv(S, K, T, r, v, 0.01, False, 1 << 16, sv, shift, 1), _hip.european_qmc_cv(S, K, T, r, v, 0.01, False, 1 << 16, sv, shift)
    assert (k1.sum_d, k1.sum_s, k1.sum_dd, k1.sum_ss, k1.sum_ds, k1.value) == (k0.sum_d, k0.sum_s, k0.sum_dd, k0.sum_ss, k0.sum_ds, k0.value)


def _child_n_devices_price_what_one_device_prices(rank_counts):
    S, K, T, r, v = ATM
    N, M, seed = 2_000_003, 64, 9
    whole = _hip.european(S, K, T, r, v, 0.0, True, N, M, seed, True)
    cv0 = _hip.european_cv(S, K, T, r, v, 0.01, False, N, M, seed, True)
    sv, shift = ol.monte_carlo.sobol_tables(32, 42, 1 << 20)
    q0 = _hip.european_qmc(S, K, T, r, v, 0.0, True, 1 << 20, sv, shift)
    qg0, _ = _hip.european_qmc_greeks_fd(S, K, T, r, v, 0.0, True, 1 << 20, sv, shift, True, want_evals=False)
    qc0 = _hip.european_qmc_cv(S, K, T, r, v, 0.01, False, 1 << 20, sv, shift)
    for n_gpus in rank_counts:
        for launch in (0, -1):                          # launcher threads, then the serial form: the same bits
            _hip.tune(_hip.TUNE_MULTI_LAUNCH, launch)
            try:
                got = _hip.multi_gpu_european(S, K, T, r, v, 0.0, True, N, M, seed, True, n_gpus)
                parts = []
                for d in range(n_gpus):
                    lo, hi = N * d // n_gpus, N * (d + 1) // n_gpus
                    st = _hip.european(S, K, T, r, v, 0.0, True, hi - lo, M, seed, True, path_offset=lo)
                    parts.append((st.sum, st.sumsq, st.n))
                want = _hip.combine_stats(parts, r, T)
                assert got.n == whole.n == 2 * N
                # the ring's order of additions is RCCL's: rank-ordered sum to rounding, the one-device sums to 1e-13
                assert got.sum == pytest.approx(want.sum, rel=1e-15) and got.sumsq == pytest.approx(want.sumsq, rel=1e-15)
                assert got.sum == pytest.approx(whole.sum, rel=1e-13) and got.price == pytest.approx(whole.price, rel=1e-13)
                for second in (False, True):
                    gn, en = _hip.multi_gpu_greeks_fd(S, K, T, r, v, 0.0, True, N, M, seed, second, n_gpus, want_evals=True)
                    g1, e1 = _hip.european_greeks_fd(S, K, T, r, v, 0.0, True, N, M, seed, second, want_evals=True)
                    for a, b in zip(e1[:14 if second else 8], en):
                        assert b.n == a.n and b.sum == pytest.approx(a.sum, rel=1e-13) and b.sumsq == pytest.approx(a.sumsq, rel=1e-13)
                    assert gn[0] == pytest.approx(g1[0], rel=1e-13) and gn[1] == pytest.approx(g1[1], abs=1e-10)
                cv = _hip.multi_gpu_european_cv(S, K, T, r, v, 0.01, False, N, M, seed, True, n_gpus)
                assert cv.n == cv0.n and cv.value == pytest.approx(cv0.value, rel=1e-11)
                q = _hip.multi_gpu_european_qmc(S, K, T, r, v, 0.0, True, 1 << 20, sv, shift, n_gpus)
                assert q.n == q0.n and q.sum == pytest.approx(q0.sum, rel=1e-13) and q.price == pytest.approx(q0.price, rel=1e-13)
                qg, _ = _hip.multi_gpu_european_qmc_greeks_fd(S, K, T, r, v, 0.0, True, 1 << 20, sv, shift, True, n_gpus, want_evals=False)
                assert qg[0] == pytest.approx(qg0[0], rel=1e-13) and qg[1] == pytest.approx(qg0[1], abs=1e-9)
                qc = _hip.multi_gpu_european_qmc_cv(S, K, T, r, v, 0.01, False, 1 << 20, sv, shift, n_gpus)
                assert qc.n == qc0.n and qc.value == pytest.approx(qc0.value, rel=1e-11)
            finally:
                _hip.tune(_hip.TUNE_MULTI_LAUNCH, 0)
    assert _hip.device_info()["device"] == 0            # the thread's library device came back


def _child_the_pricer_over_n_devices(rank_counts):
    n = rank_counts[-1]
    one, many = ol.MonteCarloPricer(1_000_000, 252, 42), ol.MonteCarloPricer(1_000_000, 252, 42, n_gpus=n)
    a, b = one.price(*ATM, "call", return_error=True), many.price(*ATM, "call", return_error=True)
    assert b.n_paths == a.n_paths and b.price == pytest.approx(a.price, rel=1e-13) and b.std_error == pytest.approx(a.std_error, rel=1e-10)
    bs = ol.black_scholes(*ATM, "call")
    assert abs(b.price - bs) <= 3 * b.std_error
    g1, gn = one.greeks(*ATM, "call"), many.greeks(*ATM, "call")
    for k in g1:
        assert gn[k] == pytest.approx(g1[k], rel=1e-7, abs=1e-7), k



def _in_a_child(name: str, timeout_s: int = 900):
    """The multi-rank branch has never run on hardware (one-GPU boxes only): it runs in a CHILD process under a time limit, so that a
    hang or a fault in it is this test's failure and not the end of the whole GPU suite.  The child is killed by its PID on timeout."""
    counts = _rank_counts()
    r = subprocess.run([sys.executable, os.path.abspath(__file__), name, ",".join(map(str, counts))], capture_output=True, text=True,
                       timeout=timeout_s, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, f"child {name} rc {r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"


def test_n_devices_price_what_one_device_prices():
    _in_a_child("_child_n_devices_price_what_one_device_prices")


def test_the_pricer_over_n_devices():
    _in_a_child("_child_the_pricer_over_n_devices")


def test_caller_streams_that_die_are_never_touched_again():
    """Round 4's host fault, through the C ABI (see tests/test_gpu_instrumented.py::test_engines_come_and_go... for the story).  A
    caller stream claims a reduction workspace slot of the context that serves it; the caller may destroy the stream at any time.
    With every slot claimed by a dead stream, the next stream goes down the slot-sharing path: it must drain the device, not the dead
    handle.  Raw hipStreamCreate / hipStreamDestroy (torch pools its streams and never destroys one), three generations of 9 streams."""
    rt = _hip_runtime()
    rt.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    rt.hipStreamDestroy.argtypes = [C.c_void_p]
    rt.hipStreamSynchronize.argtypes = [C.c_void_p]
    rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    rt.hipFree.argtypes = [C.c_void_p]
    S, K, T, r, v = ATM
    N, M = 150_001, 16
    buf = C.c_void_p()
    assert rt.hipMalloc(C.byref(buf), 8 * 4 * 16) == 0
    try:
        for generation in range(3):
            streams = []
            for k in range(9):
                st = C.c_void_p()
                assert rt.hipStreamCreate(C.byref(st)) == 0
                streams.append(st)
            for k, st in enumerate(streams):
                seed = 100 * generation + k
                slot = buf.value + 32 * k
                _hip.european_shard_dev(S, K, T, r, v, 0.0, True, 0, N, M, seed, True, slot, st.value)
                got = _hip.fetch_dev(slot, 3, st.value)
                want = _hip.european(S, K, T, r, v, 0.0, True, N, M, seed, True)
                assert (got[0], got[1], int(got[2])) == (want.sum, want.sumsq, want.n)
            for st in streams:
                assert rt.hipStreamSynchronize(st) == 0
                assert rt.hipStreamDestroy(st) == 0
    finally:
        rt.hipFree(buf)
    again = _hip.european(S, K, T, r, v, 0.0, True, N, M, 7, True)
    assert again.n == 2 * N


if __name__ == "__main__":              # child of _in_a_child: one body, the rank counts the parent found
    globals()[sys.argv[1]]([int(x) for x in sys.argv[2].split(",")])
    _hip.shutdown()
