"""Pins oracle/philox_gbm.c (the CPU restatement of the device stream):
Random123 known answers, distribution of the normals, and statistical agreement
with the reference-pinned NumPy oracle."""
import math

import numpy as np
import pytest
from scipy import stats

from oracle import numpy_reference as orc
from oracle import philox_oracle as po

ATM = dict(S=100.0, K=100.0, T=1.0, r=0.05, sigma=0.2, q=0.0)

# Random123 kat_vectors, philox4x32 10 rounds
KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
    ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
    ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
     (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
]


@pytest.mark.parametrize("ctr,key,want", KAT)
def test_philox_known_answers(ctr, key, want):
    assert tuple(po.philox(ctr, key)) == want


def test_normals_are_standard_normal():
    z = po.normals(seed=42, path0=0, n_paths=40000, n_steps=10).astype(np.float64).ravel()
    n = z.size
    assert abs(z.mean()) < 4 / math.sqrt(n)
    assert abs(z.var() - 1.0) < 4 * math.sqrt(2.0 / n)
    assert abs(stats.skew(z)) < 4 * math.sqrt(6.0 / n)
    assert abs(stats.kurtosis(z)) < 4 * math.sqrt(24.0 / n)
    assert stats.kstest(z, "norm").pvalue > 1e-3
    assert np.abs(z).max() <= math.sqrt(2 * 33 * math.log(2)) + 1e-6   # tail cap of 32-bit uniforms


def test_normals_decorrelated_across_paths_steps_and_seeds():
    a = po.normals(1, 0, 20000, 8).astype(np.float64)
    b = po.normals(2, 0, 20000, 8).astype(np.float64)
    lim = 4 / math.sqrt(20000)
    assert abs(np.corrcoef(a[:, 0], a[:, 1])[0, 1]) < lim       # cos/sin pair of one Box-Muller
    assert abs(np.corrcoef(a[:, 1], a[:, 2])[0, 1]) < lim       # across the two pairs of a block
    assert abs(np.corrcoef(a[:, 3], a[:, 4])[0, 1]) < lim       # across Philox blocks
    assert abs(np.corrcoef(a[:-1, 0], a[1:, 0])[0, 1]) < lim    # neighbouring paths
    assert abs(np.corrcoef(a[:, 0], b[:, 0])[0, 1]) < lim       # seeds


def test_stream_depends_only_on_global_path_index():
    whole = po.normals(9, 0, 64, 7)
    part = po.normals(9, 40, 24, 7)
    assert np.array_equal(whole[40:], part)
    far = po.normals(9, (1 << 32) + 5, 2, 7)      # counter word 1 carries the high half
    assert not np.array_equal(far, po.normals(9, 5, 2, 7))


@pytest.mark.parametrize("typ,q,M", [("call", 0.0, 252), ("put", 0.02, 50), ("call", 0.0, 1), ("put", 0.0, 7)])
def test_european_agrees_with_reference_oracle_within_3_sigma(typ, q, M):
    N = 100000
    p = dict(ATM, q=q)
    sx, sxx, ss, sss, sxs, n = po.european_moments(p["S"], p["K"], p["T"], p["r"], p["sigma"], q, typ == "call", N, M, 42)
    price, se = po.price_and_error(sx, sxx, n, p["r"], p["T"])
    ref = orc.OraclePricer(N, M, 42).price(p["S"], p["K"], p["T"], p["r"], p["sigma"], typ, q, return_error=True)
    assert n == ref.n_paths == 2 * N
    assert abs(price - ref.price) <= 3 * math.hypot(se, ref.std_error)
    assert abs(price - float(orc.bs_price(p["S"], p["K"], p["T"], p["r"], p["sigma"], typ, q))) <= 3 * se
    assert abs(se / ref.std_error - 1) < 0.05
    # antithetic pair symmetry of the control moment: E[S_T] = S e^{(r-q)T}
    se_s = math.sqrt(max(sss / n - (ss / n) ** 2, 0) / n)
    assert abs(ss / n - p["S"] * math.exp((p["r"] - q) * p["T"])) <= 4 * se_s


def test_terminal_layout_pos_then_neg():
    st = po.european_terminal(100.0, 1.0, 0.05, 0.2, 0.0, 1000, 12, 3)
    a = math.log(100.0) + (0.05 - 0.5 * 0.04) * 1.0
    # antithetic legs mirror around the drifted log-spot (gbm_numpy.py:46-51)
    assert np.allclose(np.log(st[:1000]) - a, -(np.log(st[1000:]) - a), rtol=0, atol=1e-12)
    sharded = np.concatenate([po.european_terminal(100.0, 1.0, 0.05, 0.2, 0.0, 400, 12, 3, True, 0)[:400],
                              po.european_terminal(100.0, 1.0, 0.05, 0.2, 0.0, 600, 12, 3, True, 400)[:600]])
    assert np.array_equal(sharded, st[:1000])


@pytest.mark.parametrize("geometric", [False, True])
def test_asian_agrees_with_reference_oracle(geometric):
    N, M = 20000, 64
    sx, sxx, n = po.asian_moments(100.0, 100.0, 1.0, 0.05, 0.2, 0.0, True, geometric, N, M, 42)
    price, se = po.price_and_error(sx, sxx, n, 0.05, 1.0)
    ref = float(orc.asian_price(100.0, 100.0, 1.0, 0.05, 0.2, 0.0, 42, N, M, "geometric" if geometric else "arithmetic", "call"))
    assert n == N
    assert abs(price - ref) <= 3 * math.sqrt(2) * se      # both estimates carry ~se
