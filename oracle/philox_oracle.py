"""ctypes wrapper of oracle/philox_gbm.c (CPU restatement of the device stream).
TEST INFRASTRUCTURE ONLY -- see the header of philox_gbm.c."""
import ctypes as C
import math
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libphilox_oracle.so")
_U64 = (1 << 64) - 1


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "philox_gbm.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(src) > os.path.getmtime(LIB):
        subprocess.run(["make", "-s", "-C", HERE, "-B"], check=True)
    return LIB


_lib = None


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def philox(ctr, key):
    c, k, o = (C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), (C.c_uint32 * 4)()
    _load().ol_philox4x32_10(c, k, o)
    return list(o)


def philox_words(seed, path0, n_paths, block0, n_blocks, tag=0):
    """Same layout as olmc_philox_words: out[p, b, w]."""
    out = np.empty((n_paths, n_blocks, 4), dtype=np.uint32)
    s = int(seed) & _U64
    key = (s & 0xFFFFFFFF, s >> 32)
    for p in range(n_paths):
        g = path0 + p
        for b in range(n_blocks):
            out[p, b] = philox((g & 0xFFFFFFFF, g >> 32, block0 + b, tag), key)
    return out


def normals(seed, path0, n_paths, n_steps):
    out = np.empty((n_paths, n_steps), dtype=np.float32)
    _load().ol_normals(C.c_uint64(int(seed) & _U64), C.c_int64(path0), C.c_int64(n_paths), C.c_int32(n_steps),
                       out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def european_terminal(S, T, r, sigma, q, n_paths, n_steps, seed, antithetic=True, path0=0):
    out = np.empty(n_paths * (2 if antithetic else 1), dtype=np.float64)
    _load().ol_european_terminal(C.c_double(S), C.c_double(T), C.c_double(r), C.c_double(sigma), C.c_double(q),
                                 C.c_int64(path0), C.c_int64(n_paths), C.c_int32(n_steps),
                                 C.c_uint64(int(seed) & _U64), C.c_int(int(antithetic)),
                                 out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def european_moments(S, K, T, r, sigma, q, is_call, n_paths, n_steps, seed, antithetic=True, path0=0):
    """-> (sum_x, sum_xx, sum_s, sum_ss, sum_xs, n)"""
    m = (C.c_double * 5)()
    _load().ol_european_moments(C.c_double(S), C.c_double(K), C.c_double(T), C.c_double(r), C.c_double(sigma),
                                C.c_double(q), C.c_int(int(is_call)), C.c_int64(path0), C.c_int64(n_paths),
                                C.c_int32(n_steps), C.c_uint64(int(seed) & _U64), C.c_int(int(antithetic)), m)
    return (*list(m), n_paths * (2 if antithetic else 1))


def asian_moments(S, K, T, r, sigma, q, is_call, geometric, n_paths, n_steps, seed, antithetic=False, path0=0):
    m = (C.c_double * 2)()
    _load().ol_asian_moments(C.c_double(S), C.c_double(K), C.c_double(T), C.c_double(r), C.c_double(sigma),
                             C.c_double(q), C.c_int(int(is_call)), C.c_int(int(geometric)), C.c_int64(path0),
                             C.c_int64(n_paths), C.c_int32(n_steps), C.c_uint64(int(seed) & _U64),
                             C.c_int(int(antithetic)), m)
    return m[0], m[1], n_paths * (2 if antithetic else 1)


def extrema_moments(S, K, T, r, sigma, q, is_call, payoff, barrier, n_paths, n_steps, seed, antithetic=False, path0=0):
    """payoff: 0 up-out, 1 up-in, 2 down-out, 3 down-in, 4 lookback floating, 5 lookback fixed."""
    m = (C.c_double * 2)()
    _load().ol_extrema_moments(C.c_double(S), C.c_double(K), C.c_double(T), C.c_double(r), C.c_double(sigma), C.c_double(q),
                               C.c_int(int(is_call)), C.c_int(int(payoff)), C.c_double(barrier), C.c_int64(path0),
                               C.c_int64(n_paths), C.c_int32(n_steps), C.c_uint64(int(seed) & _U64), C.c_int(int(antithetic)), m)
    return m[0], m[1], n_paths * (2 if antithetic else 1)


def heston_moments(S, K, T, r, q, is_call, kappa, theta, sigma_v, rho, v0, n_paths, n_steps, seed, antithetic=False, path0=0):
    m = (C.c_double * 2)()
    _load().ol_heston_moments(C.c_double(S), C.c_double(K), C.c_double(T), C.c_double(r), C.c_double(q), C.c_int(int(is_call)),
                              C.c_double(kappa), C.c_double(theta), C.c_double(sigma_v), C.c_double(rho), C.c_double(v0),
                              C.c_int64(path0), C.c_int64(n_paths), C.c_int32(n_steps), C.c_uint64(int(seed) & _U64),
                              C.c_int(int(antithetic)), m)
    return m[0], m[1], n_paths * (2 if antithetic else 1)


def autocall_moments(S, T, r, sigma, q, autocall_b, coupon_b, coupon_rate, ki_b, freq, n_paths, n_steps, seed, antithetic=False, path0=0):
    m = (C.c_double * 2)()
    _load().ol_autocall_moments(*(C.c_double(x) for x in (S, T, r, sigma, q, autocall_b, coupon_b, coupon_rate, ki_b)), C.c_int32(freq),
                                C.c_int64(path0), C.c_int64(n_paths), C.c_int32(n_steps), C.c_uint64(int(seed) & _U64),
                                C.c_int(int(antithetic)), m)
    return m[0], m[1], n_paths * (2 if antithetic else 1)


def cliquet_moments(S, T, r, sigma, q, lcap, lfloor, gcap, gfloor, n_periods, n_paths, n_steps, seed, antithetic=False, path0=0):
    m = (C.c_double * 2)()
    _load().ol_cliquet_moments(*(C.c_double(x) for x in (S, T, r, sigma, q, lcap, lfloor, gcap, gfloor)), C.c_int32(n_periods),
                               C.c_int64(path0), C.c_int64(n_paths), C.c_int32(n_steps), C.c_uint64(int(seed) & _U64),
                               C.c_int(int(antithetic)), m)
    return m[0], m[1], n_paths * (2 if antithetic else 1)


def american_lsm(S, K, T, r, sigma, q, is_call, n_paths, n_steps, degree, seed):
    """-> (sum, sumsq, n) of the time-0 cash flows (already discounted)."""
    m = (C.c_double * 2)()
    rc = _load().ol_american_lsm(C.c_double(S), C.c_double(K), C.c_double(T), C.c_double(r), C.c_double(sigma), C.c_double(q),
                                 C.c_int(int(is_call)), C.c_int64(n_paths), C.c_int32(n_steps), C.c_int32(degree),
                                 C.c_uint64(int(seed) & _U64), m)
    assert rc == 0
    return m[0], m[1], n_paths


def jump_moments(S, K, T, r, sigma, q, is_call, kou, lambda_j, a1, a2, a3, n_paths, n_steps, seed, path0=0):
    m = (C.c_double * 2)()
    _load().ol_jump_moments(C.c_double(S), C.c_double(K), C.c_double(T), C.c_double(r), C.c_double(sigma), C.c_double(q),
                            C.c_int(int(is_call)), C.c_int(int(kou)), C.c_double(lambda_j), C.c_double(a1), C.c_double(a2),
                            C.c_double(a3), C.c_int64(path0), C.c_int64(n_paths), C.c_int32(n_steps), C.c_uint64(int(seed) & _U64), m)
    return m[0], m[1], n_paths


def gbm_paths(S, T, r, sigma, q, n_paths, n_steps, seed):
    out = np.empty((n_steps + 1, n_paths), dtype=np.float64)
    _load().ol_gbm_paths(C.c_double(S), C.c_double(T), C.c_double(r), C.c_double(sigma), C.c_double(q), C.c_int64(n_paths),
                         C.c_int32(n_steps), C.c_uint64(int(seed) & _U64), out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def heston_paths(S, T, r, q, kappa, theta, sigma_v, rho, v0, n_paths, n_steps, seed):
    spot = np.empty((n_steps + 1, n_paths), dtype=np.float64)
    var = np.empty_like(spot)
    _load().ol_heston_paths(*(C.c_double(x) for x in (S, T, r, q, kappa, theta, sigma_v, rho, v0)), C.c_int64(n_paths), C.c_int32(n_steps),
                            C.c_uint64(int(seed) & _U64), spot.ctypes.data_as(C.POINTER(C.c_double)), var.ctypes.data_as(C.POINTER(C.c_double)))
    return spot, var


def jump_paths(S, T, r, sigma, q, kou, lambda_j, a1, a2, a3, n_paths, n_steps, seed):
    out = np.empty((n_steps + 1, n_paths), dtype=np.float64)
    _load().ol_jump_paths(*(C.c_double(x) for x in (S, T, r, sigma, q)), C.c_int(int(kou)), C.c_double(lambda_j), C.c_double(a1),
                          C.c_double(a2), C.c_double(a3), C.c_int64(n_paths), C.c_int32(n_steps), C.c_uint64(int(seed) & _U64),
                          out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def price_and_error(sum_x, sum_xx, n, r, T):
    """monte_carlo.py:145-150 on the moments."""
    disc, mean = math.exp(-r * T), sum_x / n
    var = max(sum_xx / n - mean * mean, 0.0)
    return disc * mean, disc * math.sqrt(var) / math.sqrt(n)
