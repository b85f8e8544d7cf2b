"""CPU oracle: NumPy restatement of the reference Monte Carlo hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``optionslab_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and there only as the checker / the timed CPU
baseline, never as the thing shipped.

Parity status: PINNED.  Every function below is checked bit-for-bit against
vectors captured by importing the reference itself in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/reference_vectors.json``,
NumPy 2.2.6 / SciPy 1.15.3); see ``tests/test_oracle_golden.py``.

The arithmetic order of every expression is the reference's, because fp64
results are compared with ``==``.  Citations are ``/root/reference`` paths.
"""

from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Callable, Optional

import numpy as np
from scipy.stats import norm
from scipy.stats.qmc import Sobol

SOBOL_MAX_DIM = 21201  # src/simulation/gbm_qmc.py:29-30


# --------------------------------------------------------------------------
# a13  Black-Scholes-Merton closed form (src/pricing_models/black_scholes.py:9-52)
# --------------------------------------------------------------------------
def bs_price(S, K, T, r, sigma, option_type="call", q=0.0):
    if S <= 0 or K <= 0 or T < 0 or sigma < 0:  # :31-34
        raise ValueError("Invalid input: all inputs must be positive, and T, sigma >= 0")
    if T == 0:  # :37-38
        return max(S - K, 0.0) if option_type == "call" else max(K - S, 0.0)
    sq = sigma * np.sqrt(T)
    d1 = (np.log(S / K) + (r - q + 0.5 * sigma**2) * T) / sq  # :40
    d2 = d1 - sq  # :41
    if option_type == "call":  # :43-44
        return S * np.exp(-q * T) * norm.cdf(d1) - K * np.exp(-r * T) * norm.cdf(d2)
    if option_type == "put":  # :46-47
        return K * np.exp(-r * T) * norm.cdf(-d2) - S * np.exp(-q * T) * norm.cdf(-d1)
    raise ValueError("option_type must be 'call' or 'put'")


def bs_greeks(S, K, T, r, sigma, option_type="call", q=0.0):
    """Analytic BSM Greeks (not in the reference; SURVEY G0' accuracy anchor)."""
    sq = sigma * math.sqrt(T)
    d1 = (math.log(S / K) + (r - q + 0.5 * sigma * sigma) * T) / sq
    d2 = d1 - sq
    pdf = math.exp(-0.5 * d1 * d1) / math.sqrt(2.0 * math.pi)
    eq, er = math.exp(-q * T), math.exp(-r * T)
    sign = 1.0 if option_type == "call" else -1.0
    Nd1, Nd2 = float(norm.cdf(sign * d1)), float(norm.cdf(sign * d2))
    return OrderedDict(
        delta=sign * eq * Nd1,
        gamma=eq * pdf / (S * sq),
        vega=S * eq * pdf * math.sqrt(T),
        theta=(-S * eq * pdf * sigma / (2.0 * math.sqrt(T)) - sign * r * K * er * Nd2 + sign * q * S * eq * Nd1),
        rho=sign * K * T * er * Nd2,
    )


# --------------------------------------------------------------------------
# a4/a5/a10  terminal-price backends (src/simulation/*.py)
# --------------------------------------------------------------------------
def terminal_multistep(S, T, r, sigma, q, n_paths, n_steps, seed, antithetic=True):
    """src/simulation/gbm_numpy.py:15-53 -- PCG64 normals (N, M), row sums."""
    gen = np.random.default_rng(seed)  # :32
    dt = T / n_steps  # :35
    drift = (r - q - 0.5 * sigma * sigma) * dt  # :36
    vol = sigma * np.sqrt(dt)  # :37
    total_drift = drift * n_steps  # :38
    log_S0 = np.log(S)  # :39
    Z = gen.standard_normal((n_paths, n_steps))  # :43
    up = log_S0 + total_drift + vol * np.sum(Z, axis=1)  # :46
    if not antithetic:
        return np.exp(up)  # :53
    dn = log_S0 + total_drift - vol * np.sum(Z, axis=1)  # :50
    return np.concatenate([np.exp(up), np.exp(dn)])  # :51  layout [pos | neg]


def terminal_singlestep(S, T, r, sigma, q, n_paths, seed):
    """src/simulation/gbm_numpy.py:56-83 -- closed-form single draw per path."""
    gen = np.random.default_rng(seed)  # :71
    drift = (r - q - 0.5 * sigma * sigma) * T  # :73
    vol = sigma * np.sqrt(T)  # :74
    log_S0 = np.log(S)  # :75
    Z = gen.standard_normal(n_paths)  # :77
    return np.concatenate([np.exp(log_S0 + drift + vol * Z), np.exp(log_S0 + drift - vol * Z)])  # :80-83


def full_paths(S, T, r, sigma, q, n_paths, n_steps, seed):
    """src/simulation/gbm_numpy.py:86-118 -- (n_paths, n_steps + 1), column 0 = S, PCG64, no antithetic."""
    gen = np.random.default_rng(seed)
    dt = T / n_steps
    drift = (r - q - 0.5 * sigma * sigma) * dt
    vol = sigma * np.sqrt(dt)
    Z = gen.standard_normal((n_paths, n_steps))
    log_S = np.log(S) + np.cumsum(drift + vol * Z, axis=1)
    out = np.empty((n_paths, n_steps + 1))
    out[:, 0] = S
    out[:, 1:] = np.exp(log_S)
    return out


def terminal_sobol(S, T, r, sigma, q, n_paths, n_steps, seed):
    """src/simulation/gbm_qmc.py:14-46 -- scrambled Sobol, no antithetic."""
    dims = min(n_steps, SOBOL_MAX_DIM)  # :30
    u = Sobol(d=dims, scramble=True, seed=seed).random(n_paths)  # :32-33
    z = norm.ppf(np.clip(u, 1e-10, 1 - 1e-10))  # :36
    dt = T / dims  # :38
    drift = (r - q - 0.5 * sigma * sigma) * dt
    vol = sigma * np.sqrt(dt)
    log_S0 = np.log(S)
    return np.exp(log_S0 + drift * dims + vol * np.sum(z, axis=1))  # :44-46


def terminal_sobol_antithetic(S, T, r, sigma, q, n_paths, n_steps, seed):
    """src/simulation/gbm_qmc.py:49-76 -- Sobol normals and their mirror, [pos | neg]."""
    dims = min(n_steps, SOBOL_MAX_DIM)
    u = Sobol(d=dims, scramble=True, seed=seed).random(n_paths)
    z = norm.ppf(np.clip(u, 1e-10, 1 - 1e-10))
    dt = T / dims
    drift_total = (r - q - 0.5 * sigma * sigma) * T  # :69
    vol = sigma * np.sqrt(dt)
    sum_z = vol * np.sum(z, axis=1)
    return np.concatenate([np.exp(np.log(S) + drift_total + sum_z), np.exp(np.log(S) + drift_total - sum_z)])


# --------------------------------------------------------------------------
# a1-a3, a6-a8  the pricer (src/pricing_models/monte_carlo.py:28-186)
# --------------------------------------------------------------------------
@dataclass
class OracleResult:  # MCResult, monte_carlo.py:37-43
    price: float
    std_error: float = 0.0
    n_paths: int = 0


class OraclePricer:
    """MonteCarloPricer restated (monte_carlo.py:46-186); method is a plain
    string in {"numpy", "numba", "qmc", "fast"} (MCMethod values, :28-34)."""

    def __init__(self, num_simulations=100000, num_steps=1, seed=None, method="numpy"):
        if num_simulations < 1:  # :63-64
            raise ValueError("num_simulations must be >= 1")
        self.num_simulations = num_simulations
        self.num_steps = num_steps
        self.seed = seed if seed is not None else np.random.default_rng().integers(0, 2**31)  # :68-70
        self.method = method

    def terminal(self, S, T, r, sigma, q, seed=None):  # _simulate, :74-106
        s = seed if seed is not None else self.seed  # :84
        if self.method == "fast" or (self.num_steps == 1 and self.method == "numpy"):  # :87-92
            return terminal_singlestep(S, T, r, sigma, q, self.num_simulations, s)
        if self.method == "qmc":  # :94-97
            return terminal_sobol(S, T, r, sigma, q, self.num_simulations, self.num_steps, s)
        # numba is absent wherever this oracle runs => silent NumPy path (:72, :99-106)
        return terminal_multistep(S, T, r, sigma, q, self.num_simulations, self.num_steps, s)

    def price(self, S, K, T, r, sigma, option_type, q=0.0, seed=None, return_error=False):  # :108-152
        if T <= 0:  # :133-135
            intrinsic = max(S - K, 0) if option_type == "call" else max(K - S, 0)
            return OracleResult(intrinsic, 0.0, 0) if return_error else intrinsic
        st = self.terminal(S, T, r, sigma, q, seed)
        x = np.maximum(st - K, 0.0) if option_type == "call" else np.maximum(K - st, 0.0)  # :140-143
        disc = np.exp(-r * T)
        value = float(disc * np.mean(x))  # :145-146
        if return_error:  # :148-150  naive ddof=0 estimator over all 2N samples
            return OracleResult(value, float(disc * np.std(x) / np.sqrt(len(x))), len(x))
        return value

    def price_with_control_variate(self, S, K, T, r, sigma, option_type, q=0.0, seed=None):  # :154-186
        st = self.terminal(S, T, r, sigma, q, seed)
        x = np.maximum(st - K, 0.0) if option_type == "call" else np.maximum(K - st, 0.0)
        d = np.exp(-r * T) * x  # :175
        fwd = S * np.exp((r - q) * T)  # :179
        c = np.cov(d, st)  # :181  (ddof=1)
        beta = c[0, 1] / c[1, 1] if c[1, 1] > 1e-10 else 0.0  # :182
        return float(np.mean(d) - beta * (np.mean(st) - fwd))  # :184


# --------------------------------------------------------------------------
# a11  finite-difference Greeks driver (src/greeks/unified_greeks.py:235-367)
# --------------------------------------------------------------------------
FD_H_RATE = 1e-4  # :276
FD_H_TIME = 1 / 365.0  # :277


def fd_steps(S):
    """(h_S, h_sigma, h_r, h_T) -- unified_greeks.py:274-277."""
    return max(1e-4, 0.01 * S), max(1e-4, 0.01), FD_H_RATE, FD_H_TIME


def fd_greeks(price_fn: Callable[..., float], S, K, T, r, sigma, option_type="call", q=0.0,
              include_second_order=True, **kw):
    """``price_fn(S,K,T,r,sigma,option_type,q,**kw)``; bump-and-reprice with a
    memo on the parameter tuple (:280-288).  Returns the same OrderedDict."""
    h_S, h_v, h_r, h_T = fd_steps(S)
    memo = {}

    def P(S_=S, T_=T, r_=r, v_=sigma):
        k = (S_, K, T_, r_, v_, q)
        if k not in memo:
            memo[k] = price_fn(S_, K, T_, r_, v_, option_type, q, **kw)
        return memo[k]

    mid = P()  # :295
    su, sd = P(S_=S + h_S), P(S_=S - h_S)  # :298-299
    delta = (su - sd) / (2 * h_S)  # :301
    gamma = (su - 2 * mid + sd) / (h_S**2)  # :302
    vu, vd = P(v_=sigma + h_v), P(v_=sigma - h_v)  # :305-306
    vega = (vu - vd) / (2 * h_v)  # :307
    if T > h_T:  # :310-314
        theta = (P(T_=T - h_T) - mid) / h_T
    else:
        theta = -mid / max(T, 1e-6)
    ru, rd = P(r_=r + h_r), P(r_=r - h_r)  # :317-318
    rho = (ru - rd) / (2 * h_r)  # :319
    out = OrderedDict(price=mid, delta=delta, gamma=gamma, vega=vega, theta=theta, rho=rho)
    if include_second_order:  # :336-362
        uu, ud = P(S_=S + h_S, v_=sigma + h_v), P(S_=S + h_S, v_=sigma - h_v)
        du, dd = P(S_=S - h_S, v_=sigma + h_v), P(S_=S - h_S, v_=sigma - h_v)
        out["vanna"] = (uu - ud - du + dd) / (4 * h_S * h_v)  # :343-345
        if T > h_T:  # :348-354
            d_down = (P(S_=S + h_S, T_=T - h_T) - P(S_=S - h_S, T_=T - h_T)) / (2 * h_S)
            out["charm"] = (d_down - delta) / h_T
        else:
            out["charm"] = 0.0
        out["vomma"] = (vu - 2 * mid + vd) / (h_v**2)  # :357
    return out


# --------------------------------------------------------------------------
# a12  Asian option on full paths (src/pricing_models/exotic_options.py:40-160)
# --------------------------------------------------------------------------
def asian_paths(S, T, r, sigma, q, n_paths, n_steps, seed: Optional[int]):
    """exotic_options.py:40-67 -- legacy global RandomState (MT19937), no antithetic."""
    if seed is not None:  # :51-52
        np.random.seed(seed)
    dt = T / n_steps
    drift = (r - q - 0.5 * sigma**2) * dt  # :55
    diffusion = sigma * np.sqrt(dt)  # :56
    Z = np.random.standard_normal((n_paths, n_steps))  # :59
    steps = drift + diffusion * Z  # :62
    log_S = np.zeros((n_paths, n_steps + 1))
    log_S[:, 0] = np.log(S)
    log_S[:, 1:] = np.log(S) + np.cumsum(steps, axis=1)  # :65
    return np.exp(log_S)  # :67


def asian_price(S, K, T, r, sigma, q=0.0, seed=None, n_paths=100000, n_steps=252,
                avg_type="arithmetic", option_type="call", return_error=False):
    """exotic_options.py:97-131 -- average over t=1..M (t=0 excluded).  `return_error=True` adds what the reference does not
    return: the standard error of the SAME payoffs, disc * std(x, ddof=0) / sqrt(n) as monte_carlo.py:147-149 takes it."""
    paths = asian_paths(S, T, r, sigma, q, n_paths, n_steps, seed)
    if avg_type == "arithmetic":  # :119-120
        avg = np.mean(paths[:, 1:], axis=1)
    else:  # :121-122
        avg = np.exp(np.mean(np.log(paths[:, 1:]), axis=1))
    del paths
    x = np.maximum(avg - K, 0) if option_type == "call" else np.maximum(K - avg, 0)  # :125-128
    price = np.exp(-r * T) * np.mean(x)  # :131  (np.float64)
    if return_error:
        return price, np.exp(-r * T) * np.std(x) / np.sqrt(n_paths)
    return price


def barrier_price(S, K, T, r, sigma, barrier, q=0.0, seed=None, n_paths=100000, n_steps=252,
                  barrier_type="up-and-out", option_type="call"):
    """exotic_options.py:174-224 -- monitoring includes t = 0 (paths[:, 0] = S)."""
    if barrier <= 0:  # :195-196
        raise ValueError("Barrier must be positive")
    paths = asian_paths(S, T, r, sigma, q, n_paths, n_steps, seed)
    crossed = np.any(paths >= barrier, axis=1) if barrier_type.startswith("up") else np.any(paths <= barrier, axis=1)
    active = ~crossed if barrier_type.endswith("out") else crossed  # :207-212
    st = paths[:, -1]
    x = np.maximum(st - K, 0) if option_type == "call" else np.maximum(K - st, 0)
    return np.exp(-r * T) * np.mean(x * active)  # :221-223


def lookback_price(S, K, T, r, sigma, q=0.0, seed=None, n_paths=100000, n_steps=252, lookback_type="floating",
                   option_type="call"):
    """exotic_options.py:359-401"""
    paths = asian_paths(S, T, r, sigma, q, n_paths, n_steps, seed)
    st, hi, lo = paths[:, -1], np.max(paths, axis=1), np.min(paths, axis=1)
    if lookback_type == "floating":
        x = st - lo if option_type == "call" else hi - st
    else:
        x = np.maximum(hi - K, 0) if option_type == "call" else np.maximum(K - lo, 0)
    return np.exp(-r * T) * np.mean(x)


def american_price(S, K, T, r, sigma, q=0.0, seed=None, n_paths=50000, n_steps=50, option_type="put", poly_degree=3):
    """exotic_options.py:237-305 Longstaff-Schwartz."""
    paths = asian_paths(S, T, r, sigma, q, n_paths, n_steps, seed)
    dt = T / n_steps
    discount = np.exp(-r * dt)
    intrinsic = np.maximum(paths - K, 0) if option_type == "call" else np.maximum(K - paths, 0)
    cf = intrinsic[:, -1].copy()
    for t in range(n_steps - 1, 0, -1):  # :273-300
        cf = cf * discount
        itm = intrinsic[:, t] > 0
        if np.sum(itm) > poly_degree + 1:
            X, Y = paths[itm, t], cf[itm]
            Xp = np.column_stack([X**i for i in range(poly_degree + 1)])
            coeffs = np.linalg.lstsq(Xp, Y, rcond=None)[0]
            cont = Xp @ coeffs
            ex = intrinsic[itm, t] > cont
            idx = np.where(itm)[0][ex]
            cf[idx] = intrinsic[idx, t]
    return np.mean(cf * discount)


def exercise_boundary_from_paths(paths, K, option_type="put"):
    """exotic_options.py:322-345 on a given (n_paths, n_steps + 1) matrix."""
    intrinsic = np.maximum(paths - K, 0) if option_type == "call" else np.maximum(K - paths, 0)
    boundary = np.zeros(paths.shape[1])
    for t in range(paths.shape[1]):
        itm = intrinsic[:, t] > 0
        if np.sum(itm) > 0:
            boundary[t] = np.percentile(paths[itm, t], 10 if option_type == "put" else 90)
        else:
            boundary[t] = np.nan
    return boundary


def american_exercise_boundary(S, K, T, r, sigma, q=0.0, seed=None, n_paths=10000, n_steps=50, option_type="put"):
    """exotic_options.py:309-345 -- (times, boundary)."""
    paths = asian_paths(S, T, r, sigma, q, n_paths, n_steps, seed)
    return np.linspace(0, T, n_steps + 1), exercise_boundary_from_paths(paths, K, option_type)


def autocallable_price(S, T, r, sigma, q=0.0, seed=None, n_paths=100000, n_steps=252, observation_freq=21,
                       autocall_barrier=1.0, coupon_barrier=0.8, coupon_rate=0.10, ki_barrier=0.6):
    """exotic_options.py:404-491"""
    paths = asian_paths(S, T, r, sigma, q, n_paths, n_steps, seed)
    dt = T / n_steps
    obs = list(range(observation_freq, n_steps + 1, observation_freq))
    pay = np.zeros(n_paths)
    redeemed = np.zeros(n_paths, dtype=bool)
    knocked_in = np.min(paths / S, axis=1) <= ki_barrier  # :445-446
    for i, t in enumerate(obs):  # :448-463
        active = ~redeemed
        hit = np.where(active)[0][paths[active, t] / S >= autocall_barrier]
        pay[hit] = (1 + coupon_rate * ((i + 1) / len(obs)) * T) * np.exp(-r * t * dt)
        redeemed[hit] = True
    still = ~redeemed
    rel = paths[still, -1] / S
    fin = np.ones(np.sum(still))
    fin[rel >= coupon_barrier] += coupon_rate * T  # :477-478
    loss = knocked_in[still] & (rel < 1.0)  # :481-483
    fin[loss] = rel[loss]
    pay[still] = fin * np.exp(-r * T)
    return np.mean(pay)


def cliquet_price(S, T, r, sigma, q=0.0, seed=None, n_paths=100000, n_steps=252, n_periods=12, local_cap=0.05,
                  local_floor=-0.05, global_cap=0.30, global_floor=0.0):
    """exotic_options.py:494-554"""
    paths = asian_paths(S, T, r, sigma, q, n_paths, n_steps, seed)
    spp = n_steps // n_periods
    total = np.zeros(n_paths)
    for p_ in range(n_periods):
        a, b = paths[:, p_ * spp], paths[:, (p_ + 1) * spp]
        total += np.clip((b - a) / a, local_floor, local_cap)
    total = np.clip(total, global_floor, global_cap)
    return np.exp(-r * T) * np.mean(np.maximum(total, 0) * S)


def asian_geometric_closed_form(S, K, T, r, sigma, q=0.0, option_type="call"):
    """exotic_options.py:133-160 (continuous-monitoring lognormal approximation)."""
    v = sigma / np.sqrt(3)  # :145
    b = 0.5 * (r - q - sigma**2 / 6)  # :146
    d1 = (np.log(S / K) + (b + 0.5 * v**2) * T) / (v * np.sqrt(T))  # :148-150
    d2 = d1 - v * np.sqrt(T)  # :151
    if option_type == "call":  # :153-156
        return S * np.exp((b - r) * T) * norm.cdf(d1) - K * np.exp(-r * T) * norm.cdf(d2)
    return K * np.exp(-r * T) * norm.cdf(-d2) - S * np.exp((b - r) * T) * norm.cdf(-d1)  # :158-160


# --------------------------------------------------------------------------
# MonteCarloPricerUni, NumPy backend (src/pricing_models/monte_carlo_unified.py:298-343, 451-689)
# --------------------------------------------------------------------------
class OracleUni:
    def __init__(self, num_simulations=100_000, num_steps=100, seed=None):
        if num_simulations <= 0 or num_steps <= 0:  # :277-280
            raise ValueError("num_simulations and num_steps must be positive integers")
        self.num_simulations, self.num_steps = num_simulations, num_steps
        self.seed = seed if seed is not None else np.random.default_rng().integers(0, 2**31)
        self.rng = np.random.default_rng(seed)  # :287

    def terminal(self, S, T, r, sigma, q, seed=None):
        """:298-343 -- Z has shape (n_options, sims, steps): option j consumes slice j of ONE stream."""
        gen = np.random.default_rng(seed if seed is not None else self.seed)
        n = len(S)
        dt = T / self.num_steps
        drift = (r - q - 0.5 * sigma**2)[:, None] * dt[:, None]
        vol = sigma[:, None] * np.sqrt(dt[:, None])
        Z = gen.standard_normal((n, self.num_simulations, self.num_steps))
        log_S = np.log(S)[:, None, None]
        up = log_S + np.cumsum(drift[:, None, :] + vol[:, None, :] * Z, axis=2)
        dn = log_S + np.cumsum(drift[:, None, :] - vol[:, None, :] * Z, axis=2)
        return np.concatenate([np.exp(up[:, :, -1]), np.exp(dn[:, :, -1])], axis=1)

    def price(self, S, K, T, r, sigma, option_type, q=0.0, seed=None):  # :451-511
        st = self.terminal(np.array([S]), np.array([T]), np.array([r]), np.array([sigma]), np.array([q]), seed)[0]
        x = np.maximum(st - K, 0.0) if option_type == "call" else np.maximum(K - st, 0.0)
        return float(np.exp(-r * T) * np.mean(x))

    def delta_gamma(self, S, K, T, r, sigma, option_type, q=0.0, h=1e-4, seed=None):  # :513-560
        if seed is None:
            seed = int(self.rng.integers(0, 2**31))
        up = self.price(S + h, K, T, r, sigma, option_type, q, seed=seed)
        mid = self.price(S, K, T, r, sigma, option_type, q, seed=seed)
        dn = self.price(S - h, K, T, r, sigma, option_type, q, seed=seed)
        return (up - dn) / (2 * h), (up - 2 * mid + dn) / (h**2)

    def price_batch(self, S, K, T, r, sigma, option_type, q=0.0):  # :562-631
        S, K, T, r, sigma = (np.asarray(a, dtype=np.float64) for a in (S, K, T, r, sigma))
        q = np.full_like(S, q) if isinstance(q, (int, float)) else np.asarray(q, dtype=np.float64)
        st = self.terminal(S, T, r, sigma, q)
        x = np.maximum(st - K[:, None], 0.0) if option_type == "call" else np.maximum(K[:, None] - st, 0.0)
        return np.exp(-r * T) * np.mean(x, axis=1)

    def delta_gamma_batch(self, S, K, T, r, sigma, option_type, q=0.0, h=1e-4):  # :633-689
        S = np.asarray(S, dtype=np.float64)
        dn = self.price_batch(S - h, K, T, r, sigma, option_type, q)
        mid = self.price_batch(S, K, T, r, sigma, option_type, q)
        up = self.price_batch(S + h, K, T, r, sigma, option_type, q)
        return (up - dn) / (2 * h), (up - 2 * mid + dn) / (h**2)


# --------------------------------------------------------------------------
# Heston (src/pricing_models/heston.py:80-255)
# --------------------------------------------------------------------------
def heston_cf(u, S, K, T, r, q, kappa, theta, sigma_v, rho, v0):
    """:80-129 Gatheral formulation."""
    x = np.log(S / K) + (r - q) * T
    alpha = -0.5 * u * (u + 1j)
    beta = kappa - rho * sigma_v * 1j * u
    gamma = 0.5 * sigma_v**2
    d = np.sqrt(beta**2 - 4 * alpha * gamma)
    r_plus = (beta + d) / (sigma_v**2)
    r_minus = (beta - d) / (sigma_v**2)
    g = r_minus / r_plus
    exp_dT = np.exp(-d * T)
    C_ = kappa * (r_minus * T - (2 / sigma_v**2) * np.log((1 - g * exp_dT) / (1 - g)))
    D_ = r_minus * (1 - exp_dT) / (1 - g * exp_dT)
    return np.exp(C_ * theta + D_ * v0 + 1j * u * x)


def heston_price_european(S, K, T, r, q, option_type, kappa, theta, sigma_v, rho, v0):
    """:131-182 Lewis formula, quad on [0, 100]."""
    from scipy.integrate import quad

    if T <= 0:
        return max(S - K, 0) if option_type == "call" else max(K - S, 0)
    F = S * np.exp((r - q) * T)

    def integrand(u):
        cf = heston_cf(u - 0.5j, S, K, T, r, q, kappa, theta, sigma_v, rho, v0)
        return np.real(np.exp(-1j * u * np.log(K / F)) * cf / (u**2 + 0.25))

    integral, _ = quad(integrand, 0, 100, limit=100)
    call = S * np.exp(-q * T) - (np.sqrt(K * F) / np.pi) * np.exp(-r * T) * integral
    if option_type == "call":
        return max(call, 0.0)
    return max(call - S * np.exp(-q * T) + K * np.exp(-r * T), 0.0)


def heston_price_mc(S, K, T, r, q, option_type, kappa, theta, sigma_v, rho, v0, n_paths=100000, n_steps=252, seed=None):
    """:184-255 full-truncation Euler, legacy global RandomState, no antithetic."""
    if seed is not None:
        np.random.seed(seed)
    dt = T / n_steps
    sqrt_dt = np.sqrt(dt)
    log_S = np.full(n_paths, np.log(S))
    v = np.full(n_paths, v0)
    rho_sqrt = np.sqrt(1 - rho**2)
    for _ in range(n_steps):
        Z1 = np.random.standard_normal(n_paths)
        Z2 = rho * Z1 + rho_sqrt * np.random.standard_normal(n_paths)
        v_pos = np.maximum(v, 0)
        sqrt_v = np.sqrt(v_pos)
        log_S += (r - q - 0.5 * v_pos) * dt + sqrt_v * sqrt_dt * Z1
        v += kappa * (theta - v_pos) * dt + sigma_v * sqrt_v * sqrt_dt * Z2
        v = np.maximum(v, 0)
    st = np.exp(log_S)
    x = np.maximum(st - K, 0) if option_type == "call" else np.maximum(K - st, 0)
    return np.exp(-r * T) * np.mean(x)


def heston_simulate_paths(S, T, r, q, kappa, theta, sigma_v, rho, v0, n_paths=1000, n_steps=252, seed=None):
    """:257-305 -- (spot_paths, var_paths), each (n_paths, n_steps + 1); column 0 = (S, v0)."""
    if seed is not None:
        np.random.seed(seed)
    dt = T / n_steps
    sqrt_dt = np.sqrt(dt)
    rho_sqrt = np.sqrt(1 - rho**2)
    spot = np.zeros((n_paths, n_steps + 1))
    var = np.zeros((n_paths, n_steps + 1))
    spot[:, 0] = S
    var[:, 0] = v0
    log_S = np.log(S) * np.ones(n_paths)
    v = v0 * np.ones(n_paths)
    for t in range(1, n_steps + 1):
        Z1 = np.random.standard_normal(n_paths)
        Z2 = rho * Z1 + rho_sqrt * np.random.standard_normal(n_paths)
        v_pos = np.maximum(v, 0)
        sqrt_v = np.sqrt(v_pos)
        log_S += (r - q - 0.5 * v_pos) * dt + sqrt_v * sqrt_dt * Z1
        v += kappa * (theta - v_pos) * dt + sigma_v * sqrt_v * sqrt_dt * Z2
        v = np.maximum(v, 0)
        spot[:, t] = np.exp(log_S)
        var[:, t] = v
    return spot, var


# --------------------------------------------------------------------------
# Jump diffusion (src/pricing_models/jump_diffusion.py)
# --------------------------------------------------------------------------
def merton_kappa(mu_j, sigma_j):  # :61-67
    return np.exp(mu_j + 0.5 * sigma_j**2) - 1


def _merton_bs(S, K, T, r, sigma, option_type, q):  # :134-158
    if sigma <= 0 or T <= 0:
        if option_type == "call":
            return max(S * np.exp(-q * T) - K * np.exp(-r * T), 0)
        return max(K * np.exp(-r * T) - S * np.exp(-q * T), 0)
    d1 = (np.log(S / K) + (r - q + 0.5 * sigma**2) * T) / (sigma * np.sqrt(T))
    d2 = d1 - sigma * np.sqrt(T)
    if option_type == "call":
        return S * np.exp(-q * T) * norm.cdf(d1) - K * np.exp(-r * T) * norm.cdf(d2)
    return K * np.exp(-r * T) * norm.cdf(-d2) - S * np.exp(-q * T) * norm.cdf(-d1)


def merton_series(S, K, T, r, sigma, lambda_j, mu_j, sigma_j, option_type="call", q=0.0, n_terms=50):  # :69-132
    from scipy.special import factorial

    if T <= 0:
        return max(S - K, 0) if option_type == "call" else max(K - S, 0)
    kappa = merton_kappa(mu_j, sigma_j)
    lam_p = lambda_j * (1 + kappa)
    price = 0.0
    for n in range(n_terms):
        w = np.exp(-lam_p * T) * (lam_p * T) ** n / factorial(n)
        sigma_n = np.sqrt(sigma**2 + n * sigma_j**2 / T)
        r_n = r - lambda_j * kappa + n * np.log(1 + kappa) / T
        price += w * _merton_bs(S, K, T, r_n, sigma_n, option_type, q)
        if w < 1e-12:
            break
    return price


def merton_mc(S, K, T, r, sigma, lambda_j, mu_j, sigma_j, option_type="call", q=0.0, n_paths=100000, n_steps=252, seed=None):
    """:160-225 (legacy global RandomState; per-path Python loop for the jumps)."""
    if seed is not None:
        np.random.seed(seed)
    dt = T / n_steps
    kappa = merton_kappa(mu_j, sigma_j)
    drift = (r - q - lambda_j * kappa - 0.5 * sigma**2) * dt
    vol = sigma * np.sqrt(dt)
    log_S = np.full(n_paths, np.log(S))
    for _ in range(n_steps):
        log_S += drift + vol * np.random.standard_normal(n_paths)
        n_jumps = np.random.poisson(lambda_j * dt, n_paths)
        for i in np.nonzero(n_jumps)[0]:  # same draws in the same order as the reference's `for i in range(n_paths)`
            log_S[i] += np.sum(np.random.normal(mu_j, sigma_j, n_jumps[i]))
    st = np.exp(log_S)
    x = np.maximum(st - K, 0) if option_type == "call" else np.maximum(K - st, 0)
    return np.exp(-r * T) * np.mean(x)


def merton_simulate_path(S, T, r, sigma, lambda_j, mu_j, sigma_j, q=0.0, n_steps=252, seed=None):
    """:227-272 -- one path, scalar draws from the legacy global RandomState."""
    if seed is not None:
        np.random.seed(seed)
    dt = T / n_steps
    drift = (r - q - lambda_j * merton_kappa(mu_j, sigma_j) - 0.5 * sigma**2) * dt
    vol = sigma * np.sqrt(dt)
    path = np.zeros(n_steps + 1)
    path[0] = S
    log_S = np.log(S)
    for t in range(1, n_steps + 1):
        log_S += drift + vol * np.random.standard_normal()
        n_jumps = np.random.poisson(lambda_j * dt)
        if n_jumps > 0:
            log_S += np.sum(np.random.normal(mu_j, sigma_j, n_jumps))
        path[t] = np.exp(log_S)
    return path


def kou_kappa(p, eta1, eta2):  # :293-299
    return p * eta1 / (eta1 - 1) + (1 - p) * eta2 / (eta2 + 1) - 1


def kou_mc(S, K, T, r, sigma, lambda_j, p, eta1, eta2, option_type="call", q=0.0, n_paths=100000, n_steps=252, seed=None):
    """:325-372"""
    if seed is not None:
        np.random.seed(seed)
    dt = T / n_steps
    drift = (r - q - lambda_j * kou_kappa(p, eta1, eta2) - 0.5 * sigma**2) * dt
    vol = sigma * np.sqrt(dt)
    log_S = np.full(n_paths, np.log(S))
    for _ in range(n_steps):
        log_S += drift + vol * np.random.standard_normal(n_paths)
        n_jumps = np.random.poisson(lambda_j * dt, n_paths)
        total = np.sum(n_jumps)
        if total > 0:
            jumps = np.zeros(total)  # simulate_jump, :301-316
            u = np.random.uniform(0, 1, total)
            up = u < p
            jumps[up] = np.random.exponential(1 / eta1, np.sum(up))
            jumps[~up] = -np.random.exponential(1 / eta2, np.sum(~up))
            idx = 0
            for i in np.nonzero(n_jumps)[0]:
                log_S[i] += np.sum(jumps[idx: idx + n_jumps[i]])
                idx += n_jumps[i]
    st = np.exp(log_S)
    x = np.maximum(st - K, 0) if option_type == "call" else np.maximum(K - st, 0)
    return np.exp(-r * T) * np.mean(x)
