"""AsianOption / BarrierOption / LookbackOption on the device step loop (reference: ExoticOptionBase._generate_paths
+ AsianOption, src/pricing_models/exotic_options.py:28-160, price_asian :558-572).

Same dataclass fields and ``price(n_paths, n_steps, avg_type, option_type)``
signature; the running average lives in registers, no (n_paths, n_steps) matrix
exists.  Like the reference there is no antithetic mirror unless asked for, and
the return value is a ``numpy.float64``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Literal, Optional

import numpy as np

from . import _hip
from .black_scholes import _ncdf


@dataclass
class AsianOption:
    S: float
    K: float
    T: float
    r: float
    sigma: float
    q: float = 0.0
    seed: Optional[int] = None

    def price(self, n_paths: int = 100000, n_steps: int = 252,
              avg_type: Literal["arithmetic", "geometric"] = "arithmetic",
              option_type: Literal["call", "put"] = "call", antithetic: bool = False,
              return_error: bool = False, precision: Literal["fp64", "fp32"] = "fp64"):
        """precision (additive, arithmetic average only): "fp64" = the reference's arithmetic (fp64 cumulative
        log-return, one fp64 exponential per monitoring date, :62-67); "fp32" = the opt-in fast kernel (one
        hardware fp32 exponential per date: ~2e-6 on the price, 1.6x faster)."""
        if n_paths < 1 or n_steps < 1:
            raise ValueError("n_paths and n_steps must be >= 1")
        if precision not in ("fp64", "fp32"):
            raise ValueError("precision must be 'fp64' or 'fp32'")
        # seed=None: the reference leaves the global RandomState unseeded (:51-52) => fresh draw
        seed = self.seed if self.seed is not None else int(np.random.default_rng().integers(0, 2**31))
        st = _hip.asian(self.S, self.K, self.T, self.r, self.sigma, self.q, option_type == "call",
                        avg_type != "arithmetic", n_paths, n_steps, seed, antithetic, fast=precision == "fp32")
        if return_error:
            return np.float64(st.price), float(st.std_error)
        return np.float64(st.price)

    def price_geometric_closed_form(self, option_type: Literal["call", "put"] = "call") -> float:
        """exotic_options.py:133-160"""
        v = self.sigma / math.sqrt(3)
        b = 0.5 * (self.r - self.q - self.sigma**2 / 6)
        d1 = (math.log(self.S / self.K) + (b + 0.5 * v**2) * self.T) / (v * math.sqrt(self.T))
        d2 = d1 - v * math.sqrt(self.T)
        grow, disc = math.exp((b - self.r) * self.T), math.exp(-self.r * self.T)
        if option_type == "call":
            return self.S * grow * _ncdf(d1) - self.K * disc * _ncdf(d2)
        return self.K * disc * _ncdf(-d2) - self.S * grow * _ncdf(-d1)


@dataclass
class BarrierOption:
    """exotic_options.py:163-224: knock-in / knock-out on discrete monitoring dates t = 0..M."""

    S: float
    K: float
    T: float
    r: float
    sigma: float
    q: float = 0.0
    seed: Optional[int] = None
    barrier: float = 0.0

    def price(self, n_paths: int = 100000, n_steps: int = 252,
              barrier_type: Literal["up-and-out", "up-and-in", "down-and-out", "down-and-in"] = "up-and-out",
              option_type: Literal["call", "put"] = "call", antithetic: bool = False, return_error: bool = False):
        if self.barrier <= 0:                       # :195-196
            raise ValueError("Barrier must be positive")
        if n_paths < 1 or n_steps < 1:
            raise ValueError("n_paths and n_steps must be >= 1")
        # the reference dispatches on startswith("up") / endswith("out") (:201-212)
        kind = (0 if barrier_type.startswith("up") else 2) + (0 if barrier_type.endswith("out") else 1)
        seed = self.seed if self.seed is not None else int(np.random.default_rng().integers(0, 2**31))
        st = _hip.barrier(self.S, self.K, self.T, self.r, self.sigma, self.q, option_type == "call", self.barrier, kind,
                          n_paths, n_steps, seed, antithetic)
        return (np.float64(st.price), float(st.std_error)) if return_error else np.float64(st.price)


@dataclass
class LookbackOption:
    """exotic_options.py:347-401: floating / fixed strike on the discrete path extrema (t = 0 included)."""

    S: float
    K: float
    T: float
    r: float
    sigma: float
    q: float = 0.0
    seed: Optional[int] = None

    def price(self, n_paths: int = 100000, n_steps: int = 252, lookback_type: Literal["floating", "fixed"] = "floating",
              option_type: Literal["call", "put"] = "call", antithetic: bool = False, return_error: bool = False):
        if n_paths < 1 or n_steps < 1:
            raise ValueError("n_paths and n_steps must be >= 1")
        seed = self.seed if self.seed is not None else int(np.random.default_rng().integers(0, 2**31))
        st = _hip.lookback(self.S, self.K, self.T, self.r, self.sigma, self.q, option_type == "call",
                           lookback_type != "floating", n_paths, n_steps, seed, antithetic)
        return (np.float64(st.price), float(st.std_error)) if return_error else np.float64(st.price)


@dataclass
class AmericanOption:
    """exotic_options.py:227-345: Longstaff-Schwartz least-squares Monte Carlo on stored device paths."""

    S: float
    K: float
    T: float
    r: float
    sigma: float
    q: float = 0.0
    seed: Optional[int] = None

    def price(self, n_paths: int = 50000, n_steps: int = 50, option_type: Literal["call", "put"] = "put",
              poly_degree: int = 3, return_error: bool = False):
        if n_paths < 1 or n_steps < 1:
            raise ValueError("n_paths and n_steps must be >= 1")
        seed = self.seed if self.seed is not None else int(np.random.default_rng().integers(0, 2**31))
        st = _hip.american_lsm(self.S, self.K, self.T, self.r, self.sigma, self.q, option_type == "call", n_paths, n_steps,
                               poly_degree, seed)
        return (np.float64(st.price), float(st.std_error)) if return_error else np.float64(st.price)

    def early_exercise_boundary(self, n_paths: int = 10000, n_steps: int = 50, option_type: Literal["call", "put"] = "put"):
        """exotic_options.py:309-345: (times, boundary): per date the 10th (put) / 90th (call) percentile of the
        in-the-money simulated prices, NaN where none is; selected on the device from the LSM path set."""
        if n_paths < 1 or n_steps < 1:
            raise ValueError("n_paths and n_steps must be >= 1")
        seed = self.seed if self.seed is not None else int(np.random.default_rng().integers(0, 2**31))
        boundary = _hip.exercise_boundary(self.S, self.K, self.T, self.r, self.sigma, self.q, option_type == "call", n_paths, n_steps, seed)
        return np.linspace(0, self.T, n_steps + 1), boundary


def price_american(S: float, K: float, T: float, r: float, sigma: float, option_type: str = "put", n_paths: int = 50000,
                   seed: int = None) -> float:
    """exotic_options.py:593-606"""
    return AmericanOption(S=S, K=K, T=T, r=r, sigma=sigma, seed=seed).price(n_paths=n_paths, option_type=option_type)


@dataclass
class AutocallableOption:
    """exotic_options.py:404-491 (snowball note; barriers relative to spot; result = fraction of notional)."""

    S: float
    K: float
    T: float
    r: float
    sigma: float
    q: float = 0.0
    seed: Optional[int] = None
    autocall_barrier: float = 1.0
    coupon_barrier: float = 0.8
    coupon_rate: float = 0.10
    ki_barrier: float = 0.6

    def price(self, n_paths: int = 100000, n_steps: int = 252, observation_freq: int = 21, antithetic: bool = False,
              return_error: bool = False, **kwargs):
        if n_paths < 1 or n_steps < 1:
            raise ValueError("n_paths and n_steps must be >= 1")
        seed = self.seed if self.seed is not None else int(np.random.default_rng().integers(0, 2**31))
        st = _hip.autocallable(self.S, self.T, self.r, self.sigma, self.q, self.autocall_barrier, self.coupon_barrier,
                               self.coupon_rate, self.ki_barrier, observation_freq, n_paths, n_steps, seed, antithetic)
        return (np.float64(st.price), float(st.std_error)) if return_error else np.float64(st.price)


@dataclass
class CliquetOption:
    """exotic_options.py:494-554 (ratchet: sum of locally capped/floored period returns, globally clipped)."""

    S: float
    K: float
    T: float
    r: float
    sigma: float
    q: float = 0.0
    seed: Optional[int] = None
    local_cap: float = 0.05
    local_floor: float = -0.05
    global_cap: float = 0.30
    global_floor: float = 0.0

    def price(self, n_paths: int = 100000, n_steps: int = 252, n_periods: int = 12, antithetic: bool = False,
              return_error: bool = False, **kwargs):
        if n_paths < 1 or n_steps < 1:
            raise ValueError("n_paths and n_steps must be >= 1")
        seed = self.seed if self.seed is not None else int(np.random.default_rng().integers(0, 2**31))
        st = _hip.cliquet(self.S, self.T, self.r, self.sigma, self.q, self.local_cap, self.local_floor, self.global_cap,
                          self.global_floor, n_periods, n_paths, n_steps, seed, antithetic)
        return (np.float64(st.price), float(st.std_error)) if return_error else np.float64(st.price)


def price_barrier(S: float, K: float, T: float, r: float, sigma: float, barrier: float, barrier_type: str = "up-and-out",
                  option_type: str = "call", n_paths: int = 100000, seed: int = None) -> float:
    """exotic_options.py:575-590"""
    return BarrierOption(S=S, K=K, T=T, r=r, sigma=sigma, barrier=barrier, seed=seed).price(
        n_paths=n_paths, barrier_type=barrier_type, option_type=option_type)


def price_asian(S: float, K: float, T: float, r: float, sigma: float, avg_type: str = "arithmetic",
                option_type: str = "call", n_paths: int = 100000, seed: int = None) -> float:
    return AsianOption(S=S, K=K, T=T, r=r, sigma=sigma, seed=seed).price(
        n_paths=n_paths, avg_type=avg_type, option_type=option_type)
