"""Merton and Kou jump-diffusion pricers on the device step loop (reference:
src/pricing_models/jump_diffusion.py:38-372).

`price_monte_carlo` runs on the GPU (one Philox block per step: diffusion normal, Poisson
uniform, jump draws).  `MertonJumpDiffusion.price` is the reference's Poisson-weighted
Black-Scholes series (:69-132) -- scalar host arithmetic, the accuracy anchor of the MC.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Literal, Optional

import numpy as np

from . import _hip
from .black_scholes import _ncdf


def _seed(seed):
    return seed if seed is not None else int(np.random.default_rng().integers(0, 2**31))


@dataclass
class MertonJumpDiffusion:
    lambda_j: float
    mu_j: float
    sigma_j: float

    def __post_init__(self):            # :54-59
        if self.lambda_j < 0:
            raise ValueError("lambda_j must be non-negative")
        if self.sigma_j < 0:
            raise ValueError("sigma_j must be non-negative")

    @property
    def kappa(self) -> float:           # :61-67
        return math.exp(self.mu_j + 0.5 * self.sigma_j**2) - 1

    def price(self, S, K, T, r, sigma, option_type: Literal["call", "put"] = "call", q: float = 0.0, n_terms: int = 50) -> float:
        """Merton's series (:69-132): sum_n Poisson(lambda' T; n) * BS(r_n, sigma_n)."""
        if T <= 0:
            return max(S - K, 0) if option_type == "call" else max(K - S, 0)
        kappa = self.kappa
        lam_p = self.lambda_j * (1 + kappa)
        total = 0.0
        for n in range(n_terms):
            weight = math.exp(-lam_p * T) * (lam_p * T) ** n / math.factorial(n)
            sigma_n = math.sqrt(sigma**2 + n * self.sigma_j**2 / T)
            r_n = r - self.lambda_j * kappa + n * math.log(1 + kappa) / T
            total += weight * self._black_scholes(S, K, T, r_n, sigma_n, option_type, q)
            if weight < 1e-12:          # :128-130 (checked after adding the term)
                break
        return total

    @staticmethod
    def _black_scholes(S, K, T, r, sigma, option_type, q) -> float:     # :134-158
        fwd, strike = S * math.exp(-q * T), K * math.exp(-r * T)
        if sigma <= 0 or T <= 0:
            return max(fwd - strike, 0) if option_type == "call" else max(strike - fwd, 0)
        root = sigma * math.sqrt(T)
        d1 = (math.log(S / K) + (r - q + 0.5 * sigma**2) * T) / root
        d2 = d1 - root
        if option_type == "call":
            return fwd * _ncdf(d1) - strike * _ncdf(d2)
        return strike * _ncdf(-d2) - fwd * _ncdf(-d1)

    def price_monte_carlo(self, S, K, T, r, sigma, option_type: Literal["call", "put"] = "call", q: float = 0.0,
                          n_paths: int = 100000, n_steps: int = 252, seed: Optional[int] = None, return_error: bool = False):
        if n_paths < 1 or n_steps < 1:
            raise ValueError("n_paths and n_steps must be >= 1")
        st = _hip.jump_diffusion(S, K, T, r, sigma, q, option_type == "call", False, self.lambda_j, self.mu_j, self.sigma_j, 0.0,
                                 n_paths, n_steps, _seed(seed))
        return (np.float64(st.price), float(st.std_error)) if return_error else np.float64(st.price)


    def simulate_paths(self, S, T, r, sigma, q: float = 0.0, n_paths: int = 1, n_steps: int = 252,
                       seed: Optional[int] = None) -> np.ndarray:
        """(n_paths, n_steps + 1) price paths of price_monte_carlo's recursion for the same seed, column 0 = S."""
        if n_paths < 1 or n_steps < 1:
            raise ValueError("n_paths and n_steps must be >= 1")
        return _hip.jump_paths(S, T, r, sigma, q, False, self.lambda_j, self.mu_j, self.sigma_j, 0.0, n_paths, n_steps, _seed(seed),
                               path_major=True)

    def simulate_path(self, S, T, r, sigma, q: float = 0.0, n_steps: int = 252, seed: Optional[int] = None) -> np.ndarray:
        """jump_diffusion.py:227-272: one path with jumps, shape (n_steps + 1,)."""
        return self.simulate_paths(S, T, r, sigma, q, 1, n_steps, seed)[0]


@dataclass
class KouJumpDiffusion:
    lambda_j: float
    p: float
    eta1: float
    eta2: float

    def __post_init__(self):            # :285-291
        if not 0 <= self.p <= 1:
            raise ValueError("p must be in [0, 1]")
        if self.eta1 <= 1:
            raise ValueError("eta1 must be > 1 for finite mean")
        if self.eta2 <= 0:
            raise ValueError("eta2 must be positive")

    @property
    def kappa(self) -> float:           # :293-299
        return self.p * self.eta1 / (self.eta1 - 1) + (1 - self.p) * self.eta2 / (self.eta2 + 1) - 1

    def price_monte_carlo(self, S, K, T, r, sigma, option_type: Literal["call", "put"] = "call", q: float = 0.0,
                          n_paths: int = 100000, n_steps: int = 252, seed: Optional[int] = None, return_error: bool = False):
        if n_paths < 1 or n_steps < 1:
            raise ValueError("n_paths and n_steps must be >= 1")
        st = _hip.jump_diffusion(S, K, T, r, sigma, q, option_type == "call", True, self.lambda_j, self.p, self.eta1, self.eta2,
                                 n_paths, n_steps, _seed(seed))
        return (np.float64(st.price), float(st.std_error)) if return_error else np.float64(st.price)

    def simulate_paths(self, S, T, r, sigma, q: float = 0.0, n_paths: int = 1, n_steps: int = 252,
                       seed: Optional[int] = None) -> np.ndarray:
        """(n_paths, n_steps + 1) price paths of price_monte_carlo's recursion (additive: the reference has none for Kou)."""
        if n_paths < 1 or n_steps < 1:
            raise ValueError("n_paths and n_steps must be >= 1")
        return _hip.jump_paths(S, T, r, sigma, q, True, self.lambda_j, self.p, self.eta1, self.eta2, n_paths, n_steps, _seed(seed),
                               path_major=True)


class JumpDiffusionAdapter:
    """unified_greeks.py:155-175: PricerProtocol over a jump-diffusion model's series price."""

    def __init__(self, jd_model):
        self.jd = jd_model

    def price(self, S, K, T, r, sigma, option_type, q=0.0, **kwargs) -> float:
        return self.jd.price(S, K, T, r, sigma, option_type, q)
