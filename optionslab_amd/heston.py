"""HestonPricer on the device step loop (reference: src/pricing_models/heston.py:41-305).

`price_monte_carlo` (full-truncation Euler, :184-255) runs on the GPU: two normals per step,
(ln S, v) in fp64 registers.  `price_european` (:131-182) is the reference's semi-analytic
Lewis/Gatheral quadrature -- scalar host arithmetic, kept so `HestonAdapter`-style callers and
accuracy checks have the same oracle the reference has; it is not a Monte Carlo path.
"""
from __future__ import annotations

import warnings
from dataclasses import dataclass
from typing import Tuple, Literal, Optional

import numpy as np

from . import _hip


@dataclass
class HestonPricer:
    kappa: float
    theta: float
    sigma_v: float
    rho: float
    v0: float

    def __post_init__(self):     # heston.py:61-78
        if self.kappa <= 0:
            raise ValueError("kappa must be positive")
        if self.theta <= 0:
            raise ValueError("theta must be positive")
        if self.sigma_v <= 0:
            raise ValueError("sigma_v must be positive")
        if not -1 <= self.rho <= 1:
            raise ValueError("rho must be in [-1, 1]")
        if self.v0 <= 0:
            raise ValueError("v0 must be positive")
        feller = 2 * self.kappa * self.theta - self.sigma_v**2
        if feller < 0:
            warnings.warn(f"Feller condition not satisfied (2κθ - σᵥ² = {feller:.4f} < 0). "
                          "Variance may hit zero in simulations.")

    def _characteristic_function(self, u, S, K, T, r, q):       # :80-129
        kappa, theta, sigma_v, rho, v0 = self.kappa, self.theta, self.sigma_v, self.rho, self.v0
        x = np.log(S / K) + (r - q) * T
        alpha = -0.5 * u * (u + 1j)
        beta = kappa - rho * sigma_v * 1j * u
        d = np.sqrt(beta**2 - 4 * alpha * (0.5 * sigma_v**2))
        r_minus = (beta - d) / (sigma_v**2)
        g = r_minus / ((beta + d) / (sigma_v**2))
        e = np.exp(-d * T)
        big_c = kappa * (r_minus * T - (2 / sigma_v**2) * np.log((1 - g * e) / (1 - g)))
        big_d = r_minus * (1 - e) / (1 - g * e)
        return np.exp(big_c * theta + big_d * v0 + 1j * u * x)

    def price_european(self, S: float, K: float, T: float, r: float, q: float = 0.0,
                       option_type: Literal["call", "put"] = "call") -> float:
        from scipy.integrate import quad

        if T <= 0:
            return max(S - K, 0) if option_type == "call" else max(K - S, 0)
        fwd = S * np.exp((r - q) * T)

        def integrand(u):
            cf = self._characteristic_function(u - 0.5j, S, K, T, r, q)
            return np.real(np.exp(-1j * u * np.log(K / fwd)) * cf / (u**2 + 0.25))

        integral, _ = quad(integrand, 0, 100, limit=100)
        call = S * np.exp(-q * T) - (np.sqrt(K * fwd) / np.pi) * np.exp(-r * T) * integral
        if option_type == "call":
            return max(call, 0.0)
        return max(call - S * np.exp(-q * T) + K * np.exp(-r * T), 0.0)

    def price_monte_carlo(self, S: float, K: float, T: float, r: float, q: float = 0.0,
                          option_type: Literal["call", "put"] = "call", n_paths: int = 100000, n_steps: int = 252,
                          seed: Optional[int] = None, antithetic: bool = False, return_error: bool = False):
        if n_paths < 1 or n_steps < 1:
            raise ValueError("n_paths and n_steps must be >= 1")
        s = seed if seed is not None else int(np.random.default_rng().integers(0, 2**31))
        st = _hip.heston(S, K, T, r, q, option_type == "call", self.kappa, self.theta, self.sigma_v, self.rho, self.v0,
                         n_paths, n_steps, s, antithetic)
        return (np.float64(st.price), float(st.std_error)) if return_error else np.float64(st.price)


    def simulate_paths(self, S: float, T: float, r: float, q: float = 0.0, n_paths: int = 1000, n_steps: int = 252,
                       seed: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
        """heston.py:257-305: (spot_paths, variance_paths), each (n_paths, n_steps + 1), column 0 = (S, v0).
        The states of price_monte_carlo's recursion for the same seed."""
        if n_paths < 1 or n_steps < 1:
            raise ValueError("n_paths and n_steps must be >= 1")
        s = seed if seed is not None else int(np.random.default_rng().integers(0, 2**31))
        return _hip.heston_paths(S, T, r, q, self.kappa, self.theta, self.sigma_v, self.rho, self.v0, n_paths, n_steps, s, path_major=True)


class HestonAdapter:
    """unified_greeks.py:74-104: sigma -> v0 = sigma^2, prices with the semi-analytic formula."""

    def __init__(self, heston_pricer):
        self.heston = heston_pricer
        self._original_v0 = heston_pricer.v0

    def price(self, S, K, T, r, sigma, option_type, q=0.0, **kwargs) -> float:
        self.heston.v0 = sigma**2
        try:
            return self.heston.price_european(S, K, T, r, q, option_type)
        finally:
            self.heston.v0 = self._original_v0


def greeks_heston(heston_pricer, S: float, K: float, T: float, r: float, sigma: float, option_type: str = "call", q: float = 0.0):
    """unified_greeks.py:375-388"""
    from .greeks import compute_greeks_unified

    return compute_greeks_unified(HestonAdapter(heston_pricer), S, K, T, r, sigma, option_type, q)
