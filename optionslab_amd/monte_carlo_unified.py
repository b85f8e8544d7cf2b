"""MonteCarloPricerUni on the MI355X path engine: single contracts, CRN delta/gamma and
batches of independent contracts in one launch.

Mirror of the reference class (src/pricing_models/monte_carlo_unified.py:236-689): same
constructor and attributes, same validation (`InputValidationError`), same error wrapping
(`MonteCarloError("Monte Carlo pricing failed: ...")`), `price` -> float,
`delta_gamma` -> (float, float) with h = 1e-4 and common random numbers,
`price_batch` -> ndarray, `delta_gamma_batch` -> (ndarray, ndarray).  Every backend flag
(`use_numba`, `use_gpu`) ends on the GPU; the CuPy branch of the reference (:345-402), which
materialises the (n, sims, steps) normal tensor three times, is what the register-resident
kernel replaces.  The `MLSurrogate` regression model of that module is out of scope (not Monte Carlo).
"""
from __future__ import annotations

import threading
from typing import Literal, Optional, Tuple, Union

import numpy as np

from . import _hip

NUMBA_AVAILABLE = False
GPU_AVAILABLE = True       # this class always runs on the HIP device (or raises)

__all__ = ["MonteCarloPricerUni", "InputValidationError", "MonteCarloError", "NUMBA_AVAILABLE", "GPU_AVAILABLE"]


class InputValidationError(Exception):      # monte_carlo_unified.py:88-91 (module-local class)
    pass


class MonteCarloError(Exception):           # :94-97
    pass


class MonteCarloPricerUni:
    def __init__(self, num_simulations: int = 100_000, num_steps: int = 100, seed: Optional[int] = None,
                 use_numba: bool = True, use_gpu: bool = False) -> None:
        if num_simulations <= 0 or num_steps <= 0:      # :277-280
            raise InputValidationError("num_simulations and num_steps must be positive integers")
        self.num_simulations = num_simulations
        self.num_steps = num_steps
        self.seed = seed if seed is not None else int(np.random.default_rng().integers(0, 2**31))
        self.rng = np.random.default_rng(seed)          # :287 (feeds delta_gamma's seed draw)
        self.use_numba = False
        self.use_gpu = True
        self._lock = threading.RLock()

    # ------------------------------------------------------------------ single contract
    def price(self, S: float, K: float, T: float, r: float, sigma: float, option_type: Literal["call", "put"],
              q: float = 0.0, seed: Optional[int] = None) -> float:
        if S <= 0 or K <= 0 or T <= 0 or sigma < 0:     # :485-488
            raise InputValidationError("S, K, T must be positive; sigma must be non-negative")
        if option_type not in {"call", "put"}:          # :489-490
            raise InputValidationError("option_type must be 'call' or 'put'")
        try:
            st = _hip.european(S, K, T, r, sigma, q, option_type == "call", self.num_simulations, self.num_steps,
                               seed if seed is not None else self.seed, True)
            return float(st.price)
        except Exception as e:                          # :510-511
            raise MonteCarloError(f"Monte Carlo pricing failed: {e}")

    def delta_gamma(self, S: float, K: float, T: float, r: float, sigma: float, option_type: Literal["call", "put"],
                    q: float = 0.0, h: float = 1e-4, seed: Optional[int] = None) -> Tuple[float, float]:
        """Central differences on S -/+ h with one seed (:513-560); the three contracts are priced
        on the same normals in ONE launch."""
        if seed is None:
            seed = int(self.rng.integers(0, 2**31))     # :549-550
        for s_ in (S + h, S, S - h):                    # each price() validates (:553-555)
            if s_ <= 0 or K <= 0 or T <= 0 or sigma < 0:
                raise InputValidationError("S, K, T must be positive; sigma must be non-negative")
        if option_type not in {"call", "put"}:
            raise InputValidationError("option_type must be 'call' or 'put'")
        try:
            call = option_type == "call"
            up, mid, dn = _hip.european_batch([(S + h, K, T, r, sigma, q, call), (S, K, T, r, sigma, q, call),
                                               (S - h, K, T, r, sigma, q, call)], self.num_simulations, self.num_steps, seed)
        except Exception as e:
            raise MonteCarloError(f"Monte Carlo pricing failed: {e}")
        delta = (up.price - dn.price) / (2 * h)         # :557
        gamma = (up.price - 2 * mid.price + dn.price) / (h**2)   # :558
        return delta, gamma

    # ------------------------------------------------------------------ batches
    @staticmethod
    def _arrays(S_vals, K_vals, T_vals, r_vals, sigma_vals, q_vals):
        S, K, T, r, v = (np.asarray(a, dtype=np.float64) for a in (S_vals, K_vals, T_vals, r_vals, sigma_vals))
        q = np.full_like(S, q_vals) if isinstance(q_vals, (int, float)) else np.asarray(q_vals, dtype=np.float64)  # :608-611
        return S, K, T, r, v, q

    def price_batch(self, S_vals, K_vals, T_vals, r_vals, sigma_vals, option_type: Literal["call", "put"],
                    q_vals: Union[float, np.ndarray] = 0.0) -> np.ndarray:
        """One launch for all contracts; contract j draws its own stream (tag j) under the
        pricer's seed, so repeated calls reuse the normals as the reference does (:300, :615-617)."""
        S, K, T, r, v, q = self._arrays(S_vals, K_vals, T_vals, r_vals, sigma_vals, q_vals)
        if len(S) == 0:
            return np.empty(0, dtype=np.float64)
        res = _hip.european_multi(S, K, T, r, v, q, option_type == "call", self.num_simulations, self.num_steps, self.seed)
        return res["price"].copy()

    def delta_gamma_batch(self, S_vals, K_vals, T_vals, r_vals, sigma_vals, option_type: Literal["call", "put"],
                          q_vals: Union[float, np.ndarray] = 0.0, h: float = 1e-4) -> Tuple[np.ndarray, np.ndarray]:
        """(:633-689) three batches at S-h, S, S+h under one seed, fused into ONE launch of 3n
        contracts where the three copies of contract j share stream tag j."""
        S, K, T, r, v, q = self._arrays(S_vals, K_vals, T_vals, r_vals, sigma_vals, q_vals)
        n = len(S)
        if n == 0:
            return np.empty(0), np.empty(0)
        tags = np.tile(np.arange(n, dtype=np.uint32), 3)
        res = _hip.european_multi(np.concatenate([S - h, S, S + h]), np.tile(K, 3), np.tile(T, 3), np.tile(r, 3), np.tile(v, 3),
                                  np.tile(q, 3), option_type == "call", self.num_simulations, self.num_steps, self.seed, True, tags)
        p = res["price"]
        down, mid, up = p[:n], p[n:2 * n], p[2 * n:]
        return (up - down) / (2 * h), (up - 2 * mid + down) / (h**2)    # :684-687
