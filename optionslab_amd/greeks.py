"""Finite-difference Greeks for any pricer that has ``.price(S, K, T, r, sigma,
option_type, q=0.0, **kw)`` -- the reference's ``compute_greeks_unified``
(src/greeks/unified_greeks.py:235-367) with the same bumps, the same memo on the
parameter tuple, the same OrderedDict and the same GreeksError wrapping.

For the device pricer the 8 / 14 evaluations are fused into one launch
(olmc_european_greeks_fd); ``fused=False`` forces the literal one-launch-per-
evaluation form, which gives the same numbers.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Literal, Optional, Protocol, runtime_checkable

from .exceptions import GreeksError

__all__ = ["PricerProtocol", "compute_greeks_unified", "ExoticAdapter", "fd_steps"]


@runtime_checkable
class PricerProtocol(Protocol):  # unified_greeks.py:45-66
    def price(self, S: float, K: float, T: float, r: float, sigma: float,
              option_type: Literal["call", "put"], q: float = 0.0, **kwargs) -> float: ...


def fd_steps(S: float):
    """(h_S, h_sigma, h_r, h_T), unified_greeks.py:274-277."""
    return max(1e-4, 0.01 * S), max(1e-4, 0.01), 1e-4, 1 / 365.0


class ExoticAdapter:
    """unified_greeks.py:177-227: re-parameterise an exotic option object and call
    its ``price(n_paths=, n_steps=, **kw)``."""

    def __init__(self, exotic_option, n_paths: int = 50000, n_steps: int = 252, **exotic_kwargs):
        self.exotic = exotic_option
        self.n_paths = n_paths
        self.n_steps = n_steps
        self.exotic_kwargs = exotic_kwargs

    def price(self, S, K, T, r, sigma, option_type, q=0.0, **kwargs) -> float:
        ex = self.exotic
        ex.S, ex.K, ex.T, ex.r, ex.sigma, ex.q = S, K, T, r, sigma, q
        kw = {**self.exotic_kwargs, **kwargs}
        kw.setdefault("option_type", option_type)
        return ex.price(n_paths=self.n_paths, n_steps=self.n_steps, **kw)

    # -- additive: the 8 / 14 evaluations of compute_greeks_unified in ONE launch where the device has a fused kernel for the payoff --
    #    the Asian (arithmetic at the reference's precision, or geometric), barrier and lookback options (the payoffs streamlit_app/pages/
    #    7_Exotic_Options.py:266-284 asks Greeks of) -- and the option has a fixed seed (an unseeded one draws fresh normals per
    #    evaluation, as the reference's does: nothing to share).  Same bumps, same formulas, same normals as the 8 / 14 price() calls.
    def _fused_plan(self):
        """(kind, payoff code, barrier level) of the fused kernel that prices this adapter's option, or None."""
        from .exotic import AsianOption, BarrierOption, LookbackOption
        ex, kw = self.exotic, self.exotic_kwargs
        if ex.seed is None or not (1 <= self.n_paths <= 1 << 26) or self.n_steps < 1:
            return None
        if type(ex) is AsianOption:
            geometric = kw.get("avg_type", "arithmetic") != "arithmetic"             # AsianOption.price: anything but "arithmetic" is geometric
            ok = set(kw) <= {"avg_type", "antithetic", "precision", "option_type"} and kw.get("precision", "fp64") in (("fp64", "fp32") if geometric else ("fp64",))
            return ("asian", 1 if geometric else 0, 0.0) if ok else None
        if type(ex) is BarrierOption:
            kind = kw.get("barrier_type", "up-and-out")
            if not (set(kw) <= {"barrier_type", "antithetic", "option_type"} and isinstance(kind, str) and ex.barrier > 0):
                return None
            return ("extrema", (0 if kind.startswith("up") else 2) + (0 if kind.endswith("out") else 1), float(ex.barrier))      # exotic_options.py:201-212
        if type(ex) is LookbackOption:
            if not set(kw) <= {"lookback_type", "antithetic", "option_type"}:
                return None
            return ("extrema", 4 if kw.get("lookback_type", "floating") == "floating" else 5, 0.0)
        return None

    def _can_fuse(self, pricer_kwargs) -> bool:
        return not pricer_kwargs and self._fused_plan() is not None

    def _fused_greeks(self, S, K, T, r, sigma, option_type, q, include_second_order, seed=None):
        import numpy as np

        from . import _hip
        ex, kw = self.exotic, self.exotic_kwargs
        kind, payoff, level = self._fused_plan()
        is_call, anti = kw.get("option_type", option_type) == "call", bool(kw.get("antithetic", False))
        if kind == "asian":
            vals, _ = _hip.asian_greeks_fd(S, K, T, r, sigma, q, is_call, self.n_paths, self.n_steps, ex.seed, anti, include_second_order, want_evals=False,
                                           geometric=payoff == 1)
        else:
            vals, _ = _hip.extrema_greeks_fd(S, K, T, r, sigma, q, is_call, payoff, level, self.n_paths, self.n_steps, ex.seed, anti, include_second_order,
                                             want_evals=False)
        ex.S, ex.K, ex.T, ex.r, ex.sigma, ex.q = S, K, T, r, sigma, q
        keys = ("price", "delta", "gamma", "vega", "theta", "rho", "vanna", "charm", "vomma")
        return OrderedDict((k, np.float64(v)) for k, v in zip(keys if include_second_order else keys[:6], vals))


def compute_greeks_unified(pricer: PricerProtocol, S: float, K: float, T: float, r: float, sigma: float,
                           option_type: Literal["call", "put"] = "call", q: float = 0.0,
                           include_second_order: bool = True, fused: Optional[bool] = None,
                           **pricer_kwargs) -> "OrderedDict[str, float]":
    try:
        can_fuse = hasattr(pricer, "_fused_greeks") and T > 0 and set(pricer_kwargs) <= {"seed"}
        if can_fuse and hasattr(pricer, "_can_fuse"):                  # an adapter knows whether ITS payoff has a fused kernel
            can_fuse = pricer._can_fuse(pricer_kwargs)
        if fused is None:
            fused = can_fuse
        if fused:
            if not can_fuse:
                raise ValueError("fused Greeks need the device MonteCarloPricer (or an ExoticAdapter over a seeded Asian, barrier or lookback option), T > 0 and no extra pricer kwargs")
            return pricer._fused_greeks(S, K, T, r, sigma, option_type, q, include_second_order,
                                        pricer_kwargs.get("seed"))

        h_S, h_v, h_r, h_T = fd_steps(S)
        memo = {}

        def P(S_=S, T_=T, r_=r, v_=sigma):
            key = (S_, K, T_, r_, v_, q)
            if key not in memo:
                memo[key] = pricer.price(S_, K, T_, r_, v_, option_type, q, **pricer_kwargs)
            return memo[key]

        mid = P()
        s_up, s_dn = P(S_=S + h_S), P(S_=S - h_S)
        delta = (s_up - s_dn) / (2 * h_S)
        gamma = (s_up - 2 * mid + s_dn) / (h_S**2)
        v_up, v_dn = P(v_=sigma + h_v), P(v_=sigma - h_v)
        vega = (v_up - v_dn) / (2 * h_v)
        theta = (P(T_=T - h_T) - mid) / h_T if T > h_T else -mid / max(T, 1e-6)
        r_up, r_dn = P(r_=r + h_r), P(r_=r - h_r)
        rho = (r_up - r_dn) / (2 * h_r)
        out = OrderedDict([("price", mid), ("delta", delta), ("gamma", gamma), ("vega", vega),
                           ("theta", theta), ("rho", rho)])
        if include_second_order:
            uu, ud = P(S_=S + h_S, v_=sigma + h_v), P(S_=S + h_S, v_=sigma - h_v)
            du, dd = P(S_=S - h_S, v_=sigma + h_v), P(S_=S - h_S, v_=sigma - h_v)
            out["vanna"] = (uu - ud - du + dd) / (4 * h_S * h_v)
            if T > h_T:
                delta_earlier = (P(S_=S + h_S, T_=T - h_T) - P(S_=S - h_S, T_=T - h_T)) / (2 * h_S)
                out["charm"] = (delta_earlier - delta) / h_T
            else:
                out["charm"] = 0.0
            out["vomma"] = (v_up - 2 * mid + v_dn) / (h_v**2)
        return out
    except Exception as e:  # unified_greeks.py:366-367
        raise GreeksError(f"Failed to compute unified Greeks: {str(e)}") from e
