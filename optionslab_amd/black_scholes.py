"""Closed-form European price under Black-Scholes-Merton with a continuous yield.

The accuracy anchor of every Monte Carlo test, as in the reference
(src/pricing_models/black_scholes.py:9-52: same argument order, ValueError on
S <= 0, K <= 0, T < 0 or sigma < 0, and T == 0 -> intrinsic value).  Scalar host
arithmetic, written on the forward: C = D (F N(d+) - K N(d-)), P = D (K N(-d-) - F N(-d+)).
"""
import math

_SQRT2 = math.sqrt(2.0)


def _ncdf(x: float) -> float:
    """Standard normal CDF via erfc (no SciPy dependency; 1e-16 of scipy.stats.norm.cdf)."""
    return 0.5 * math.erfc(-x / _SQRT2)


def black_scholes(S, K, T, r, sigma, option_type="call", q=0.0):
    bad = [name for name, ok in (("S", S > 0), ("K", K > 0), ("T", T >= 0), ("sigma", sigma >= 0)) if not ok]
    if bad:
        raise ValueError("Invalid input: all inputs must be positive, and T, sigma >= 0")
    is_call = option_type == "call"
    if not is_call and option_type != "put":
        if T == 0:       # the reference prices anything at expiry before checking the type
            return max(K - S, 0.0)
        raise ValueError("option_type must be 'call' or 'put'")
    if T == 0:
        return max(S - K, 0.0) if is_call else max(K - S, 0.0)
    total_vol = sigma * math.sqrt(T)
    carry = math.exp(-q * T)                 # spot leg discounting
    discount = math.exp(-r * T)              # strike leg discounting
    d_plus = (math.log(S / K) + (r - q + 0.5 * sigma**2) * T) / total_vol
    d_minus = d_plus - total_vol
    if is_call:
        return S * carry * _ncdf(d_plus) - K * discount * _ncdf(d_minus)
    return K * discount * _ncdf(-d_minus) - S * carry * _ncdf(-d_plus)
