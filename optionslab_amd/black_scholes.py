"""Black-Scholes-Merton closed form: the accuracy anchor of every MC test
(reference: src/pricing_models/black_scholes.py:9-52; same argument order,
same ValueError conditions, T == 0 -> intrinsic)."""
import math
from typing import Literal


def _ncdf(x: float) -> float:
    return 0.5 * math.erfc(-x / math.sqrt(2.0))


def black_scholes(S: float, K: float, T: float, r: float, sigma: float,
                  option_type: Literal["call", "put"] = "call", q: float = 0.0) -> float:
    if S <= 0 or K <= 0 or T < 0 or sigma < 0:
        raise ValueError("Invalid input: all inputs must be positive, and T, sigma >= 0")
    if T == 0:
        return max(S - K, 0.0) if option_type == "call" else max(K - S, 0.0)
    root = sigma * math.sqrt(T)
    d1 = (math.log(S / K) + (r - q + 0.5 * sigma**2) * T) / root
    d2 = d1 - root
    fwd, strike = S * math.exp(-q * T), K * math.exp(-r * T)
    if option_type == "call":
        return fwd * _ncdf(d1) - strike * _ncdf(d2)
    if option_type == "put":
        return strike * _ncdf(-d2) - fwd * _ncdf(-d1)
    raise ValueError("option_type must be 'call' or 'put'")
