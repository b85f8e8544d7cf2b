"""optionslab_amd -- MI355X-native Monte Carlo path engine behind the OptionsLab
``MonteCarloPricer.price()`` / ``compute_greeks_unified()`` API.

Importing the package touches neither the GPU nor libolmc.so; the first compute
call loads the library and raises ``AccelerationError(backend="hip")`` if it (or
a device) is missing -- there is no CPU fallback.
"""
from .black_scholes import black_scholes
from .exceptions import AccelerationError, ConvergenceError, GreeksError, InputValidationError, MonteCarloError
from .exotic import AmericanOption, price_american, AsianOption, AutocallableOption, BarrierOption, CliquetOption, LookbackOption, price_asian, price_barrier
from .greeks import ExoticAdapter, PricerProtocol, compute_greeks_unified
from .heston import HestonAdapter, HestonPricer, greeks_heston
from .jump_diffusion import JumpDiffusionAdapter, KouJumpDiffusion, MertonJumpDiffusion
from .monte_carlo import NUMBA_AVAILABLE, MCMethod, MCResult, MonteCarloPricer
from .monte_carlo_unified import MonteCarloPricerUni
from .simulation import (hip_available, simulate_gbm_hip, simulate_gbm_hip_fast, simulate_gbm_paths_hip,
                         simulate_gbm_qmc_antithetic_hip, simulate_gbm_qmc_hip)
from . import sharding  # noqa: E402  (torch is imported lazily inside)

__version__ = "0.2.0"

__all__ = [
    "MonteCarloPricer", "MonteCarloPricerUni", "MCMethod", "MCResult", "NUMBA_AVAILABLE", "compute_greeks_unified", "PricerProtocol",
    "ExoticAdapter", "HestonPricer", "HestonAdapter", "greeks_heston", "MertonJumpDiffusion", "KouJumpDiffusion", "JumpDiffusionAdapter", "AsianOption", "BarrierOption", "LookbackOption", "AmericanOption", "price_american", "AutocallableOption", "CliquetOption", "price_asian", "price_barrier", "black_scholes", "simulate_gbm_hip", "simulate_gbm_hip_fast", "simulate_gbm_paths_hip", "simulate_gbm_qmc_hip", "simulate_gbm_qmc_antithetic_hip",
    "hip_available", "MonteCarloError", "InputValidationError", "ConvergenceError", "AccelerationError", "GreeksError",
]


def __getattr__(name):
    """`HIP_AVAILABLE`: availability flag in the style of the reference's NUMBA_AVAILABLE / GPU_AVAILABLE
    (src/pricing_models/__init__.py:52-59), evaluated on first access because it touches the device."""
    if name == "HIP_AVAILABLE":
        return hip_available()
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
