"""Import-path compatibility: make the reference's own import statements resolve to this engine.

    import optionslab_amd.compat as compat
    compat.install()
    from src.pricing_models.monte_carlo import MonteCarloPricer, MCMethod      # the HIP pricer
    from src.greeks.unified_greeks import compute_greeks_unified

`install()` registers the modules of the hot path under the reference's names
(src/pricing_models/monte_carlo.py, src/simulation, src/greeks/unified_greeks.py, ...), so the
Streamlit page (streamlit_app/pages/1_MonteCarlo_Basic.py:139-160) and tests/test_monte_carlo.py
run unchanged.  When the real OptionsLab `src` package is importable only the hot-path submodules
are overridden; everything else of OptionsLab stays the reference's.  `uninstall()` restores.
"""
from __future__ import annotations

import importlib
import sys
import types

from .black_scholes import black_scholes as _black_scholes
from . import exceptions as _exc
from . import exotic as _exotic
from . import greeks as _greeks
from . import heston as _heston
from . import jump_diffusion as _jump
from . import monte_carlo as _mc
from . import monte_carlo_unified as _uni
from . import simulation as _sim

_installed: dict = {}
_package_attrs: list = []        # (package, name, previous value or _MISSING): names re-exported at package level
_MISSING = object()

# What the reference's package __init__ files re-export from the modules above
# (src/pricing_models/__init__.py:23-66, src/greeks/__init__.py:11-23, src/exceptions/__init__.py).
_PACKAGE_EXPORTS = {
    "src.pricing_models": ["src.pricing_models.monte_carlo", "src.pricing_models.monte_carlo_unified", "src.pricing_models.black_scholes",
                           "src.pricing_models.exotic_options", "src.pricing_models.heston", "src.pricing_models.jump_diffusion"],
    "src.greeks": ["src.greeks.unified_greeks"],
    "src.exceptions": ["src.exceptions.montecarlo_exceptions", "src.exceptions.greek_exceptions"],
}


def _module(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__optionslab_amd__ = True
    return m


def _targets() -> dict:
    sim = _module("src.simulation", simulate_gbm_hip=_sim.simulate_gbm_hip, simulate_gbm_hip_fast=_sim.simulate_gbm_hip_fast,
                  # the reference's backend names resolve to the device backend (same contract, src/simulation/__init__.py:5-6)
                  simulate_gbm_numpy=_sim.simulate_gbm_hip, simulate_gbm_numpy_fast=_sim.simulate_gbm_hip_fast,
                  simulate_gbm_paths=_sim.simulate_gbm_paths_hip, simulate_gbm_paths_hip=_sim.simulate_gbm_paths_hip,
                  simulate_gbm_numba=_sim.simulate_gbm_hip, simulate_gbm_qmc=_sim.simulate_gbm_qmc_hip,
                  simulate_gbm_qmc_antithetic=_sim.simulate_gbm_qmc_antithetic_hip,
                  NUMBA_AVAILABLE=False)
    sim_numpy = _module("src.simulation.gbm_numpy", simulate_gbm_numpy=_sim.simulate_gbm_hip, simulate_gbm_numpy_fast=_sim.simulate_gbm_hip_fast,
                        simulate_gbm_paths=_sim.simulate_gbm_paths_hip)
    sim_numba = _module("src.simulation.gbm_numba", simulate_gbm_numba=_sim.simulate_gbm_hip, NUMBA_AVAILABLE=False)
    sim_qmc = _module("src.simulation.gbm_qmc", simulate_gbm_qmc=_sim.simulate_gbm_qmc_hip,
                      simulate_gbm_qmc_antithetic=_sim.simulate_gbm_qmc_antithetic_hip)
    return {
        "src.simulation.gbm_numpy": sim_numpy, "src.simulation.gbm_numba": sim_numba, "src.simulation.gbm_qmc": sim_qmc,
        "src.pricing_models.monte_carlo": _module("src.pricing_models.monte_carlo", MonteCarloPricer=_mc.MonteCarloPricer,
                                                  MCMethod=_mc.MCMethod, MCResult=_mc.MCResult, NUMBA_AVAILABLE=False),
        "src.pricing_models.monte_carlo_unified": _module("src.pricing_models.monte_carlo_unified", MonteCarloPricerUni=_uni.MonteCarloPricerUni,
                                                          InputValidationError=_uni.InputValidationError, MonteCarloError=_uni.MonteCarloError,
                                                          NUMBA_AVAILABLE=False, GPU_AVAILABLE=True),
        "src.pricing_models.black_scholes": _module("src.pricing_models.black_scholes", black_scholes=_black_scholes),
        "src.pricing_models.exotic_options": _module("src.pricing_models.exotic_options", AsianOption=_exotic.AsianOption,
                                                     BarrierOption=_exotic.BarrierOption, LookbackOption=_exotic.LookbackOption,
                                                     AmericanOption=_exotic.AmericanOption, AutocallableOption=_exotic.AutocallableOption,
                                                     CliquetOption=_exotic.CliquetOption, price_asian=_exotic.price_asian,
                                                     price_barrier=_exotic.price_barrier, price_american=_exotic.price_american),
        "src.pricing_models.heston": _module("src.pricing_models.heston", HestonPricer=_heston.HestonPricer),
        "src.pricing_models.jump_diffusion": _module("src.pricing_models.jump_diffusion", MertonJumpDiffusion=_jump.MertonJumpDiffusion,
                                                     KouJumpDiffusion=_jump.KouJumpDiffusion),
        "src.simulation": sim,
        "src.greeks.unified_greeks": _module("src.greeks.unified_greeks", compute_greeks_unified=_greeks.compute_greeks_unified,
                                             PricerProtocol=_greeks.PricerProtocol, ExoticAdapter=_greeks.ExoticAdapter,
                                             HestonAdapter=_heston.HestonAdapter, JumpDiffusionAdapter=_jump.JumpDiffusionAdapter,
                                             greeks_heston=_heston.greeks_heston),
        "src.exceptions.montecarlo_exceptions": _module("src.exceptions.montecarlo_exceptions", MonteCarloError=_exc.MonteCarloError,
                                                        InputValidationError=_exc.InputValidationError, ConvergenceError=_exc.ConvergenceError,
                                                        AccelerationError=_exc.AccelerationError),
        "src.exceptions.greek_exceptions": _module("src.exceptions.greek_exceptions", GreeksError=_exc.GreeksError),
    }


def install() -> None:
    """Idempotent.  Parents (`src`, `src.pricing_models`, ...) are created as empty packages only if the
    real ones cannot be imported."""
    if _installed:
        return
    for name, mod in _targets().items():
        parts = name.split(".")
        for depth in range(1, len(parts)):
            parent = ".".join(parts[:depth])
            if parent not in sys.modules:
                try:
                    importlib.import_module(parent)
                except Exception:       # the reference's package __init__ pulls optional deps (numba, streamlit): treat as absent
                    pkg = types.ModuleType(parent)
                    pkg.__path__ = []
                    pkg.__optionslab_amd__ = True
                    sys.modules[parent] = pkg
                    _installed[parent] = None
        _installed[name] = sys.modules.get(name)
        sys.modules[name] = mod
        setattr(sys.modules[".".join(parts[:-1])], parts[-1], mod)
    # `from src.pricing_models import MonteCarloPricer`, `from src.greeks import compute_greeks_unified, HestonAdapter` ...
    for package, sources in _PACKAGE_EXPORTS.items():
        pkg = sys.modules[package]
        for source in sources:
            for attr, value in vars(sys.modules[source]).items():
                if attr.startswith("__"):
                    continue
                _package_attrs.append((pkg, attr, getattr(pkg, attr, _MISSING)))
                setattr(pkg, attr, value)


def uninstall() -> None:
    for pkg, attr, previous in reversed(_package_attrs):
        if previous is _MISSING:
            if hasattr(pkg, attr):
                delattr(pkg, attr)
        else:
            setattr(pkg, attr, previous)
    _package_attrs.clear()
    for name, previous in reversed(list(_installed.items())):
        if previous is None:
            sys.modules.pop(name, None)
        else:
            sys.modules[name] = previous
    _installed.clear()
