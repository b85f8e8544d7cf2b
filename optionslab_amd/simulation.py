"""HIP simulation backend with the reference's backend contract
``simulate_*(S, T, r, sigma, q, n_paths, n_steps, seed) -> ndarray[float64]``
(src/simulation/__init__.py:5-6): terminal prices, antithetic => length
2*n_paths ordered [pos | neg] (src/simulation/gbm_numpy.py:51)."""
import numpy as np

from . import _hip

__all__ = ["simulate_gbm_hip", "simulate_gbm_hip_fast", "simulate_gbm_paths_hip", "simulate_gbm_qmc_hip",
           "simulate_gbm_qmc_antithetic_hip", "hip_available"]

hip_available = _hip.hip_available


def simulate_gbm_hip(S: float, T: float, r: float, sigma: float, q: float, n_paths: int, n_steps: int, seed: int,
                     antithetic: bool = True) -> np.ndarray:
    """Multi-step terminal prices (counterpart of simulate_gbm_numpy, gbm_numpy.py:15-53)."""
    if n_paths < 1:
        raise ValueError("n_paths must be >= 1")
    if n_steps < 1:
        raise ValueError("n_steps must be >= 1")
    return _hip.european_terminal(S, T, r, sigma, q, n_paths, n_steps, seed, antithetic)


def simulate_gbm_hip_fast(S: float, T: float, r: float, sigma: float, q: float, n_paths: int, seed: int) -> np.ndarray:
    """Single-step closed form (counterpart of simulate_gbm_numpy_fast, gbm_numpy.py:56-83)."""
    return simulate_gbm_hip(S, T, r, sigma, q, n_paths, 1, seed, True)


def simulate_gbm_paths_hip(S: float, T: float, r: float, sigma: float, q: float, n_paths: int, n_steps: int, seed: int) -> np.ndarray:
    """Full paths, shape (n_paths, n_steps + 1), column 0 = S (counterpart of simulate_gbm_paths,
    gbm_numpy.py:86-118).  The kernel writes the reference's C-order (n_paths, n_steps + 1) layout
    itself, so the array arrives in one device-to-host copy with no host-side transpose."""
    if n_paths < 1 or n_steps < 1:
        raise ValueError("n_paths and n_steps must be >= 1")
    return _hip.gbm_paths(S, T, r, sigma, q, n_paths, n_steps, seed, path_major=True)


def simulate_gbm_qmc_hip(S: float, T: float, r: float, sigma: float, q: float, n_paths: int, n_steps: int, seed: int) -> np.ndarray:
    """Scrambled-Sobol terminal prices, length n_paths (counterpart of simulate_gbm_qmc, gbm_qmc.py:14-46)."""
    from .monte_carlo import sobol_tables

    sv, shift = sobol_tables(n_steps, seed, n_paths)
    return _hip.european_qmc_terminal(S, T, r, sigma, q, n_paths, sv, shift)


def simulate_gbm_qmc_antithetic_hip(S: float, T: float, r: float, sigma: float, q: float, n_paths: int, n_steps: int,
                                    seed: int) -> np.ndarray:
    """Sobol points plus their mirrored normals, length 2 * n_paths, [pos | neg]
    (counterpart of simulate_gbm_qmc_antithetic, gbm_qmc.py:49-76)."""
    from .monte_carlo import sobol_tables

    sv, shift = sobol_tables(n_steps, seed, n_paths)
    return _hip.european_qmc_terminal(S, T, r, sigma, q, n_paths, sv, shift, antithetic=True)
