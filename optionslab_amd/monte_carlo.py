"""MonteCarloPricer on the MI355X path engine.

Same constructor, attributes, method signatures, return types and error
behaviour as the reference class (src/pricing_models/monte_carlo.py:28-186), so
it drops into the Streamlit page (streamlit_app/pages/1_MonteCarlo_Basic.py:139-160),
``compute_greeks_unified`` and tests/test_monte_carlo.py unchanged.  Every
``MCMethod`` executes on the GPU through libolmc; nothing here computes paths
on the CPU.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from enum import Enum
from typing import Literal, Optional, Union

import numpy as np

from . import _hip
from .exceptions import AccelerationError

SOBOL_MAX_DIM = 21201            # src/simulation/gbm_qmc.py:29-30
SOBOL_VALIDATED_SCIPY = "1.15.3"  # the SciPy whose private Sobol tables (_sv, _shift) this layout was validated against
                                  # (the golden QMC vectors of tests/golden were captured with it); other versions are
                                  # accepted only if they pass _check_sobol_tables
_sobol_cache: "OrderedDict[tuple, tuple]" = OrderedDict()


class SobolDirections(np.ndarray):
    """The (dims, 30) uint32 direction matrix with the number of its columns that are KNOWN: tables derived from the engine's
    public behaviour hold only the columns the requested points can select (the others are zero), so a point index at or beyond
    2**valid_bits would silently repeat points.  `_hip.european_qmc*` refuses such a range (ValueError)."""

    valid_bits: int = 30

    def __new__(cls, sv: np.ndarray, valid_bits: int = 30):
        obj = np.ascontiguousarray(sv, dtype=np.uint32).view(cls)
        obj.valid_bits = int(valid_bits)
        return obj

    def __array_finalize__(self, obj):
        self.valid_bits = getattr(obj, "valid_bits", 30)


def sobol_tables(n_steps: int, seed: int, n_points: int = 1 << 20):
    """(sv, shift) of scipy.stats.qmc.Sobol(d=min(n_steps, 21201), scramble=True, seed=seed):
    the scrambled direction matrix and digital shift the device expands into the very same
    points SciPy's .random(n) returns (src/simulation/gbm_qmc.py:32-33).  Only this small
    host-side table construction uses SciPy (a dependency of the reference); the points, the
    inverse normal and the payoff reduction run on the GPU.  Cached: FD Greeks reuse it.

    The tables are SciPy privates (`_sv`, `_shift`).  Where they are absent or no longer reproduce
    the engine's own points (_check_sobol_tables), they are DERIVED from the engine's public
    behaviour alone (_derive_sobol_tables: .random() and .fast_forward()), for as many columns as
    the first `n_points` points (point offset included) can select; only an engine that offers
    neither way disables MCMethod.QMC."""
    d = min(int(n_steps), SOBOL_MAX_DIM)
    key = (d, int(seed))
    need_bits = max(int(n_points) - 1, 1).bit_length()
    hit = _sobol_cache.get(key)
    if hit is not None and hit[2] >= need_bits:
        _sobol_cache.move_to_end(key)
        return hit[0], hit[1]
    from scipy.stats import qmc

    eng = qmc.Sobol(d=d, scramble=True, seed=seed)
    if getattr(eng, "bits", None) != 30:
        raise AccelerationError("this SciPy's Sobol engine is not the 30-bit engine the device kernels expand "
                                f"(validated with SciPy {SOBOL_VALIDATED_SCIPY})", backend="hip")
    val = None
    if hasattr(eng, "_sv") and hasattr(eng, "_shift"):
        try:
            cand = (np.ascontiguousarray(eng._sv, dtype=np.uint32), np.ascontiguousarray(eng._shift, dtype=np.uint32))
            _check_sobol_tables(cand, eng, d)
            val = (SobolDirections(cand[0], 30), cand[1], 30)
        except AccelerationError as e:
            private_failure = e
    else:
        private_failure = AccelerationError("this SciPy's Sobol engine does not expose 30-bit _sv/_shift tables "
                                            f"(validated with SciPy {SOBOL_VALIDATED_SCIPY})", backend="hip")
    if val is None:
        try:
            sv, shift = _derive_sobol_tables(qmc.Sobol, d, seed, need_bits)
        except (AttributeError, TypeError, ValueError) as e:
            raise AccelerationError(f"{private_failure}; and the tables cannot be derived from the engine's public behaviour either "
                                    f"({type(e).__name__}: {e}); MCMethod.QMC is unavailable", backend="hip") from e
        val = (SobolDirections(sv, need_bits), shift, need_bits)
    _sobol_cache[key] = val
    while len(_sobol_cache) > 8:
        _sobol_cache.popitem(last=False)
    return val[0], val[1]


def _derive_sobol_tables(engine_class, d: int, seed: int, bits: int):
    """(sv, shift) from the PUBLIC behaviour of the engine: point 0 is the digital shift, and consecutive points of a Gray-code Sobol
    sequence differ by exactly one column of the direction matrix -- x_i = x_{i-1} ^ sv[:, ctz(i)] -- so points 2^c - 1 and 2^c give
    column c.  .fast_forward() skips the points in between (O(2^bits * d) inside SciPy: 0.3 s for 2^20 points x 252 dimensions; the
    columns beyond `bits` stay zero and are never selected by a point index below 2^bits: the caller records `bits` with the table,
    SobolDirections.valid_bits).  The result must reproduce the engine's own points at the head of the sequence AND across the last
    derived column, or it is refused (ValueError -- explicit raises, not `assert`: `python -O` must not skip the checks)."""
    import warnings

    bits = min(max(int(bits), 4), 30)
    eng = engine_class(d=d, scramble=True, seed=seed)

    def draw(engine, n):                  # SciPy warns about every n that is not a power of two; these draws are bookkeeping, not a sample
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", UserWarning)
            return engine.random(n)

    def as_int(u):
        u = np.asarray(u, dtype=np.float64)
        x = np.rint(u * 2.0 ** 30)
        if not (np.array_equal(x * 2.0 ** -30, u) and x.min() >= 0 and x.max() < 2.0 ** 30):
            raise ValueError("the engine's points are not multiples of 2^-30")
        return x.astype(np.uint32)

    shift = as_int(draw(eng, 1)[0])
    sv = np.zeros((d, 30), dtype=np.uint32)
    last_idx, last_x = 0, shift
    for c in range(bits):
        t = 1 << c
        if last_idx < t - 1:
            skip = t - 1 - (last_idx + 1)
            if skip:
                eng.fast_forward(skip)
            two = draw(eng, 2)
            before, at = as_int(two[0]), as_int(two[1])
        else:
            before, at = last_x, as_int(draw(eng, 1)[0])
        sv[:, c] = before ^ at
        last_idx, last_x = t, at
    check = engine_class(d=d, scramble=True, seed=seed)
    if not np.array_equal(expand_sobol_points(sv, shift, 0, 8), np.asarray(draw(check, 8), dtype=np.float64)):
        raise ValueError("derived tables do not reproduce the head of the engine's sequence")
    far = (1 << bits) - 5
    check.fast_forward(far - 8)
    if not np.array_equal(expand_sobol_points(sv, shift, far, 4), np.asarray(draw(check, 4), dtype=np.float64)):
        raise ValueError("derived tables do not reproduce the end of the derived range")
    return sv, shift


def expand_sobol_points(sv: np.ndarray, shift: np.ndarray, first: int, count: int) -> np.ndarray:
    """Host restatement of what the device does with the tables (olmc.h, olmc_european_qmc): point k =
    shift ^ XOR over the set bits b of gray(k) = k ^ (k >> 1) of sv[:, b]; u = x * 2^-30.  Used only to CHECK the
    tables against SciPy's own points on a few points, never to price."""
    out = np.empty((count, sv.shape[0]), dtype=np.float64)
    for i in range(count):
        k = first + i
        g, x, b = k ^ (k >> 1), shift.copy(), 0
        while g:
            if g & 1:
                x ^= sv[:, b]
            g >>= 1
            b += 1
        out[i] = x.astype(np.float64) * 2.0 ** -30
    return out


def _check_sobol_tables(val, eng, d: int) -> None:
    """`_sv` / `_shift` are SciPy privates: a SciPy release may keep the names and change what they mean, which no
    hasattr() sees.  So the tables are checked by BEHAVIOUR on every (uncached) construction: the first 8 points
    expanded from them must equal, bit for bit, the points the very same engine returns from .random(8) (a fresh
    engine has drawn nothing, so this consumes only `eng`, which is discarded).  Any mismatch disables MCMethod.QMC
    loudly instead of pricing on a different point set than the reference's."""
    sv, shift = val
    if sv.shape != (d, 30) or shift.shape != (d,):
        raise AccelerationError(f"SciPy Sobol tables have shape {sv.shape} / {shift.shape}, expected ({d}, 30) / ({d},) "
                                f"(validated with SciPy {SOBOL_VALIDATED_SCIPY})", backend="hip")
    want = np.asarray(eng.random(8), dtype=np.float64)
    if not np.array_equal(expand_sobol_points(sv, shift, 0, 8), want):
        import scipy
        raise AccelerationError(f"SciPy {scipy.__version__}: the Sobol engine's _sv/_shift tables no longer reproduce its own points "
                                f"(layout validated with SciPy {SOBOL_VALIDATED_SCIPY}); MCMethod.QMC is unavailable", backend="hip")


NUMBA_AVAILABLE = False          # kept for `from ...monte_carlo import NUMBA_AVAILABLE` (monte_carlo.py:189)
GREEK_KEYS = ("price", "delta", "gamma", "vega", "theta", "rho", "vanna", "charm", "vomma")
_GREEK_KEYS6 = GREEK_KEYS[:6]


class MCMethod(Enum):
    """Backend selector (monte_carlo.py:28-34) plus the new member HIP.  NUMPY,
    NUMBA and HIP all mean "multi-step on the device"; FAST forces a single step
    (monte_carlo.py:87); QMC is the device scrambled-Sobol path (no antithetic)."""

    NUMPY = "numpy"
    NUMBA = "numba"
    QMC = "qmc"
    FAST = "fast"
    HIP = "hip"


@dataclass
class MCResult:
    """monte_carlo.py:37-43"""

    price: float
    std_error: float = 0.0
    n_paths: int = 0


class MonteCarloPricer:
    """monte_carlo.py:46-186 on the device.  Additive: `n_gpus` (keyword only, default 1) -- with n_gpus > 1 the pseudo-random
    price(), greeks() and price_with_control_variate() shard `num_simulations` over the first n_gpus devices of THIS process
    (contiguous global path ranges, one launch per device, ONE RCCL all-reduce of the sums over xGMI: olmc_multi_gpu_*, no
    torch); the paths, hence the results up to the association of the sums, do not depend on n_gpus.  MCMethod.QMC shards its
    price(), greeks() and price_with_control_variate() the same way (contiguous blocks of Sobol POINTS, cut on multiples of 512); its
    terminal array stays on one device."""

    __slots__ = ("num_simulations", "num_steps", "seed", "method", "_use_numba", "n_gpus")

    def __init__(self, num_simulations: int = 100000, num_steps: int = 1, seed: Optional[int] = None,
                 method: MCMethod = MCMethod.NUMPY, *, n_gpus: int = 1):
        if num_simulations < 1:
            raise ValueError("num_simulations must be >= 1")
        if n_gpus < 1:
            raise ValueError("n_gpus must be >= 1")
        self.n_gpus = int(n_gpus)
        self.num_simulations = num_simulations
        self.num_steps = num_steps
        # one 31-bit draw at construction, then fixed: repeated price() calls and
        # all FD bumps share the normals (monte_carlo.py:68-70)
        self.seed = seed if seed is not None else int(np.random.default_rng().integers(0, 2**31))
        self.method = method
        self._use_numba = False

    # ------------------------------------------------------------------ helpers
    def _steps(self) -> int:
        if self.method == MCMethod.FAST:
            return 1
        if self.num_steps < 1:
            raise ValueError("num_steps must be >= 1")
        return int(self.num_steps)

    def _simulate(self, S: float, T: float, r: float, sigma: float, q: float, seed: Optional[int] = None) -> np.ndarray:
        """Terminal prices, length 2*num_simulations, [pos | neg] (monte_carlo.py:74-106)."""
        actual_seed = seed if seed is not None else self.seed
        if self.method == MCMethod.QMC:
            sv, shift = sobol_tables(self._steps(), actual_seed, self.num_simulations)
            return _hip.european_qmc_terminal(S, T, r, sigma, q, self.num_simulations, sv, shift)
        return _hip.european_terminal(S, T, r, sigma, q, self.num_simulations, self._steps(), actual_seed, True)

    # -------------------------------------------------------------------- price
    def price(self, S: float, K: float, T: float, r: float, sigma: float, option_type: Literal["call", "put"],
              q: float = 0.0, seed: Optional[int] = None, return_error: bool = False) -> Union[float, MCResult]:
        if T <= 0:  # monte_carlo.py:133-135
            intrinsic = max(S - K, 0) if option_type == "call" else max(K - S, 0)
            return MCResult(intrinsic, 0.0, 0) if return_error else intrinsic
        actual_seed = seed if seed is not None else self.seed
        if self.method == MCMethod.QMC:
            sv, shift = sobol_tables(self._steps(), actual_seed, self.num_simulations)
            if self.n_gpus > 1:
                st = _hip.multi_gpu_european_qmc(S, K, T, r, sigma, q, option_type == "call", self.num_simulations, sv, shift, self.n_gpus)
            else:
                st = _hip.european_qmc(S, K, T, r, sigma, q, option_type == "call", self.num_simulations, sv, shift)
        elif self.n_gpus > 1:
            st = _hip.multi_gpu_european(S, K, T, r, sigma, q, option_type == "call", self.num_simulations, self._steps(), actual_seed, True,
                                         self.n_gpus)
        else:
            st = _hip.european(S, K, T, r, sigma, q, option_type == "call", self.num_simulations, self._steps(),
                               actual_seed, True)
        if return_error:
            return MCResult(float(st.price), float(st.std_error), int(st.n))
        return float(st.price)

    def price_with_control_variate(self, S: float, K: float, T: float, r: float, sigma: float,
                                   option_type: Literal["call", "put"], q: float = 0.0,
                                   seed: Optional[int] = None) -> float:
        """Terminal spot as control, E[S_T] = S e^{(r-q)T} (monte_carlo.py:154-186)."""
        actual_seed = seed if seed is not None else self.seed
        if self.method == MCMethod.QMC:   # the same five-moment reduction on the Sobol points (N samples, no mirror)
            sv, shift = sobol_tables(self.num_steps, actual_seed, self.num_simulations)
            if self.n_gpus > 1:
                return float(_hip.multi_gpu_european_qmc_cv(S, K, T, r, sigma, q, option_type == "call", self.num_simulations, sv, shift, self.n_gpus).value)
            return float(_hip.european_qmc_cv(S, K, T, r, sigma, q, option_type == "call", self.num_simulations, sv, shift).value)
        if self.n_gpus > 1:
            m = _hip.multi_gpu_european_cv(S, K, T, r, sigma, q, option_type == "call", self.num_simulations, self._steps(), actual_seed, True,
                                           self.n_gpus)
        else:
            m = _hip.european_cv(S, K, T, r, sigma, q, option_type == "call", self.num_simulations, self._steps(),
                                 actual_seed, True)
        return float(m.value)

    # ------------------------------------------------------------------- greeks
    def greeks(self, S: float, K: float, T: float, r: float, sigma: float,
               option_type: Literal["call", "put"] = "call", q: float = 0.0, seed: Optional[int] = None,
               include_second_order: bool = True) -> "OrderedDict[str, float]":
        """Additive convenience: the OrderedDict compute_greeks_unified(self, ...)
        returns (src/greeks/unified_greeks.py:235-367), from ONE fused launch that
        draws the normals once and evaluates all 8 / 14 bumped contracts per path."""
        if T > 0:                   # what compute_greeks_unified(self, ...) does for this pricer, minus its dispatch (T <= 0 takes its branch there)
            try:
                return self._fused_greeks(S, K, T, r, sigma, option_type, q, include_second_order, seed)
            except Exception as e:  # unified_greeks.py:366-367
                from .exceptions import GreeksError
                raise GreeksError(f"Failed to compute unified Greeks: {str(e)}") from e
        from .greeks import compute_greeks_unified

        kw = {} if seed is None else {"seed": seed}
        return compute_greeks_unified(self, S, K, T, r, sigma, option_type, q, include_second_order, **kw)

    def _fused_greeks(self, S, K, T, r, sigma, option_type, q, include_second_order, seed=None):
        actual_seed = seed if seed is not None else self.seed
        if self.method == MCMethod.QMC:   # the bumped contracts share the Sobol points (same dims, same seed): one launch prices them all
            sv, shift = sobol_tables(self._steps(), actual_seed, self.num_simulations)
            if self.n_gpus > 1:
                vals, _ = _hip.multi_gpu_european_qmc_greeks_fd(S, K, T, r, sigma, q, option_type == "call", self.num_simulations, sv, shift,
                                                                include_second_order, self.n_gpus, want_evals=False)
            else:
                vals, _ = _hip.european_qmc_greeks_fd(S, K, T, r, sigma, q, option_type == "call", self.num_simulations, sv, shift,
                                                      include_second_order, want_evals=False)
        elif self.n_gpus > 1:
            vals, _ = _hip.multi_gpu_greeks_fd(S, K, T, r, sigma, q, option_type == "call", self.num_simulations, self._steps(), actual_seed,
                                               include_second_order, self.n_gpus, want_evals=False)
        else:
            vals, _ = _hip.european_greeks_fd(S, K, T, r, sigma, q, option_type == "call", self.num_simulations,
                                              self._steps(), actual_seed, include_second_order, want_evals=False)
        return OrderedDict(zip(GREEK_KEYS if include_second_order else _GREEK_KEYS6, vals))      # zip stops at the shorter: 6 or 9 floats


__all__ = ["MonteCarloPricer", "MCMethod", "MCResult", "NUMBA_AVAILABLE"]
