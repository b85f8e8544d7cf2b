"""ctypes binding of libolmc.so (C ABI: include/olmc.h).

The library is loaded on first use and there is no CPU fallback: if the shared
object is missing, or no HIP device answers, every call raises
``AccelerationError(..., backend="hip")`` (the reference defines that exception
for exactly this, src/exceptions/montecarlo_exceptions.py:105-131).
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import sys
import threading
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .exceptions import AccelerationError

# $OLMC_LIBRARY overrides the in-tree build (A/B measurements of alternative builds)
LIBRARY_PATH = os.environ.get("OLMC_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libolmc.so")
MAX_BATCH = 16
AVG_ARITHMETIC, AVG_GEOMETRIC, AVG_ARITHMETIC_FAST = 0, 1, 2
_U64 = (1 << 64) - 1


class Stats(C.Structure):
    _fields_ = [("sum", C.c_double), ("sumsq", C.c_double), ("n", C.c_int64),
                ("price", C.c_double), ("std_error", C.c_double)]


class Option(C.Structure):
    _fields_ = [("S", C.c_double), ("K", C.c_double), ("T", C.c_double), ("r", C.c_double),
                ("sigma", C.c_double), ("q", C.c_double), ("is_call", C.c_int32), ("reserved", C.c_int32)]


class CvMoments(C.Structure):
    _fields_ = [("sum_d", C.c_double), ("sum_s", C.c_double), ("sum_dd", C.c_double), ("sum_ss", C.c_double),
                ("sum_ds", C.c_double), ("n", C.c_int64), ("value", C.c_double)]


class DevInfo(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("arch", C.c_char * 32), ("compute_units", C.c_int32),
                ("clock_mhz", C.c_int32), ("wavefront", C.c_int32), ("device", C.c_int32), ("hbm_bytes", C.c_int64)]


_D, _I, _I32, _I64, _U64T, _P = C.c_double, C.c_int, C.c_int32, C.c_int64, C.c_uint64, C.c_void_p
_byref = C.byref
_Out9 = C.c_double * 9
_SIX = [_D] * 6

# name -> (restype, argtypes); must list every symbol include/olmc.h declares
PROTOTYPES = {
    "olmc_abi_version": (_I, []),
    "olmc_init": (_I, [_I]),
    "olmc_shutdown": (_I, []),
    "olmc_last_error": (C.c_char_p, []),
    "olmc_device_info": (_I, [C.POINTER(DevInfo)]),
    "olmc_european": (_I, _SIX + [_I, _I64, _I32, _U64T, _I, C.POINTER(Stats)]),
    "olmc_european_shard": (_I, _SIX + [_I, _I64, _I64, _I32, _U64T, _I, C.POINTER(Stats)]),
    "olmc_european_shard_dev": (_I, _SIX + [_I, _I64, _I64, _I32, _U64T, _I, _P, _P]),
    "olmc_fetch_dev": (_I, [_P, _I32, _P, C.POINTER(_D)]),
    "olmc_european_batch": (_I, [C.POINTER(Option), _I32, _I64, _I64, _I32, _U64T, _I, C.POINTER(Stats)]),
    "olmc_european_multi": (_I, [C.POINTER(Option), C.POINTER(C.c_uint32), _I64, _I64, _I32, _U64T, _I, C.POINTER(Stats)]),
    "olmc_multi_capacity": (_I, [C.POINTER(_I64)]),
    "olmc_contract_layout": (_I, [C.POINTER(Option), _I32, _I32, C.POINTER(_I32), C.POINTER(_I32), C.POINTER(C.c_uint32), C.POINTER(_I32), C.POINTER(_D)]),
    "olmc_european_greeks_fd": (_I, _SIX + [_I, _I64, _I32, _U64T, _I, C.POINTER(_D), C.POINTER(Stats)]),
    "olmc_european_terminal": (_I, [_D] * 5 + [_I64, _I32, _U64T, _I, C.POINTER(_D)]),
    "olmc_gbm_paths": (_I, [_D] * 5 + [_I64, _I32, _U64T, _I, C.POINTER(_D)]),
    "olmc_european_cv": (_I, _SIX + [_I, _I64, _I32, _U64T, _I, C.POINTER(CvMoments)]),
    "olmc_european_cv_shard": (_I, _SIX + [_I, _I64, _I64, _I32, _U64T, _I, C.POINTER(CvMoments)]),
    "olmc_combine_cv": (_I, [C.POINTER(CvMoments), _I32, _D, _D, _D, _D, C.POINTER(CvMoments)]),
    "olmc_european_qmc": (_I, _SIX + [_I, _I64, _I64, _I32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), _I32, C.POINTER(Stats)]),
    "olmc_european_qmc_cv": (_I, _SIX + [_I, _I64, _I64, _I32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), _I32, C.POINTER(CvMoments)]),
    "olmc_european_qmc_batch": (_I, [C.POINTER(Option), _I32, _I64, _I64, _I32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), _I32, C.POINTER(Stats)]),
    "olmc_european_qmc_greeks_fd": (_I, _SIX + [_I, _I64, _I32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), _I32, _I, C.POINTER(_D), C.POINTER(Stats)]),
    "olmc_european_qmc_terminal": (_I, [_D] * 5 + [_I64, _I64, _I32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), _I32, _I, C.POINTER(_D)]),
    "olmc_asian": (_I, _SIX + [_I, _I, _I64, _I64, _I32, _U64T, _I, C.POINTER(Stats)]),
    "olmc_asian_greeks_fd": (_I, _SIX + [_I, _I, _I64, _I32, _U64T, _I, _I, C.POINTER(_D), C.POINTER(Stats)]),
    "olmc_extrema_greeks_fd": (_I, _SIX + [_I, _I, _D, _I64, _I32, _U64T, _I, _I, C.POINTER(_D), C.POINTER(Stats)]),
    "olmc_barrier": (_I, _SIX + [_I, _D, _I, _I64, _I64, _I32, _U64T, _I, C.POINTER(Stats)]),
    "olmc_lookback": (_I, _SIX + [_I, _I, _I64, _I64, _I32, _U64T, _I, C.POINTER(Stats)]),
    "olmc_autocallable": (_I, [_D] * 9 + [_I32, _I64, _I64, _I32, _U64T, _I, C.POINTER(Stats)]),
    "olmc_cliquet": (_I, [_D] * 9 + [_I32, _I64, _I64, _I32, _U64T, _I, C.POINTER(Stats)]),
    "olmc_exercise_boundary": (_I, _SIX + [_I, _I64, _I32, _U64T, C.POINTER(_D)]),
    "olmc_heston_paths": (_I, [_D] * 9 + [_I64, _I32, _U64T, _I, C.POINTER(_D), C.POINTER(_D)]),
    "olmc_jump_paths": (_I, [_D] * 5 + [_I, _D, _D, _D, _D, _I64, _I32, _U64T, _I, C.POINTER(_D)]),
    "olmc_american_lsm": (_I, _SIX + [_I, _I64, _I32, _I32, _U64T, C.POINTER(Stats)]),
    "olmc_jump_diffusion": (_I, _SIX + [_I, _I, _D, _D, _D, _D, _I64, _I64, _I32, _U64T, C.POINTER(Stats)]),
    "olmc_heston": (_I, [_D] * 5 + [_I] + [_D] * 5 + [_I64, _I64, _I32, _U64T, _I, C.POINTER(Stats)]),
    "olmc_multi_gpu_european": (_I, _SIX + [_I, _I64, _I32, _U64T, _I, _I, C.POINTER(Stats)]),
    "olmc_multi_gpu_greeks_fd": (_I, _SIX + [_I, _I64, _I32, _U64T, _I, _I, C.POINTER(_D), C.POINTER(Stats)]),
    "olmc_multi_gpu_european_cv": (_I, _SIX + [_I, _I64, _I32, _U64T, _I, _I, C.POINTER(CvMoments)]),
    "olmc_multi_gpu_european_qmc": (_I, _SIX + [_I, _I64, _I32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), _I32, _I, C.POINTER(Stats)]),
    "olmc_multi_gpu_european_qmc_greeks_fd": (_I, _SIX + [_I, _I64, _I32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), _I32, _I, _I, C.POINTER(_D), C.POINTER(Stats)]),
    "olmc_multi_gpu_european_qmc_cv": (_I, _SIX + [_I, _I64, _I32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), _I32, _I, C.POINTER(CvMoments)]),
    "olmc_multi_gpu_spans": (_I, [C.POINTER(_D)]),
    "olmc_combine_stats": (_I, [C.POINTER(Stats), _I32, _D, _D, C.POINTER(Stats)]),
    "olmc_philox_words": (_I, [_U64T, _I64, _I64, _I32, _I32, C.c_uint32, C.POINTER(C.c_uint32)]),
    "olmc_normals": (_I, [_U64T, _I64, _I64, _I32, C.POINTER(C.c_float)]),
    "olmc_profile_enable": (_I, [_I]),
    "olmc_tune": (_I, [_I, _I]),
    "olmc_profile_reset": (_I, []),
    "olmc_kernel_time": (_I, [C.POINTER(_I64), C.POINTER(_D)]),
}

_lock = threading.Lock()
_lib: Optional[C.CDLL] = None
_initialised = False


def _share_hip_runtime_with_torch() -> None:
    """PyTorch-ROCm wheels bundle their own HIP runtime (torch/lib/libamdhip64.so, unversioned soname).  If
    libolmc initialises the system runtime first, a later `import torch` in the same process brings up a
    SECOND runtime and reports "no GPUs found"; with torch's copy loaded globally first, libolmc's HIP symbols
    bind to it (global scope wins) and the process has one runtime whichever is imported first -- needed
    because torch streams / device buffers are handed to olmc_european_shard_dev.  torch itself is NOT
    imported here.  OLMC_SYSTEM_HIP=1 opts out."""
    if "torch" in sys.modules or os.environ.get("OLMC_SYSTEM_HIP") == "1":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load_library() -> C.CDLL:
    """dlopen libolmc.so and bind every prototype (no device is touched)."""
    global _lib
    with _lock:
        if _lib is None:
            _share_hip_runtime_with_torch()
            if not os.path.exists(LIBRARY_PATH):
                raise AccelerationError(
                    f"{LIBRARY_PATH} is not built (run `python -m optionslab_amd.build`); there is no CPU fallback",
                    backend="hip")
            try:
                lib = C.CDLL(LIBRARY_PATH)
            except OSError as e:
                raise AccelerationError(f"cannot load {LIBRARY_PATH}: {e}", backend="hip") from e
            for name, (res, args) in PROTOTYPES.items():
                try:
                    fn = getattr(lib, name)
                except AttributeError:
                    if os.environ.get("OLMC_LIBRARY"):      # an older build under A/B measurement: newer entry points are simply absent
                        continue
                    raise AccelerationError(f"{LIBRARY_PATH} lacks {name}: stale build (run `python -m optionslab_amd.build`)", backend="hip")
                fn.restype, fn.argtypes = res, args
            _lib = lib
        return _lib


def _check(rc: int) -> None:
    if rc != 0:
        msg = load_library().olmc_last_error().decode("utf-8", "replace")
        raise AccelerationError(f"libolmc error {rc}: {msg}", backend="hip")


def lib() -> C.CDLL:
    """Loaded library with the device initialised (device = $OLMC_DEVICE, else $LOCAL_RANK, else 0)."""
    global _initialised
    l = load_library()
    if not _initialised:
        dev = int(os.environ.get("OLMC_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        _check(l.olmc_init(dev))
        _initialised = True
    return l


def hip_available() -> bool:
    try:
        lib()
        return True
    except AccelerationError:
        return False


def shutdown() -> None:
    global _initialised
    if _lib is not None:
        _lib.olmc_shutdown()
    _initialised = False


def seed64(seed: int) -> int:
    return int(seed) & _U64


def device_info() -> dict:
    info = DevInfo()
    _check(lib().olmc_device_info(C.byref(info)))
    arch = info.arch.decode()
    return dict(name=info.name.decode() or f"AMD Instinct ({arch.split(':')[0]})", arch=arch, compute_units=info.compute_units,
                clock_mhz=info.clock_mhz, wavefront=info.wavefront, device=info.device, hbm_bytes=info.hbm_bytes)


def european(S, K, T, r, sigma, q, is_call: bool, n_paths: int, n_steps: int, seed: int, antithetic: bool = True,
             path_offset: int = 0) -> Stats:
    out = Stats()
    rc = lib().olmc_european_shard(S, K, T, r, sigma, q, bool(is_call), int(path_offset), int(n_paths), int(n_steps), int(seed) & _U64,
                                   bool(antithetic), _byref(out))
    if rc:                              # the hot blocking call of price(): no helper frames on the success path
        _check(rc)
    return out


def european_shard_dev(S, K, T, r, sigma, q, is_call: bool, path_offset: int, n_local: int, n_steps: int, seed: int,
                       antithetic: bool, d_triple_ptr: int, stream_ptr: int = 0) -> None:
    _check(lib().olmc_european_shard_dev(S, K, T, r, sigma, q, int(is_call), int(path_offset), int(n_local), int(n_steps),
                                         seed64(seed), int(antithetic), C.c_void_p(d_triple_ptr),
                                         C.c_void_p(stream_ptr) if stream_ptr else None))


def fetch_dev(d_src_ptr: int, n: int, stream_ptr: int = 0) -> List[float]:
    """Blocking fetch of n doubles that work queued on the stream leaves at the device pointer (olmc_fetch_dev)."""
    out = (C.c_double * int(n))()
    _check(lib().olmc_fetch_dev(C.c_void_p(d_src_ptr), int(n), C.c_void_p(stream_ptr) if stream_ptr else None, out))
    return list(out)


def european_batch(options: Sequence[Tuple[float, float, float, float, float, float, bool]], n_paths: int, n_steps: int,
                   seed: int, antithetic: bool = True, path_offset: int = 0) -> List[Stats]:
    k = len(options)
    arr = (Option * k)(*[Option(S, K, T, r, v, q, int(c), 0) for (S, K, T, r, v, q, c) in options])
    out = (Stats * k)()
    _check(lib().olmc_european_batch(arr, k, int(path_offset), int(n_paths), int(n_steps), seed64(seed), int(antithetic), out))
    return list(out)


_OPTION_DT = np.dtype([("S", "f8"), ("K", "f8"), ("T", "f8"), ("r", "f8"), ("sigma", "f8"), ("q", "f8"), ("is_call", "i4"), ("reserved", "i4")])
_STATS_DT = np.dtype([("sum", "f8"), ("sumsq", "f8"), ("n", "i8"), ("price", "f8"), ("std_error", "f8")])
assert _OPTION_DT.itemsize == C.sizeof(Option) == 56 and _STATS_DT.itemsize == C.sizeof(Stats) == 40
_P_OPTION, _P_STATS, _P_U32 = C.POINTER(Option), C.POINTER(Stats), C.POINTER(C.c_uint32)


def european_multi(S, K, T, r, sigma, q, is_call, n_paths: int, n_steps: int, seed: int, antithetic: bool = True,
                   tags=None) -> np.ndarray:
    """Arrays of contracts -> structured result array with fields of olmc_stats (one launch).  The marshalling is on the
    latency path of small batches (it was 33 us of a 69 us one-contract call), hence the prebuilt dtypes and the direct
    field assignments (NumPy broadcasts scalars and arrays alike)."""
    S = np.asarray(S, dtype=np.float64)
    n = S.shape[0]
    opts = np.empty((n, 7))                      # one olmc_option per row: six doubles, then {is_call, reserved} as two int32
    opts[:, 0], opts[:, 1], opts[:, 2], opts[:, 3], opts[:, 4], opts[:, 5] = S, K, T, r, sigma, q
    flags = opts.view(np.int32)
    flags[:, 12] = is_call
    flags[:, 13] = 0
    out = np.empty((n, 5))                       # one olmc_stats per row (the third column is the int64 sample count)
    ptags = None
    if tags is not None:
        tags = np.ascontiguousarray(tags, dtype=np.uint32)
        ptags = C.cast(tags.ctypes.data, _P_U32)
    _check(lib().olmc_european_multi(C.cast(opts.ctypes.data, _P_OPTION), ptags, n, int(n_paths), int(n_steps), int(seed) & _U64,
                                     bool(antithetic), C.cast(out.ctypes.data, _P_STATS)))
    return out.view(_STATS_DT).reshape(n)


def contract_layout(options: Sequence[Tuple[float, float, float, float, float, float, bool]], n_steps: int):
    """(nsets, pos[k], base_mask, upper_continues_slot0, scale[nsets]) of a set of contracts for the fused kernels (olmc_contract_layout;
    pure host function: loads the library, not the GPU)."""
    k = len(options)
    arr = (Option * k)(*[Option(S, K, T, r, v, q, int(c), 0) for (S, K, T, r, v, q, c) in options])
    nsets, pos, mask, cont, scale = C.c_int32(0), (C.c_int32 * k)(), C.c_uint32(0), C.c_int32(0), (C.c_double * 16)()
    _check(load_library().olmc_contract_layout(arr, k, int(n_steps), C.byref(nsets), pos, C.byref(mask), C.byref(cont), scale))
    return nsets.value, list(pos), mask.value, bool(cont.value), list(scale)[:nsets.value]


def multi_capacity() -> Tuple[int, int]:
    """(contracts, workgroups per contract) the batch workspace of european_multi is sized for right now."""
    out = (C.c_int64 * 2)()
    _check(lib().olmc_multi_capacity(out))
    return int(out[0]), int(out[1])


def european_greeks_fd(S, K, T, r, sigma, q, is_call: bool, n_paths: int, n_steps: int, seed: int,
                       second_order: bool, want_evals: bool = True) -> Tuple[List[float], List[Stats]]:
    """(price, delta, gamma, vega, theta, rho, vanna, charm, vomma) and -- unless want_evals is False, which spares a blocking
    Greeks call the marshalling of fourteen structs AND lets the library launch its prices-only kernel -- the statistics of the
    bumped contracts in the reference's call order."""
    out9 = _Out9()
    if want_evals:
        evals = (Stats * 14)()
        _check(lib().olmc_european_greeks_fd(S, K, T, r, sigma, q, int(is_call), int(n_paths), int(n_steps), seed64(seed),
                                             int(second_order), out9, evals))
        return list(out9), list(evals)
    rc = lib().olmc_european_greeks_fd(S, K, T, r, sigma, q, bool(is_call), int(n_paths), int(n_steps), int(seed) & _U64, bool(second_order), out9, None)
    if rc:                              # the hot blocking call of greeks(): no helper frames on the success path
        _check(rc)
    return out9[:], []


def european_terminal(S, T, r, sigma, q, n_paths: int, n_steps: int, seed: int, antithetic: bool = True) -> np.ndarray:
    out = np.empty(int(n_paths) * (2 if antithetic else 1), dtype=np.float64)
    _check(lib().olmc_european_terminal(S, T, r, sigma, q, int(n_paths), int(n_steps), seed64(seed), int(antithetic),
                                        out.ctypes.data_as(C.POINTER(C.c_double))))
    return out


def _path_matrix(n_paths: int, n_steps: int, path_major: bool) -> np.ndarray:
    return np.empty((int(n_paths), int(n_steps) + 1) if path_major else (int(n_steps) + 1, int(n_paths)), dtype=np.float64)


def gbm_paths(S, T, r, sigma, q, n_paths: int, n_steps: int, seed: int, path_major: bool = False) -> np.ndarray:
    """Prices at dates 0 .. n_steps (date 0 = spot): (n_steps + 1, n_paths) time-major, or with path_major
    the reference's (n_paths, n_steps + 1) C-order array, written in that layout by the kernel."""
    out = _path_matrix(n_paths, n_steps, path_major)
    _check(lib().olmc_gbm_paths(S, T, r, sigma, q, int(n_paths), int(n_steps), seed64(seed), int(path_major),
                                out.ctypes.data_as(C.POINTER(C.c_double))))
    return out


def heston_paths(S, T, r, q, kappa, theta, sigma_v, rho, v0, n_paths: int, n_steps: int, seed: int, path_major: bool = False):
    """Spot and variance at dates 0 .. n_steps (date 0 = (S, v0)); layouts as gbm_paths."""
    spot = _path_matrix(n_paths, n_steps, path_major)
    var = np.empty_like(spot)
    _check(lib().olmc_heston_paths(S, T, r, q, kappa, theta, sigma_v, rho, v0, int(n_paths), int(n_steps), seed64(seed), int(path_major),
                                   spot.ctypes.data_as(C.POINTER(C.c_double)), var.ctypes.data_as(C.POINTER(C.c_double))))
    return spot, var


def jump_paths(S, T, r, sigma, q, kou: bool, lambda_j, a1, a2, a3, n_paths: int, n_steps: int, seed: int,
               path_major: bool = False) -> np.ndarray:
    """Prices of the jump-diffusion recursion at dates 0 .. n_steps (date 0 = spot); layouts as gbm_paths."""
    out = _path_matrix(n_paths, n_steps, path_major)
    _check(lib().olmc_jump_paths(S, T, r, sigma, q, int(bool(kou)), lambda_j, a1, a2, a3, int(n_paths), int(n_steps), seed64(seed),
                                 int(path_major), out.ctypes.data_as(C.POINTER(C.c_double))))
    return out


def exercise_boundary(S, K, T, r, sigma, q, is_call: bool, n_paths: int, n_steps: int, seed: int) -> np.ndarray:
    out = np.empty(int(n_steps) + 1, dtype=np.float64)
    _check(lib().olmc_exercise_boundary(S, K, T, r, sigma, q, int(is_call), int(n_paths), int(n_steps), seed64(seed),
                                        out.ctypes.data_as(C.POINTER(C.c_double))))
    return out


def european_cv(S, K, T, r, sigma, q, is_call: bool, n_paths: int, n_steps: int, seed: int,
                antithetic: bool = True) -> CvMoments:
    out = CvMoments()
    _check(lib().olmc_european_cv(S, K, T, r, sigma, q, int(is_call), int(n_paths), int(n_steps), seed64(seed),
                                  int(antithetic), C.byref(out)))
    return out


def european_cv_shard(S, K, T, r, sigma, q, is_call: bool, path_offset: int, n_local: int, n_steps: int, seed: int,
                      antithetic: bool = True) -> CvMoments:
    out = CvMoments()
    _check(lib().olmc_european_cv_shard(S, K, T, r, sigma, q, int(is_call), int(path_offset), int(n_local), int(n_steps), seed64(seed),
                                        int(antithetic), C.byref(out)))
    return out


def combine_cv(parts, S, T, r, q) -> CvMoments:
    """Control-variate estimate from per-shard moments (pure host function; loads the library, not the GPU)."""
    arr = (CvMoments * len(parts))(*parts)
    out = CvMoments()
    _check(load_library().olmc_combine_cv(arr, len(parts), S, T, r, q, C.byref(out)))
    return out


def _u32(a: np.ndarray):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint32))


def _sobol_args(sv, shift, point_offset: int, n_paths: int):
    """The tables as C pointers -- after checking that the point range lies inside the columns the table KNOWS: tables derived from
    the engine's public behaviour (monte_carlo.SobolDirections.valid_bits < 30) hold zeros beyond, and a point index there would
    silently repeat earlier points.  Plain arrays count as complete (30 bits)."""
    bits = int(getattr(sv, "valid_bits", 30))
    if int(point_offset) + int(n_paths) > (1 << bits):
        raise ValueError(f"Sobol points [{int(point_offset)}, {int(point_offset) + int(n_paths)}) reach beyond the 2**{bits} points these tables were "
                         f"derived for: ask sobol_tables(n_steps, seed, n_points=point_offset + n_paths)")
    sv, psv = _u32(sv)
    shift, psh = _u32(shift)
    return sv, psv, shift, psh


def european_qmc(S, K, T, r, sigma, q, is_call: bool, n_paths: int, sv: np.ndarray, shift: np.ndarray,
                 point_offset: int = 0) -> Stats:
    """sv: (dims, 30) uint32 scrambled direction matrix, shift: (dims,) uint32 (see olmc.h)."""
    sv, psv, shift, psh = _sobol_args(sv, shift, point_offset, n_paths)
    out = Stats()
    _check(lib().olmc_european_qmc(S, K, T, r, sigma, q, int(is_call), int(point_offset), int(n_paths), int(sv.shape[0]),
                                   psv, psh, int(sv.shape[1]), C.byref(out)))
    return out


def european_qmc_cv(S, K, T, r, sigma, q, is_call: bool, n_paths: int, sv: np.ndarray, shift: np.ndarray, point_offset: int = 0) -> CvMoments:
    sv, psv, shift, psh = _sobol_args(sv, shift, point_offset, n_paths)
    out = CvMoments()
    _check(lib().olmc_european_qmc_cv(S, K, T, r, sigma, q, int(is_call), int(point_offset), int(n_paths), int(sv.shape[0]), psv, psh,
                                      int(sv.shape[1]), C.byref(out)))
    return out


def european_qmc_batch(options: Sequence[Tuple[float, float, float, float, float, float, bool]], n_paths: int, sv: np.ndarray, shift: np.ndarray,
                       point_offset: int = 0) -> List[Stats]:
    """k <= 16 contracts (S, K, T, r, sigma, q, is_call) on the same Sobol points, one launch (olmc_european_qmc_batch)."""
    sv, psv, shift, psh = _sobol_args(sv, shift, point_offset, n_paths)
    k = len(options)
    arr = (Option * k)(*[Option(S, K, T, r, v, q, int(c), 0) for (S, K, T, r, v, q, c) in options])
    out = (Stats * k)()
    _check(lib().olmc_european_qmc_batch(arr, k, int(point_offset), int(n_paths), int(sv.shape[0]), psv, psh, int(sv.shape[1]), out))
    return list(out)


def european_qmc_greeks_fd(S, K, T, r, sigma, q, is_call: bool, n_paths: int, sv: np.ndarray, shift: np.ndarray, second_order: bool,
                           want_evals: bool = True) -> Tuple[List[float], List[Stats]]:
    """As european_greeks_fd, on the Sobol points of a MCMethod.QMC pricer: the 8 / 14 bumped contracts in ONE launch."""
    sv, psv, shift, psh = _sobol_args(sv, shift, 0, n_paths)
    out9 = (C.c_double * 9)()
    evals = (Stats * 14)() if want_evals else None
    _check(lib().olmc_european_qmc_greeks_fd(S, K, T, r, sigma, q, int(is_call), int(n_paths), int(sv.shape[0]), psv, psh, int(sv.shape[1]),
                                             int(second_order), out9, evals))
    return list(out9), (list(evals) if want_evals else [])


def european_qmc_terminal(S, T, r, sigma, q, n_paths: int, sv: np.ndarray, shift: np.ndarray,
                          point_offset: int = 0, antithetic: bool = False) -> np.ndarray:
    sv, psv, shift, psh = _sobol_args(sv, shift, point_offset, n_paths)
    out = np.empty(int(n_paths) * (2 if antithetic else 1), dtype=np.float64)
    _check(lib().olmc_european_qmc_terminal(S, T, r, sigma, q, int(point_offset), int(n_paths), int(sv.shape[0]), psv, psh,
                                            int(sv.shape[1]), int(antithetic), out.ctypes.data_as(C.POINTER(C.c_double))))
    return out


def asian(S, K, T, r, sigma, q, is_call: bool, geometric: bool, n_paths: int, n_steps: int, seed: int,
          antithetic: bool = False, path_offset: int = 0, fast: bool = False) -> Stats:
    """fast=True (arithmetic only): the fp32-exponent kernel (OLMC_AVG_ARITHMETIC_FAST); default = reference precision."""
    out = Stats()
    kind = AVG_GEOMETRIC if geometric else (AVG_ARITHMETIC_FAST if fast else AVG_ARITHMETIC)
    _check(lib().olmc_asian(S, K, T, r, sigma, q, int(is_call), kind,
                            int(path_offset), int(n_paths), int(n_steps), seed64(seed), int(antithetic), C.byref(out)))
    return out


def asian_greeks_fd(S, K, T, r, sigma, q, is_call: bool, n_paths: int, n_steps: int, seed: int, antithetic: bool, second_order: bool,
                    want_evals: bool = True, geometric: bool = False) -> Tuple[List[float], List[Stats]]:
    """As european_greeks_fd for the Asian option (arithmetic at the reference's precision, or geometric): the 8 / 14 bumped contracts in
    ONE launch."""
    out9 = (C.c_double * 9)()
    evals = (Stats * 14)() if want_evals else None
    _check(lib().olmc_asian_greeks_fd(S, K, T, r, sigma, q, int(is_call), AVG_GEOMETRIC if geometric else AVG_ARITHMETIC, int(n_paths), int(n_steps),
                                      seed64(seed), int(antithetic), int(second_order), out9, evals))
    return list(out9), (list(evals) if want_evals else [])


LOOKBACK_FLOATING, LOOKBACK_FIXED = 4, 5


def extrema_greeks_fd(S, K, T, r, sigma, q, is_call: bool, payoff: int, barrier: float, n_paths: int, n_steps: int, seed: int, antithetic: bool,
                      second_order: bool, want_evals: bool = True) -> Tuple[List[float], List[Stats]]:
    """As european_greeks_fd for a barrier (payoff = BARRIER_KINDS value) or lookback (LOOKBACK_FLOATING / LOOKBACK_FIXED) option: ONE launch."""
    out9 = (C.c_double * 9)()
    evals = (Stats * 14)() if want_evals else None
    _check(lib().olmc_extrema_greeks_fd(S, K, T, r, sigma, q, int(is_call), int(payoff), float(barrier), int(n_paths), int(n_steps), seed64(seed),
                                        int(antithetic), int(second_order), out9, evals))
    return list(out9), (list(evals) if want_evals else [])


BARRIER_KINDS = {"up-and-out": 0, "up-and-in": 1, "down-and-out": 2, "down-and-in": 3}


def barrier(S, K, T, r, sigma, q, is_call: bool, level: float, kind: int, n_paths: int, n_steps: int, seed: int,
            antithetic: bool = False, path_offset: int = 0) -> Stats:
    out = Stats()
    _check(lib().olmc_barrier(S, K, T, r, sigma, q, int(is_call), float(level), int(kind), int(path_offset), int(n_paths),
                              int(n_steps), seed64(seed), int(antithetic), C.byref(out)))
    return out


def lookback(S, K, T, r, sigma, q, is_call: bool, fixed_strike: bool, n_paths: int, n_steps: int, seed: int,
             antithetic: bool = False, path_offset: int = 0) -> Stats:
    out = Stats()
    _check(lib().olmc_lookback(S, K, T, r, sigma, q, int(is_call), int(fixed_strike), int(path_offset), int(n_paths),
                               int(n_steps), seed64(seed), int(antithetic), C.byref(out)))
    return out


def autocallable(S, T, r, sigma, q, autocall_barrier, coupon_barrier, coupon_rate, ki_barrier, observation_freq: int,
                 n_paths: int, n_steps: int, seed: int, antithetic: bool = False, path_offset: int = 0) -> Stats:
    out = Stats()
    _check(lib().olmc_autocallable(S, T, r, sigma, q, autocall_barrier, coupon_barrier, coupon_rate, ki_barrier,
                                   int(observation_freq), int(path_offset), int(n_paths), int(n_steps), seed64(seed),
                                   int(antithetic), C.byref(out)))
    return out


def cliquet(S, T, r, sigma, q, local_cap, local_floor, global_cap, global_floor, n_periods: int, n_paths: int, n_steps: int,
            seed: int, antithetic: bool = False, path_offset: int = 0) -> Stats:
    out = Stats()
    _check(lib().olmc_cliquet(S, T, r, sigma, q, local_cap, local_floor, global_cap, global_floor, int(n_periods),
                              int(path_offset), int(n_paths), int(n_steps), seed64(seed), int(antithetic), C.byref(out)))
    return out


def american_lsm(S, K, T, r, sigma, q, is_call: bool, n_paths: int, n_steps: int, poly_degree: int, seed: int) -> Stats:
    out = Stats()
    _check(lib().olmc_american_lsm(S, K, T, r, sigma, q, int(is_call), int(n_paths), int(n_steps), int(poly_degree),
                                   seed64(seed), C.byref(out)))
    return out


def jump_diffusion(S, K, T, r, sigma, q, is_call: bool, kou: bool, lambda_j, a1, a2, a3, n_paths: int, n_steps: int, seed: int,
                   path_offset: int = 0) -> Stats:
    """Merton: (a1, a2) = (mu_j, sigma_j), a3 ignored; Kou: (a1, a2, a3) = (p, eta1, eta2)."""
    out = Stats()
    _check(lib().olmc_jump_diffusion(S, K, T, r, sigma, q, int(is_call), int(kou), lambda_j, a1, a2, a3, int(path_offset),
                                     int(n_paths), int(n_steps), seed64(seed), C.byref(out)))
    return out


def heston(S, K, T, r, q, is_call: bool, kappa, theta, sigma_v, rho, v0, n_paths: int, n_steps: int, seed: int,
           antithetic: bool = False, path_offset: int = 0) -> Stats:
    out = Stats()
    _check(lib().olmc_heston(S, K, T, r, q, int(is_call), kappa, theta, sigma_v, rho, v0, int(path_offset), int(n_paths),
                             int(n_steps), seed64(seed), int(antithetic), C.byref(out)))
    return out


def multi_gpu_european(S, K, T, r, sigma, q, is_call: bool, n_paths: int, n_steps: int, seed: int, antithetic: bool,
                       n_gpus: int) -> Stats:
    out = Stats()
    _check(lib().olmc_multi_gpu_european(S, K, T, r, sigma, q, int(is_call), int(n_paths), int(n_steps), seed64(seed),
                                         int(antithetic), int(n_gpus), C.byref(out)))
    return out


def multi_gpu_greeks_fd(S, K, T, r, sigma, q, is_call: bool, n_paths: int, n_steps: int, seed: int, second_order: bool, n_gpus: int,
                        want_evals: bool = True) -> Tuple[List[float], List[Stats]]:
    """As european_greeks_fd over n_gpus devices of this process: one launch per rank, ONE all-reduce of the 2 nsets + 1 sums."""
    out9 = (C.c_double * 9)()
    evals = (Stats * 14)() if want_evals else None
    _check(lib().olmc_multi_gpu_greeks_fd(S, K, T, r, sigma, q, int(is_call), int(n_paths), int(n_steps), seed64(seed), int(second_order),
                                          int(n_gpus), out9, evals))
    return list(out9), (list(evals) if want_evals else [])


def multi_gpu_european_cv(S, K, T, r, sigma, q, is_call: bool, n_paths: int, n_steps: int, seed: int, antithetic: bool, n_gpus: int) -> CvMoments:
    out = CvMoments()
    _check(lib().olmc_multi_gpu_european_cv(S, K, T, r, sigma, q, int(is_call), int(n_paths), int(n_steps), seed64(seed), int(antithetic),
                                            int(n_gpus), C.byref(out)))
    return out


def multi_gpu_european_qmc(S, K, T, r, sigma, q, is_call: bool, n_paths: int, sv: np.ndarray, shift: np.ndarray, n_gpus: int) -> Stats:
    """As european_qmc over n_gpus devices of this process: rank d prices points [d N / P, (d + 1) N / P), one all-reduce of 3 sums."""
    sv, psv, shift, psh = _sobol_args(sv, shift, 0, n_paths)
    out = Stats()
    _check(lib().olmc_multi_gpu_european_qmc(S, K, T, r, sigma, q, int(is_call), int(n_paths), int(sv.shape[0]), psv, psh, int(sv.shape[1]),
                                             int(n_gpus), C.byref(out)))
    return out


def multi_gpu_european_qmc_greeks_fd(S, K, T, r, sigma, q, is_call: bool, n_paths: int, sv: np.ndarray, shift: np.ndarray, second_order: bool,
                                     n_gpus: int, want_evals: bool = True) -> Tuple[List[float], List[Stats]]:
    """As european_qmc_greeks_fd over n_gpus devices of this process: every rank prices the 8 / 14 contracts on its block of the
    Sobol points in one launch, one all-reduce of 17 / 33 sums."""
    sv, psv, shift, psh = _sobol_args(sv, shift, 0, n_paths)
    out9 = (C.c_double * 9)()
    evals = (Stats * 14)() if want_evals else None
    _check(lib().olmc_multi_gpu_european_qmc_greeks_fd(S, K, T, r, sigma, q, int(is_call), int(n_paths), int(sv.shape[0]), psv, psh,
                                                       int(sv.shape[1]), int(second_order), int(n_gpus), out9, evals))
    return list(out9), (list(evals) if want_evals else [])


def multi_gpu_european_qmc_cv(S, K, T, r, sigma, q, is_call: bool, n_paths: int, sv: np.ndarray, shift: np.ndarray, n_gpus: int) -> CvMoments:
    """As european_qmc_cv over n_gpus devices of this process: one all-reduce of the five moments and n."""
    sv, psv, shift, psh = _sobol_args(sv, shift, 0, n_paths)
    out = CvMoments()
    _check(lib().olmc_multi_gpu_european_qmc_cv(S, K, T, r, sigma, q, int(is_call), int(n_paths), int(sv.shape[0]), psv, psh, int(sv.shape[1]),
                                                int(n_gpus), C.byref(out)))
    return out


def multi_gpu_spans() -> dict:
    """Host microseconds of this thread's last multi-GPU call (olmc_multi_gpu_spans)."""
    out = (C.c_double * 8)()
    _check(lib().olmc_multi_gpu_spans(out))
    return dict(zip(("launch_us", "collective_us", "fetch_us", "drain_us", "total_us", "wake_us_max", "rank_launch_us_max", "rank_launch_us_min"), out))


def combine_stats(parts: Sequence[Tuple[float, float, int]], r: float, T: float) -> Stats:
    """Pure host function: needs the library but no device."""
    k = len(parts)
    arr = (Stats * k)(*[Stats(s, ss, int(n), 0.0, 0.0) for (s, ss, n) in parts])
    out = Stats()
    _check(load_library().olmc_combine_stats(arr, k, r, T, C.byref(out)))
    return out


def philox_words(seed: int, path_offset: int, n_paths: int, block0: int, n_blocks: int, tag: int = 0) -> np.ndarray:
    out = np.empty((int(n_paths), int(n_blocks), 4), dtype=np.uint32)
    _check(lib().olmc_philox_words(seed64(seed), int(path_offset), int(n_paths), int(block0), int(n_blocks), int(tag),
                                   out.ctypes.data_as(C.POINTER(C.c_uint32))))
    return out


def normals(seed: int, path_offset: int, n_paths: int, n_steps: int) -> np.ndarray:
    out = np.empty((int(n_paths), int(n_steps)), dtype=np.float32)
    _check(lib().olmc_normals(seed64(seed), int(path_offset), int(n_paths), int(n_steps),
                              out.ctypes.data_as(C.POINTER(C.c_float))))
    return out


TUNE_GRID_CAP = 2
TUNE_QMC_BLOCK = 4
TUNE_SPLIT_TAIL = 7
TUNE_POLL = 8
TUNE_SPLIT_SAT = 9
TUNE_MULTI_LAUNCH = 10
TUNE_STAGED_COPY = 11


def tune(knob: int, value: int) -> None:
    _check(load_library().olmc_tune(int(knob), int(value)))


def profile_enable(on: bool) -> None:
    _check(lib().olmc_profile_enable(int(on)))


def profile_reset() -> None:
    _check(lib().olmc_profile_reset())


def kernel_time() -> Tuple[int, float]:
    n, ms = C.c_int64(0), C.c_double(0.0)
    _check(lib().olmc_kernel_time(C.byref(n), C.byref(ms)))
    return n.value, ms.value
