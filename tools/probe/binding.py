"""ctypes binding of libolmc_probe.so, the INSTRUMENTED build (include/olmc_probe.h).  Test and measurement infrastructure.

`hip` is a second instance of optionslab_amd/_hip.py bound to the instrumented library: every pricing wrapper of the product
binding (hip.european, hip.multi_gpu_european, ...) then runs the instrumented build's copy of the same code, with the test seams
below available.  The functions of this module are the entry points libolmc.so does NOT have.

    from tools.probe import binding as probe
    probe.tune(probe.TUNE_MULTI_REHEARSAL, 1); probe.hip.multi_gpu_european(..., n_gpus=4)
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import sys
from typing import Optional

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
LIBRARY_PATH = os.path.join(HERE, "libolmc_probe.so")

import optionslab_amd  # noqa: E402,F401  (the package must exist for the relative imports of its _hip.py)


def _second_binding():
    spec = importlib.util.spec_from_file_location("optionslab_amd._hip_probe", os.path.join(ROOT, "optionslab_amd", "_hip.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    mod.LIBRARY_PATH = LIBRARY_PATH
    _I, _I32, _I64, _D, _U64T = C.c_int, C.c_int32, C.c_int64, C.c_double, C.c_uint64
    mod.PROTOTYPES.update({
        "olmc_exp2_probe": (_I, [C.POINTER(_D), _I64, C.POINTER(_D)]),
        "olmc_exp2_probe_form": (_I, [C.POINTER(_D), _I64, C.POINTER(_D), _I]),
        "olmc_ndtri_probe": (_I, [C.POINTER(_D), _I64, C.POINTER(_D), _I]),
        "olmc_normal_moments": (_I, [_U64T, _I64, _I64, _I32, C.POINTER(_D)]),
        "olmc_phase_stamps": (_I, [_I64, _I32, _U64T, _I32, C.POINTER(C.c_uint64), _I64, C.POINTER(_I64)]),
        "olmc_clock_probe": (_I, [_I64, _I32, _U64T, C.POINTER(_D)]),
        "olmc_issue_probe": (_I, [_I, _I, C.POINTER(_D)]),
        "olmc_probe_tune": (_I, [_I, _I]),
        "olmc_european_f64_normals": (_I, [_D] * 6 + [_I, _I64, _I32, _U64T, C.POINTER(mod.Stats)]),
        "olmc_launch_gap_probe": (_I, [_I32, C.POINTER(_D)]),
    })
    return mod


hip = _second_binding()
PROBE_PROTOTYPES = ("olmc_exp2_probe", "olmc_exp2_probe_form", "olmc_ndtri_probe", "olmc_normal_moments", "olmc_phase_stamps", "olmc_clock_probe",
                    "olmc_issue_probe", "olmc_probe_tune", "olmc_european_f64_normals", "olmc_launch_gap_probe")
_check, lib, seed64 = hip._check, hip.lib, hip.seed64

TUNE_FAULT_SHARD = 5
TUNE_FORCE_NV = 6
TUNE_MULTI_REHEARSAL = 11


def tune(knob: int, value: int) -> None:
    _check(hip.load_library().olmc_probe_tune(int(knob), int(value)))


def exp2_probe(x: np.ndarray, form: Optional[int] = None) -> np.ndarray:
    """2**x by the device's fp64 exponential: the form the Asian kernel uses, or form 0 (degree-11 polynomial) / 1 (256-entry table +
    degree 4) explicitly."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    px, py = x.ctypes.data_as(C.POINTER(C.c_double)), y.ctypes.data_as(C.POINTER(C.c_double))
    if form is None:
        _check(lib().olmc_exp2_probe(px, x.size, py))
    else:
        _check(lib().olmc_exp2_probe_form(px, x.size, py, int(form)))
    return y


def ndtri_probe(p: np.ndarray, form: int = 0) -> np.ndarray:
    """Phi^-1(p) by the Sobol kernels' inverse normal: form 0 one point, 1 coefficients in registers, 2 two in lockstep, 3 eight."""
    p = np.ascontiguousarray(p, dtype=np.float64)
    z = np.empty_like(p)
    _check(lib().olmc_ndtri_probe(p.ctypes.data_as(C.POINTER(C.c_double)), p.size, z.ctypes.data_as(C.POINTER(C.c_double)), int(form)))
    return z


def normal_moments(seed: int, n_paths: int, n_steps: int, path_offset: int = 0):
    """(sum z, sum z^2, sum z^3, sum z^4) over n_paths * n_steps normals of the device stream."""
    out = (C.c_double * 4)()
    _check(lib().olmc_normal_moments(seed64(seed), int(path_offset), int(n_paths), int(n_steps), out))
    return tuple(out)


def phase_stamps(n_paths: int = 1_000_000, n_steps: int = 252, seed: int = 42, lead_launches: int = 20):
    """(stamps[workgroups, 5]: four stamps in 100 MHz ticks + HW_ID | XCC_ID << 32, final stamp, first split workgroup, dispatch ns) of
    one instrumented launch (olmc_phase_stamps)."""
    cap = 5 * ((int(n_paths) + 63) // 64 + 1024) + 1
    buf = np.zeros(cap, dtype=np.uint64)
    info = (C.c_int64 * 3)()
    _check(lib().olmc_phase_stamps(int(n_paths), int(n_steps), seed64(seed), int(lead_launches), buf.ctypes.data_as(C.POINTER(C.c_uint64)), cap, info))
    grid = int(info[0])
    return buf[:5 * grid].reshape(grid, 5).astype(np.int64), int(buf[5 * grid]), int(info[1]), int(info[2])


def clock_probe(n_paths: int = 1_000_000, n_steps: int = 252, seed: int = 42) -> dict:
    """Shader clock held under the headline kernel's load: s_memtime / s_memrealtime around the step loop, median over workgroups."""
    out = (C.c_double * 3)()
    _check(lib().olmc_clock_probe(int(n_paths), int(n_steps), seed64(seed), out))
    return dict(loop_cycles=out[0], loop_ticks_100mhz=out[1], ghz=out[2])


PROBE_CLASSES = ("v_mad_u64_u32", "v_bitop3_b32", "v_cvt_f32_u32", "v_fmamk_f32", "v_and_or_b32", "v_log_f32", "v_sqrt_f32",
                 "v_sin_f32", "v_cos_f32", "v_exp_f32", "v_add_f32", "v_fma_f32", "v_cvt_f64_f32", "v_add_f64", "v_fma_f64",
                 "v_rndne_f64", "v_ldexp_f64", "v_cvt_i32_f64",
                 "pair:v_log_f32+v_add_f32", "pair:v_log_f32+v_bitop3_b32", "v_bitop3_b32(v,v,v)", "v_bitop3_b32(v,v,const)", "v_xor_b32(v,v)",
                 "pair:v_bitop3_b32+v_add_u32", "pair:v_mad_u64_u32+v_bitop3_b32", "v_mad_u64_u32(v,v)")   # order = the OLMC_PROBE_* enum of include/olmc_probe.h


def issue_probe(waves_per_simd: int = 8) -> dict:
    """{instruction class: ns per wave64 instruction per SIMD} measured on this device (olmc_issue_probe)."""
    out = {}
    for op, name in enumerate(PROBE_CLASSES):
        ns = C.c_double(0.0)
        _check(lib().olmc_issue_probe(op, int(waves_per_simd), C.byref(ns)))
        out[name] = ns.value
    return out


def european_f64_normals(S, K, T, r, sigma, q, is_call: bool, n_paths: int, n_steps: int, seed: int):
    """The European call with fp64 normals (not a product path; see include/olmc_probe.h)."""
    out = hip.Stats()
    _check(lib().olmc_european_f64_normals(S, K, T, r, sigma, q, int(is_call), int(n_paths), int(n_steps), seed64(seed), C.byref(out)))
    return out


def launch_gap_us(n: int = 2000) -> float:
    us = C.c_double(0.0)
    _check(lib().olmc_launch_gap_probe(int(n), C.byref(us)))
    return us.value
