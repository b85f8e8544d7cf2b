#!/usr/bin/env python3
"""bench.py -- headline benchmark: GBM path-steps/s, European call, antithetic on, on-device reduction.

    python bench.py --gpus N --steps K --warmup W

N = 1: BASELINE.json configs[1] -- 1M paths x 252 steps on one GPU.  N > 1: configs[4] -- 8M paths x 252 steps per GPU,
contiguous global path ranges, ONE RCCL all-reduce of the (sum, sumsq, n) triple per pricing (weak scaling); the 1M-per-GPU
figure and the one-GPU basis of the same 8M workload travel as secondary keys of the same line.

ONE pass defines the line: a "step" is one complete BLOCKING pricing --
`MonteCarloPricer.price(S, K, T, r, sigma, "call", seed=..., return_error=True)` at N = 1 (one path kernel with its
fused reduction, result on the host), and at N > 1 the same thing sharded: this rank's kernel on its block of the
global paths -> the all-reduce -> result on the host of every rank.  After W untimed warm-up steps a pass times EXACTLY
K steps between barrier + synchronise fences (max over ranks); passes repeat until >= 1.5 s are covered, and

    value                  = global paths x 252 x K / median pass time
    ms_per_step            = median pass time / K
    roofline.avg_kernel_ms = HIP events attached to the dispatches of those very calls

so `avg_kernel_ms <= ms_per_step` by construction.  `roofline.frac` is an issue-cycle fraction: VALU issue cycles the
kernel's instruction stream needs per launch (rocprofv3 PMC, collected LIVE by a child pass of this script before the
parent touches the GPU) over the SIMD-cycles of the measured kernel time at the 2.4 GHz peak clock -- <= 1 by construction.

stdout carries ONE JSON line of <= 3 KB (`compact_line`): the contract keys, `roofline`, `cpu_baseline` and
{value, ms, frac} entries for configs[2] (`c3`), [3] (`c4`), [4] (`c5`).  Everything else -- issue tables, instruction
mixes, per-pass seconds, probes -- goes to the sidecar file the line names in `detail` (bench_detail.json next to this
script; copies of record live under profiles/).

torch is plumbing here (device result slots, stream handle, process group); the compute is libolmc.so through its
C ABI.  With N > 1 and no launcher in the environment (WORLD_SIZE unset) this script starts its own N ranks as a child
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py ...` BEFORE
anything touches a GPU and relays rank 0's line and the exit code; under an external torch.distributed.run it is a rank.
"""
import argparse
import csv
import faulthandler
import glob
import json
import math
import os
import shutil
import socket
import statistics
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORK = dict(S=100.0, K=100.0, T=1.0, r=0.05, sigma=0.2, q=0.0, option_type="call")
ATM = (100.0, 100.0, 1.0, 0.05, 0.2)
PATHS_PER_GPU = 1_000_000
N_STEPS = 252
ASIAN_STEPS = 1024
C5_PATHS_PER_GPU = 8_000_000
C5_TOTAL = 64_000_000
SEED = 42
PRE_WARM_PASSES = 45          # x 32 pricings of 1M x 252: about 150 ms of load before anything is timed
PEAK_GHZ = 2.4                # MI355X_MICROARCH.md chip table: max clock 2400 MHz
N_SIMD = 256 * 4              # 256 CU x 4 SIMD
LANE_OPS_PER_PATH_STEP = 32   # SURVEY 8(d)'s algorithmic count (kept as a secondary, model figure)
PEAK_TLANEOPS = N_SIMD * 32 * PEAK_GHZ * 1e9 / 1e12   # 78.6: SIMD-32 x 2.4 GHz (guide: 157.3 TF fp32 vector = 2 flop/FMA);
                                                      # SURVEY 8(d) assumed 64 lanes x 256 CU x 2.4 GHz = 39.3 (a SIMD-16 machine)

def device_sources_sha256():
    import hashlib
    h = hashlib.sha256()
    for name in ("olmc.hip", "olmc_kernels.h", "olmc_host_math.h"):
        with open(os.path.join(ROOT, "optionslab_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def load_isa_mix():
    """Static instruction mix of each kernel's step loop (tools/isa_mix.py: hipcc -S of this build, parsed).  The file names the
    sha256 of the device sources it was read from: a mix of ANOTHER build (isa_mix.py failed after a kernel change, build.py only
    warns) is refused -- {"_stale": why} -- and every roofline.frac that would have been priced with it is null with that reason."""
    try:
        with open(os.path.join(ROOT, "optionslab_amd", "isa_mix.json")) as f:
            mix = json.load(f)
    except OSError:
        return {"_stale": "optionslab_amd/isa_mix.json is missing (run tools/isa_mix.py)"}
    try:
        now = device_sources_sha256()
    except OSError:
        return mix                      # a packaged copy without sources: nothing to compare with
    if mix.get("_sources_sha256") != now:
        return {"_stale": f"optionslab_amd/isa_mix.json belongs to other device sources (its sha256 {str(mix.get('_sources_sha256'))[:12]}, "
                          f"now {now[:12]}): run tools/isa_mix.py"}
    return mix


PMC_KERNELS = {         # substring of the demangled kernel name -> key in the JSON
    "european_path_kernel<1, true, 0, false>": "c2_european",
    "european_path_kernel<8, true, 3, false>": "c3_fused8",        # MODE 3 = kSumOnly: what MonteCarloPricer.greeks() launches (prices only)
    "european_path_kernel<16, true, 3, false>": "c3_fused14",
    "asian_exp64_kernel<false>": "c4_asian_fp64",
    "asian_exp64_kernel<true>": "c4_asian_fp64_antithetic",
    "asian_kernel<false, false>": "c4_asian_fp32",
    "asian_kernel<true, false>": "c4_asian_fp32_antithetic",
    "asian_exp64_greeks_kernel<false, 16>": "c4_asian_greeks14",
    # SURVEY 8(f) kernels (round 4)
    "asian_kernel<false, true>": "f_asian_geometric",
    "extrema_kernel<false>": "f_extrema",
    "heston_kernel<false>": "f_heston",
    "european_multi_kernel<true>": "f_multi",
    "european_qmc_kernel<0, true, true>": "f_qmc",         # split workgroups, aligned form: what every launch below 2^22 points runs (2^21 below 128 dimensions; >= 32 dimensions, offset 0)
    "european_qmc_block_kernel<0, true>": "f_qmc_block",   # eight points per thread, aligned form (offset 0)
    # round 5: the kernels round 4 added without a fraction (VERDICT r4 "missing" 2) and the control-variate shape of the headline kernel
    "extrema_greeks_kernel<false, 16>": "f_extrema_greeks14",
    "extrema_greeks_kernel<true, 16>": "f_extrema_greeks14a",
    "asian_geometric_greeks_kernel<false, 16>": "f_geo_greeks14",
    "autocall_kernel<false>": "f_autocall",
    "cliquet_kernel<false>": "f_cliquet",
    "european_path_kernel<1, true, 2, false>": "f_cv",              # MODE 2 = kControlVariate: five moments per path (monte_carlo.py:154-186)
}
# the 8(f) workloads: (key, what, unit-steps per launch for the issue model = (paths or threads, steps or dims))
F_PATHS, F_STEPS = 1_000_000, 252
F_MULTI_CONTRACTS, F_MULTI_PATHS, F_MULTI_STEPS = 64, 65_536, 64
F_QMC_POINTS, F_QMC_DIMS = 1 << 17, 252
F_QMC_BLOCK_POINTS, F_QMC_BLOCK_DIMS = 1 << 22, 64


def f_workloads(ol, _hip):
    """{key: (callable making ONE blocking launch of that kernel, description, n_paths for the issue model, n_steps)}."""
    import numpy as np
    from optionslab_amd.monte_carlo import sobol_tables
    from optionslab_amd.monte_carlo_unified import MonteCarloPricerUni
    P = ATM + (0.0,)
    sv1, sh1 = sobol_tables(F_QMC_DIMS, SEED)
    sv8, sh8 = sobol_tables(F_QMC_BLOCK_DIMS, SEED)
    uni = MonteCarloPricerUni(F_MULTI_PATHS, F_MULTI_STEPS, seed=SEED)
    ks = np.linspace(80.0, 120.0, F_MULTI_CONTRACTS)
    ones = np.ones(F_MULTI_CONTRACTS)
    return {
        "f_heston": (lambda: _hip.heston(100.0, 100.0, 1.0, 0.05, 0.0, True, 2.0, 0.04, 0.3, -0.7, 0.04, F_PATHS, F_STEPS, SEED, False),
                     f"HestonPricer.price_monte_carlo, {F_PATHS:,} paths x {F_STEPS} steps (heston.py:184-255)", F_PATHS, F_STEPS),
        "f_extrema": (lambda: _hip.barrier(*P, True, 120.0, 0, F_PATHS, F_STEPS, SEED, False),
                      f"BarrierOption.price up-and-out call, {F_PATHS:,} paths x {F_STEPS} dates (exotic_options.py:174-224)", F_PATHS, F_STEPS),
        "f_asian_geometric": (lambda: _hip.asian(*P, True, True, F_PATHS, ASIAN_STEPS, SEED, False),
                              f"AsianOption.price geometric call, {F_PATHS:,} paths x {ASIAN_STEPS} dates (exotic_options.py:121-122)", F_PATHS, ASIAN_STEPS),
        "f_multi": (lambda: uni.price_batch(100.0 * ones, ks, ones, 0.05 * ones, 0.2 * ones, "call"),
                    f"MonteCarloPricerUni.price_batch, {F_MULTI_CONTRACTS} contracts x {F_MULTI_PATHS:,} paths x {F_MULTI_STEPS} steps, one launch "
                    "(monte_carlo_unified.py:562-631)", F_MULTI_CONTRACTS * F_MULTI_PATHS, F_MULTI_STEPS),
        "f_qmc": (lambda: _hip.european_qmc(*P, True, F_QMC_POINTS, sv1, sh1),
                  f"MCMethod.QMC price, 2^17 Sobol points x {F_QMC_DIMS} dims, 64 points per workgroup, a quarter of the dims per wave, high Gray bits "
                  "folded once per wave and dim (gbm_qmc.py:14-46)", F_QMC_POINTS, F_QMC_DIMS),
        "f_qmc_block": (lambda: _hip.european_qmc(*P, True, F_QMC_BLOCK_POINTS, sv8, sh8),
                        f"MCMethod.QMC price, 2^22 Sobol points x {F_QMC_BLOCK_DIMS} dims, eight points per thread (gbm_qmc.py:14-46)",
                        F_QMC_BLOCK_POINTS // 8, F_QMC_BLOCK_DIMS),
        "f_extrema_greeks14": (lambda: _hip.extrema_greeks_fd(*P, True, 0, 120.0, F_PATHS, F_STEPS, SEED, False, True, want_evals=False),
                               f"compute_greeks_unified(ExoticAdapter(BarrierOption up-and-out)), second order: 14 contracts = six recursions, ONE launch, "
                               f"{F_PATHS:,} paths x {F_STEPS} dates (unified_greeks.py:177-227 over exotic_options.py:163-224)", F_PATHS, F_STEPS),
        "f_extrema_greeks14a": (lambda: _hip.extrema_greeks_fd(*P, True, 0, 120.0, F_PATHS, F_STEPS, SEED, True, True, want_evals=False),
                                    f"the same with antithetic legs (twelve recursions per thread), {F_PATHS:,} paths x {F_STEPS} dates", F_PATHS, F_STEPS),
        "f_geo_greeks14": (lambda: _hip.asian_greeks_fd(*P, True, F_PATHS, ASIAN_STEPS, SEED, False, True, want_evals=False, geometric=True),
                                 f"compute_greeks_unified(ExoticAdapter(AsianOption geometric)), second order, ONE launch, {F_PATHS:,} paths x {ASIAN_STEPS} dates "
                                 "(exotic_options.py:119-122)", F_PATHS, ASIAN_STEPS),
        "f_autocall": (lambda: _hip.autocallable(100.0, 1.0, 0.05, 0.2, 0.0, 1.0, 0.8, 0.08, 0.6, 21, F_PATHS, F_STEPS, SEED),
                       f"AutocallableOption.price, monthly observation, {F_PATHS:,} paths x {F_STEPS} dates (exotic_options.py:404-491)", F_PATHS, F_STEPS),
        "f_cliquet": (lambda: _hip.cliquet(100.0, 1.0, 0.05, 0.2, 0.0, 0.05, -0.05, 0.3, 0.0, 12, F_PATHS, F_STEPS, SEED),
                      f"CliquetOption.price, 12 periods, {F_PATHS:,} paths x {F_STEPS} dates (exotic_options.py:494-554)", F_PATHS, F_STEPS),
        "f_cv": (lambda: _hip.european_cv(*P, True, F_PATHS, F_STEPS, SEED, True),
                 f"MonteCarloPricer.price_with_control_variate, {F_PATHS:,} paths x {F_STEPS} steps, antithetic: five moments per path "
                 "(monte_carlo.py:154-186)", F_PATHS, F_STEPS),
    }
PMC_PASSES = [("sq", ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]),
              ("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"])]


# ------------------------------------------------------------------------------------------------ PMC (child + parent)
def pmc_child():
    """The workload rocprofv3 counts: a few launches of every kernel bench.py reports a roofline for.  No torch."""
    import optionslab_amd as ol
    p = ol.MonteCarloPricer(PATHS_PER_GPU, N_STEPS, SEED)
    for i in range(12):
        p.price(*ATM, "call", seed=SEED + i, return_error=True)
    p8 = ol.MonteCarloPricer(C5_PATHS_PER_GPU, N_STEPS, SEED)       # configs[4]'s shard: the same kernel, another launch shape
    for i in range(4):
        p8.price(*ATM, "call", seed=SEED + i, return_error=True)
    for second in (False, True):
        for _ in range(4):
            p.greeks(*ATM, "call", include_second_order=second)
    a = ol.AsianOption(*ATM, seed=SEED)
    for precision in ("fp64", "fp32"):
        for anti in (False, True):
            for _ in range(3):
                a.price(PATHS_PER_GPU, ASIAN_STEPS, "arithmetic", "call", antithetic=anti, precision=precision)
    ad = ol.ExoticAdapter(ol.AsianOption(*ATM, seed=SEED), n_paths=PATHS_PER_GPU, n_steps=ASIAN_STEPS)
    for _ in range(3):
        ol.compute_greeks_unified(ad, *ATM, "call", include_second_order=True)
    from optionslab_amd import _hip
    for _key, (fn, _what, _n, _m) in f_workloads(ol, _hip).items():
        for _ in range(3):
            fn()
    print("pmc-child done", flush=True)


def _row_threads(row):
    """Threads of a dispatch row of rocprofv3's counter_collection.csv (Grid_Size = total work-items)."""
    try:
        return int(float(row.get("Grid_Size") or row.get("Grid_Size_X") or 0))
    except ValueError:
        return 0


def collect_pmc(keep_dir=None):
    """Runs `rocprofv3 --pmc ... -- python3 bench.py --pmc-child` once per counter pass (separate passes, as the guide
    prescribes) and returns {key: {counter: mean per dispatch}} -- or {"error": ...}.  Called BEFORE this process touches
    the GPU."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    base = os.path.abspath(keep_dir) if keep_dir else tempfile.mkdtemp(prefix="olmc_pmc_", dir="/tmp")   # the passes run with cwd = /tmp
    os.makedirs(base, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    env.pop("RANK", None)
    out = {}
    t0 = time.perf_counter()
    for name, counters in PMC_PASSES:
        d = os.path.join(base, name)
        cmd = [exe, "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", d, "--",
               sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child"]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=90)       # a pass normally takes under a second
        except (OSError, subprocess.TimeoutExpired) as e:
            return {"error": f"pass {name}: {type(e).__name__}: {e}"}
        if r.returncode != 0 or "pmc-child done" not in r.stdout:
            return {"error": f"pass {name}: rc {r.returncode}: {(r.stderr or r.stdout)[-400:]}"}
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            return {"error": f"pass {name}: no counter_collection.csv under {d}"}
        agg = {}
        for f in files:
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    key = next((k for pat, k in PMC_KERNELS.items() if pat in row["Kernel_Name"]), None)
                    if key == "c2_european" and _row_threads(row) > 4 * PATHS_PER_GPU:
                        key = "c5_shard"                  # the 8M-path launches of the same kernel (configs[4]'s per-GPU shard)
                    if key:
                        agg.setdefault((key, row["Counter_Name"]), []).append(float(row["Counter_Value"]))
        for (key, ctr), vals in agg.items():
            out.setdefault(key, {})[ctr] = sum(vals) / len(vals)
            out[key]["dispatches_" + name] = len(vals)
    out["seconds"] = time.perf_counter() - t0
    out["source"] = "live: rocprofv3 --kernel-trace --pmc child passes of this run (bench.py --pmc-child), mean per dispatch"
    if keep_dir:
        with open(os.path.join(base, "pmc.json"), "w") as f:
            json.dump(out, f, indent=1)
    else:
        shutil.rmtree(base, ignore_errors=True)
    return out


def committed_pmc():
    """Fallback when the live passes are unavailable: the newest profiles/r*_pmc.json (same layout), labelled as such."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        d = json.load(f)
    d["source"] = f"COMMITTED {os.path.relpath(files[-1], ROOT)} (another run of the same build; the live rocprofv3 passes failed)"
    return d


# Issue passes of a wave64 VALU instruction on a gfx950 SIMD (32 lanes wide): the MINIMUM number of cycles the SIMD's vector
# issue port is held, taken per class as the most optimistic figure there is evidence for.
#   2 = full rate: plain fp32 / int32 operations whose operands are VGPRs or literals -- v_add_f32, v_fmamk_f32, v_bitop3_b32 on
#       three VGPRs (olmc_issue_probe: 2.3-2.6 cycles at the clock held) and v_fma_f32 (the guide's "v_fma_f32 (wave64) 2 cyc";
#       the probe's own form read 3.9); anything unclassified is priced here too;
#   8 = transcendental unit (quarter rate; probes 7.4-8.3);
#   4 = everything else, in the cheapest operand form probed: the 64-bit multiply-add (4.5 with an SGPR or a VGPR multiplier
#       alike), VOP3 integer ops with an SGPR operand (4.1-4.4), conversions (3.9-4.2), every fp64 operation (4.1-5.0 = the
#       78.6 TF fp64 vector peak).
# The live olmc_issue_probe figures travel in the JSON (issue_costs_ns) as the evidence that no class issues faster than its entry.
ISSUE_PASSES = {"v_mad_u64_u32": 4, "v_bitop3_b32": 4, "v_bitop3_b32(v,v,v)": 2, "v_cvt_f32_u32": 4, "v_fmamk_f32": 2, "v_and_or_b32": 4,
                "v_log_f32": 8, "v_sqrt_f32": 8, "v_sin_f32": 8, "v_cos_f32": 8, "v_exp_f32": 8, "v_add_f32": 2, "v_fma_f32": 2,
                "v_cvt_f64_f32": 4, "v_add_f64": 4, "v_fma_f64": 4, "v_rndne_f64": 4, "v_ldexp_f64": 4, "v_cvt_i32_f64": 4, "other": 2}
OUTSIDE_LOOP_PASSES = 4       # per-path prologue / epilogue (fp64 exp, payoff, reduction): fp64 class


def roofline_for(pmc, key, avg_kernel_s, n_steps, n_paths, costs=None, mixes=None, clock_ghz=None):
    """Issue-cycle roofline of one kernel, <= 1 by construction.

    achieved = VALU issue cycles the kernel's instruction stream NEEDS per launch / measured kernel time, where
               needed cycles = sum over instructions of the class's issue passes (ISSUE_PASSES: 2 / 4 / 8).  The
               instruction stream: SQ_INSTS_VALU from the live PMC pass is the total; the step loop's share is the static
               mix of its body (tools/isa_mix.py) x trips per path x paths / 64, the remainder (per-path prologue and
               epilogue, trailing blocks, reduction) is priced at 4.
    peak     = 1024 SIMDs x 2.4 GHz (one issue cycle per SIMD per clock at the peak engine clock).
    Since no instruction can hold the port for less than its class's passes and no SIMD can run above 2.4 GHz,
    frac <= 1; what is missing from 1 is clock give-back under load (`clock`), issue bubbles and the launch's tail.
    Beside it: `frac_vs_isolated_rates` prices the same stream with the per-class issue TIMES olmc_issue_probe measured in
    isolation in this run (an empirical ceiling at the clock actually held; mixes can beat isolated classes by a few
    per cent, so it is not a bound), and `frac_valu_active_pmc` is the VALUBusy formula (SQ_ACTIVE_INST_VALU x 4 /
    (1024 SIMDs x kernel time x 2.4 GHz)) -- that counter charges 4 cycles per instruction (8 per transcendental:
    (4 n + 4 n_trans) / n reproduces its cycles-per-instruction on every kernel here to 1 %) whatever the SIMD-32
    needed, so it is not bounded by 1 either (the fp32 Asian kernel reads 1.05)."""
    c = (pmc or {}).get(key)
    if not c or "SQ_ACTIVE_INST_VALU" not in c or not avg_kernel_s:
        return None
    peak = N_SIMD * PEAK_GHZ * 1e9
    r = {"bound": "valu", "achieved": None, "peak": peak / 1e9, "unit": "G VALU issue-cycles/s (peak = 1024 SIMDs x 2.4 GHz)", "frac": None,
         "avg_kernel_ms": avg_kernel_s * 1e3, "valu_insts_per_launch": c.get("SQ_INSTS_VALU"), "waves_per_launch": c.get("SQ_WAVES"),
         "frac_valu_active_pmc": c["SQ_ACTIVE_INST_VALU"] * 4.0 / avg_kernel_s / peak,
         "pmc_cycles_per_valu_inst": (c["SQ_ACTIVE_INST_VALU"] * 4.0 / c["SQ_INSTS_VALU"]) if c.get("SQ_INSTS_VALU") else None}
    mix = (mixes or {}).get(key)
    if (mixes or {}).get("_stale"):
        r["why_null"] = mixes["_stale"]
    if mix and c.get("SQ_INSTS_VALU") and c.get("SQ_WAVES") and mix.get("steps_per_trip"):
        trips = n_steps // mix["steps_per_trip"]
        path_waves = n_paths / 64.0                     # whole-path wave equivalents (a split workgroup's four waves share 64 paths)
        rest = max(c["SQ_INSTS_VALU"] - path_waves * trips * mix["loop_valu_instructions"], 0.0)      # instructions outside the hot loop, per launch
        loop_cycles = sum(n * ISSUE_PASSES[cls] for cls, n in mix["by_class"].items())
        rest_passes = mix.get("rest_passes", OUTSIDE_LOOP_PASSES)   # a two-path loop (isa_mix.py MIN_PATH) prices what exceeds its cheapest trip at the minimum
        need = path_waves * trips * loop_cycles + rest * rest_passes                                  # issue cycles per launch, all SIMDs
        r.update({"achieved": need / avg_kernel_s / 1e9, "frac": need / avg_kernel_s / peak,
                  "issue_cycles_needed_per_launch": need, "speed_of_light_kernel_ms": need / peak * 1e3,
                  "issue_model": {"loop_trips_per_path": trips, "loop_valu_instructions": mix["loop_valu_instructions"], "loop_mix": mix["by_class"],
                                  "loop_issue_cycles_per_trip": loop_cycles, "issue_passes": ISSUE_PASSES,
                                  "valu_instructions_in_the_loop_per_launch": path_waves * trips * mix["loop_valu_instructions"],
                                  "valu_instructions_outside_the_loop_per_launch": rest, "outside_the_loop_passes": rest_passes}})
        if costs:
            loop_ns = sum(n * costs.get(cls, costs["v_fmamk_f32"]) for cls, n in mix["by_class"].items())
            model_s = (path_waves * trips * loop_ns + rest * costs["v_fma_f64"]) / N_SIMD * 1e-9
            r["frac_vs_isolated_rates"] = model_s / avg_kernel_s
            r["isolated_rates_kernel_ms"] = model_s * 1e3
            if clock_ghz:
                r["issue_costs_cycles_at_held_clock"] = {k: v * clock_ghz for k, v in costs.items()}
    if clock_ghz:
        r["clock_ghz_under_load"] = clock_ghz         # s_memtime / s_memrealtime around the headline kernel's step loop (`clock`)
    if c.get("SQ_BUSY_CYCLES"):
        r["sq_busy_cycles_per_se"] = c["SQ_BUSY_CYCLES"] / 32.0        # ~ the kernel's duration in shader cycles (32 shader engines)
    fetch, write = c.get("FETCH_SIZE"), c.get("WRITE_SIZE")
    if fetch is not None and write is not None:
        # rocprofv3 reports both in KB; gfx950 tallies a 128-B read request as 64 B (MI355X_MICROARCH.md, HBM): x2 on FETCH
        r["traffic"] = (2.0 * fetch + write) * 1024.0
        r["traffic_unit"] = "bytes per launch (2 x FETCH_SIZE + WRITE_SIZE)"
        r["hbm_gbps"] = r["traffic"] / avg_kernel_s / 1e9
        r["hbm_frac_of_8TBps"] = r["traffic"] / avg_kernel_s / 8e12
    else:
        r["traffic"] = None
    return r


# ------------------------------------------------------------------------------------------------ CPU baseline
def _cpu_worker(args):
    n, seed = args
    from oracle import numpy_reference as orc
    res = orc.OraclePricer(n, N_STEPS, seed).price(WORK["S"], WORK["K"], WORK["T"], WORK["r"], WORK["sigma"], "call", return_error=True)
    return res.price, res.std_error, res.n_paths


def _cpu_ready(_):
    from oracle import numpy_reference  # noqa: F401  (pays the import inside the pool, outside the timed region)
    return os.getpid()


def cpu_baseline(all_cores=True):
    """Reference-pinned NumPy oracle (oracle/numpy_reference.py == the reference bit for bit) timed on this host.
    Headline: 1 core (NumPy's Generator is serial), median of 3 full 1M x 252 pricings after a 100k warm-up.
    Beside it, labelled: `procs` worker processes each pricing 500k paths with its own seed (embarrassingly parallel)."""
    from oracle import numpy_reference as orc
    orc.OraclePricer(100_000, N_STEPS, SEED).price(WORK["S"], WORK["K"], WORK["T"], WORK["r"], WORK["sigma"], "call")
    n, times, res = 1_000_000, [], None
    for i in range(3):
        t0 = time.perf_counter()
        res = orc.OraclePricer(n, N_STEPS, SEED).price(WORK["S"], WORK["K"], WORK["T"], WORK["r"], WORK["sigma"], "call", return_error=True)
        times.append(time.perf_counter() - t0)
    dt = statistics.median(times)
    out = dict(value=n * N_STEPS / dt, unit="path-steps/s", cores=1, kind="port",
               sample=f"median of 3 x price() at {n} paths x {N_STEPS} steps, NumPy oracle pinned bitwise to the reference "
                      f"(price {res.price:.6f}); host has {os.cpu_count()} logical cores, NumPy's RNG uses 1",
               seconds=dt, seconds_each=times)
    # BASELINE configs[3] on the same core (scaled from a bounded sample): the reference's AsianOption.price builds the
    # (n, M) normal, log-return, log-price and price matrices -- 50,000 paths x 1024 dates is 1.6 GB of them
    try:
        n_a = 50_000
        t0 = time.perf_counter()
        pa = orc.asian_price(WORK["S"], WORK["K"], WORK["T"], WORK["r"], WORK["sigma"], 0.0, SEED, n_a, ASIAN_STEPS, "arithmetic", "call")
        dta = time.perf_counter() - t0
        out["c4_asian"] = dict(value=n_a * ASIAN_STEPS / dta, unit="path-steps/s", cores=1, kind="port", seconds=dta,
                               sample=f"1 x AsianOption.price({n_a} paths, {ASIAN_STEPS} dates, arithmetic call), NumPy oracle (price {float(pa):.4f})")
    except Exception as e:
        out["c4_asian"] = {"error": f"{type(e).__name__}: {e}"}
    if all_cores:
        try:
            import concurrent.futures as cf
            import multiprocessing as mp
            procs = max(1, min(os.cpu_count() or 1, 16))         # a one-GPU box's CPU share
            per = 500_000                                        # 1 GB of normals per worker
            with cf.ProcessPoolExecutor(procs, mp_context=mp.get_context("spawn")) as pool:
                list(pool.map(_cpu_ready, range(procs)))
                t0 = time.perf_counter()
                parts = list(pool.map(_cpu_worker, [(per, SEED + 100 + i) for i in range(procs)]))
                wall = time.perf_counter() - t0
            out["all_cores"] = dict(value=procs * per * N_STEPS / wall, unit="path-steps/s", cores=procs, seconds=wall,
                                    sample=f"{procs} processes x {per} paths x {N_STEPS} steps, distinct seeds, one wall clock "
                                           f"(mean price {sum(p[0] for p in parts) / procs:.4f})")
        except Exception as e:      # the baseline beside the baseline must never cost the line
            out["all_cores"] = {"error": f"{type(e).__name__}: {e}"}
    return out


# ------------------------------------------------------------------------------------------------ the line
LINE_BUDGET = 3072            # bytes of the ONE stdout line (the driver keeps an 8 KB tail of stdout; r02's 29 KB line left parsed = null)
DETAIL_FILE = os.environ.get("OLMC_BENCH_DETAIL") or os.path.join(ROOT, "bench_detail.json")


def _num(x, digits=6):
    """Floats to `digits` significant digits (the line is read by machines; 17 digits of a timing are noise)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        y = float(f"{x:.{digits}g}")
        return int(y) if abs(y) >= 1e6 and y == int(y) else y      # 2194320000000, not 2194320000000.0: two bytes a throughput
    return x


def _entry(value=None, ms=None, frac=None, **more):
    e = {"value": _num(value), "ms": _num(ms, 5)}
    if frac is not None:
        e["frac"] = _num(frac, 4)
    e.update({k: _num(v) for k, v in more.items() if v is not None})
    return e


def compact_line(full, detail_name=None):
    """The ONE stdout line, from the full record (pure: no I/O, no clocks -- tests/test_bench_line.py feeds it canned records).
    Contract keys verbatim; roofline and cpu_baseline as the task defines them; c3 / c4 / c5 as {value, ms, frac}; nothing else.
    Stays under LINE_BUDGET bytes for any record: free-text fields are clipped, error lists truncated."""
    def clip(sv, n):
        sv = str(sv)
        return sv if len(sv) <= n else sv[: n - 1] + "~"

    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                     "vs_baseline", "dtype", "data")}
    line["value"], line["ms_per_step"] = _num(line["value"], 7), _num(line["ms_per_step"], 6)
    cfg = full.get("config") or {}
    line["config"] = {"workload": clip(cfg.get("workload", ""), 220), "paths_per_gpu": cfg.get("paths_per_gpu"), "n_steps": cfg.get("n_steps"),
                      "global_paths": cfg.get("global_paths"), "parallelism": clip(cfg.get("parallelism", ""), 120)}
    line["ranks_seen"] = full.get("ranks_seen")
    line["timed_s"] = _num(sum((full.get("passes") or {}).get("seconds") or []), 4)
    acc = full.get("accuracy") or {}
    line["accuracy"] = {"max_abs_err_over_sigma": _num(acc.get("max_abs_err_over_sigma"), 4), "steps_checked": acc.get("steps_checked")}
    r = full.get("roofline") or {}
    line["roofline"] = {"bound": r.get("bound", "valu"), "achieved": _num(r.get("achieved")), "peak": _num(r.get("peak")),
                        "unit": "G VALU issue-cycles/s", "frac": _num(r.get("frac"), 4), "traffic": _num(r.get("traffic")),
                        "avg_kernel_ms": _num(r.get("avg_kernel_ms"), 5), "kernel": clip(r.get("kernel", ""), 60),
                        "pmc_source": clip(r.get("pmc_source") or r.get("why_null") or "", 60),
                        "clock_ghz_under_load": _num(r.get("clock_ghz_under_load"), 4), "hbm_gbps": _num(r.get("hbm_gbps"), 4)}
    cpu = full.get("cpu_baseline")
    if cpu:
        line["cpu_baseline"] = {"value": _num(cpu.get("value")), "unit": cpu.get("unit"), "cores": cpu.get("cores"), "kind": cpu.get("kind"),
                                "seconds": _num(cpu.get("seconds"), 4), "sample": clip(cpu.get("sample", ""), 160)}
        if isinstance(cpu.get("all_cores"), dict) and cpu["all_cores"].get("value"):
            line["cpu_baseline"]["all_cores"] = {"value": _num(cpu["all_cores"]["value"]), "cores": cpu["all_cores"].get("cores")}
        line["gpu_over_cpu"] = _num(full.get("gpu_over_cpu"), 5)

    def roof_frac(d):
        return ((d or {}).get("roofline") or {}).get("frac")

    g = full.get("c3_greeks")
    if isinstance(g, dict) and "error" not in g:
        line["c3"] = {k: _entry(v.get("path_steps_per_s"), v.get("ms_per_call"), roof_frac(v), kernel_ms=v.get("avg_kernel_ms"))
                      for k, v in g.items() if isinstance(v, dict) and "ms_per_call" in v}
    a = full.get("c4_asian")
    if isinstance(a, dict) and "error" not in a:
        line["c4"] = {k: _entry(v.get("path_steps_per_s"), v.get("ms_per_call"), roof_frac(v), kernel_ms=v.get("avg_kernel_ms"))
                      for k, v in a.items() if isinstance(v, dict) and "ms_per_call" in v}
        if a.get("gpu_over_cpu"):
            line["c4"]["gpu_over_cpu"] = _num(a["gpu_over_cpu"], 5)
        if isinstance(a.get("greeks"), dict) and "fused_14" in a["greeks"]:
            line["c4"]["greeks14"] = {"ms": _num(a["greeks"]["fused_14"]["ms"], 4), "x_literal": _num(a["greeks"].get("speedup_14"), 3),
                                      "frac": _num(roof_frac(a["greeks"]["fused_14"]), 3)}
    f = full.get("f_kernels")
    if isinstance(f, dict) and "error" not in f:
        line["f"] = {}                                  # every entry: [kernel ms (American option: blocking call ms), fraction of its bound]
        for k, v in f.items():
            if not isinstance(v, dict) or "error" in v:
                continue
            if k == "f_american_lsm":
                line["f"]["american_lsm"] = {sz: [_num(d.get("ms_per_call"), 4), _num(d.get("frac"), 3)] for sz, d in (v.get("sizes") or {}).items()}
            else:
                line["f"][k[2:]] = [_num(v.get("avg_kernel_ms"), 4), _num(roof_frac(v), 3)]
    w = full.get("c2_f64_normals")
    if isinstance(w, dict) and "error" not in w:
        line["c2_f64_normals"] = {"value": _num(w.get("value"), 4), "kernel_ms": _num(w.get("avg_kernel_ms"), 4), "x_product_kernel": _num(w.get("slowdown_vs_product_kernel"), 3),
                                  "abs_diff_over_se": _num(w.get("abs_diff_over_se"), 3)}
    sp = full.get("c5_single_process")
    if isinstance(sp, dict) and "error" not in sp:
        line.setdefault("c5", {})["single_process"] = _entry(sp.get("value"), sp.get("ms_per_step"), None, n_gpus=sp.get("n_gpus"), paths_per_gpu=sp.get("paths_per_gpu"),
                                                             enqueue_us=sp.get("enqueue_us"))
    c5 = {}
    for k in ("c5_weak", "c5_strong", "c2_1m_per_gpu", "n1_basis", "pipelined"):
        v = full.get(k)
        if isinstance(v, dict) and "error" not in v:
            c5[k] = _entry(v.get("value"), v.get("ms_per_step"), roof_frac(v), paths_per_gpu=v.get("paths_per_gpu"), kernel_ms=v.get("avg_kernel_ms"))
    for k in ("c5_weak", "c5_strong"):
        if k in c5:
            line.setdefault("c5", {})[k[3:]] = c5.pop(k)
    line.update(c5)
    errs = []
    for k in ("c3_greeks", "c4_asian", "c5_weak", "c5_strong", "c2_1m_per_gpu", "n1_basis", "pipelined", "f_kernels", "c2_f64_normals", "c5_single_process"):
        if isinstance(full.get(k), dict) and "error" in full[k]:
            errs.append(f"{k}: {full[k]['error']}")
    errs += list(full.get("errors") or [])
    if errs:
        line["errors"] = [clip(e, 120) for e in errs[:4]]
    if detail_name:
        line["detail"] = detail_name
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_BUDGET:                      # belt and braces: shed the secondary entries, never the contract keys
        for k in ("pipelined", "f", "c2_f64_normals", "c4", "c3", "c5", "errors"):
            line.pop(k, None)
            if len(json.dumps(line, separators=(",", ":"))) <= LINE_BUDGET:
                break
    return line


def write_detail(full):
    """The full record next to this script (and wherever $OLMC_BENCH_DETAIL points); returns the name the line carries."""
    try:
        with open(DETAIL_FILE, "w") as f:
            json.dump(full, f, indent=1)
        return os.path.relpath(DETAIL_FILE, ROOT) if DETAIL_FILE.startswith(ROOT) else DETAIL_FILE
    except OSError as e:
        print(f"[bench] cannot write {DETAIL_FILE}: {e}", file=sys.stderr)
        return None


# ------------------------------------------------------------------------------------------------ launcher (N > 1, no torchrun around)
def launch_ranks(args):
    """`python3 bench.py --gpus N` on its own: this process never touches a GPU; it starts the N ranks as ONE child process
    tree (torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1), relays rank 0's JSON line and the exit code."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this pool (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "1")
    env["OLMC_BENCH_SPAWNED"] = "1"
    print(f"[bench] --gpus {args.gpus} without a launcher: starting {' '.join(cmd[1:8])} ...", file=sys.stderr)
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, timeout=1500)
    except subprocess.TimeoutExpired as e:
        sys.stderr.write(f"[bench] the ranks did not finish within {e.timeout:.0f} s\n")
        return 5
    line = None
    for raw in (r.stdout or b"").decode("utf-8", "replace").splitlines():
        raw = raw.strip()
        if raw.startswith("{") and '"metric"' in raw:
            line = raw
    if line is not None:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    if r.returncode == 0 and line is None:
        sys.stderr.write("[bench] the ranks exited 0 without printing a line\n")
        return 4
    return r.returncode


# ------------------------------------------------------------------------------------------------ main
def main():
    faulthandler.enable()             # a crash in native code (HIP, RCCL) leaves a Python traceback on stderr instead of nothing
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--streams", type=int, default=8, help="HIP streams of the secondary `pipelined` pass (>= 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 counter passes (use under a profiler)")
    ap.add_argument("--no-extras", action="store_true", help="skip pipelined / C3 / C4 / C5 (profiling runs)")
    ap.add_argument("--pmc-keep", default=None, help="directory to keep the rocprofv3 CSVs of the live passes in")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--single-process-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--min-seconds", type=float, default=1.5,
                    help="timed passes repeat until the plain ones alone cover this much wall time (as many instrumented ones ride along: the GPU "
                         "is busy for about twice this, which an outside utilisation sampler can see)")
    ap.add_argument("--paths-per-gpu", type=int, default=0,
                    help="default: 1,000,000 at --gpus 1 (BASELINE configs[1]), 8,000,000 at --gpus N > 1 (configs[4]'s per-GPU shard)")
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child()
    if args.single_process_child:
        return single_process_child(args)
    if args.steps < 1 or args.warmup < 0 or args.gpus < 1:
        raise SystemExit("--gpus and --steps must be >= 1 and --warmup >= 0")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    return worker(args)


def worker(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal of the N > 1 control flow on a ONE-GPU box (never the driver's mode): OLMC_BENCH_REHEARSAL=1 puts every
    # rank on device 0 and carries the collectives over gloo (RCCL refuses two ranks on one device).
    rehearsal = os.environ.get("OLMC_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    os.environ["OLMC_DEVICE"] = str(local_rank)
    use_dist = world > 1 or os.environ.get("OLMC_BENCH_FORCE_DIST") == "1"
    paths_per_gpu = args.paths_per_gpu or (PATHS_PER_GPU if world == 1 else C5_PATHS_PER_GPU)
    pmc_key = {PATHS_PER_GPU: "c2_european", C5_PATHS_PER_GPU: "c5_shard"}.get(paths_per_gpu)

    # Child-process work first, while this process has not touched the GPU: the live PMC passes, the CPU baseline's pool, and the
    # (waiting) child that will measure the single-process multi-GPU form.
    pmc, cpu = None, None
    sp_child = start_single_process_child(args, world) if (rank == 0 and not args.no_extras and not rehearsal) else None
    if world == 1 and rank == 0:
        if not args.no_pmc:
            pmc = collect_pmc(args.pmc_keep)
            if "error" in pmc:
                print(f"[bench] live PMC passes unavailable: {pmc['error']}", file=sys.stderr)
                pmc = committed_pmc() or pmc
        if not args.no_cpu_baseline:
            cpu = cpu_baseline()
    elif rank == 0 and not args.no_pmc:
        pmc = committed_pmc()           # N > 1 ranks run no profiler passes: counters of the committed run of this build, labelled

    # stdout carries ONE JSON line and nothing else: libraries that talk on fd 1 (RCCL's version banner, gloo's connection
    # notes) are sent to stderr until the line is printed
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    import optionslab_amd as ol
    from optionslab_amd import _hip, sharding

    torch.cuda.set_device(local_rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    # a CPU-side barrier (a NCCL barrier is a kernel that spins on every GPU until the last rank arrives: not what ranks should do
    # while rank 0's child measures the single-process form on the same devices)
    gloo_group = None
    if use_dist and world > 1:
        gloo_group = dist.new_group(backend="gloo")
    info = _hip.device_info()
    S, K, T, r, sigma, q = (WORK[k] for k in ("S", "K", "T", "r", "sigma", "q"))
    bs = ol.black_scholes(S, K, T, r, sigma, "call", q)
    main_stream = torch.cuda.Stream()
    torch.cuda.set_stream(main_stream)
    K_steps, W = args.steps, args.warmup
    dbuf = torch.zeros(3, dtype=torch.float64, device="cuda")

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        if rehearsal:
            torch.cuda.synchronize()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def make_step(n_global, sharded=None):
        """step(k) -> (price, std_error, n): ONE blocking pricing of n_global paths x 252 steps, result on the host.
        sharded=False prices all n_global paths on THIS rank without any collective (the one-GPU basis of a workload)."""
        if not (use_dist if sharded is None else sharded):
            pricer = ol.MonteCarloPricer(n_global, N_STEPS, SEED)

            def step(k):
                res = pricer.price(S, K, T, r, sigma, "call", seed=SEED + k, return_error=True)     # monte_carlo.py:108-152
                return res.price, res.std_error, res.n_paths
            return step
        lo, hi = sharding.shard_bounds(n_global, rank, world)

        def step(k):
            st = torch.cuda.current_stream()
            _hip.european_shard_dev(S, K, T, r, sigma, q, True, lo, hi - lo, N_STEPS, SEED + k, True, dbuf.data_ptr(), st.cuda_stream)
            if rehearsal:
                st.synchronize()        # gloo does not order itself behind the caller's stream; RCCL does
            dist.all_reduce(dbuf, op=dist.ReduceOp.SUM)                  # the ONE collective: 3 x fp64 over RCCL/xGMI
            if rehearsal:
                torch.cuda.synchronize()
            # hand-over to the host as in a blocking pricing: a one-wave kernel behind the all-reduce writes the triple into the
            # library's pinned buffer and raises the completion word the host polls (8 us -> 3 us against a D2H copy + stream sync)
            s, ss, n = _hip.fetch_dev(dbuf.data_ptr(), 3, st.cuda_stream)
            price, se = sharding.finalize(s, ss, int(n), r, T)
            return price, se, int(n)
        return step

    def timed_passes(step, n_global, steps, warm, min_total_s=1.5, max_passes=1000, seed0=0):
        """warm untimed steps, then passes of EXACTLY `steps` blocking steps, each between fences; max over ranks per pass.
        Passes ALTERNATE plain / instrumented: an instrumented pass is the same loop with a HIP event pair attached to
        every dispatch (the kernel's own begin / end timestamps).  Attaching the pair puts a marker packet in front of the
        kernel, which costs a blocking call several microseconds, so the events ride on every other pass instead of
        on all of them: `value` comes from the plain passes, the kernel duration from the instrumented ones, and both
        pass times are reported.  Passes repeat until the plain ones alone cover `min_total_s` (at least 3, at most `max_passes`
        of each kind).  Returns (plain pass times, instrumented pass times, per-call times of
        the plain passes on this rank, avg kernel seconds, launches timed, worst |price - BS| / se, prices of the first pass)."""
        worst, calls, first = 0.0, [], []

        def check(res):
            nonlocal worst
            price, se, n = res
            assert n == 2 * n_global, (n, n_global)
            worst = max(worst, abs(price - bs) / se)

        for k in range(warm):
            check(step(seed0 + 100_000 + k))
        _hip.profile_enable(True)       # pre-creates the event pool outside the timed region
        _hip.profile_reset()
        passes, inst_passes = [], []
        # every rank sees the same max-reduced pass times, so every rank stops after the same pass
        while len(inst_passes) < len(passes) or len(passes) < 3 or (sum(passes) < min_total_s and len(passes) < max_passes):
            instrumented = len(passes) > len(inst_passes)          # plain, instrumented, plain, ...
            _hip.profile_enable(instrumented)
            base = seed0 + (len(passes if not instrumented else inst_passes) * steps) % 1_000_000     # both kinds walk the same seeds
            results, stamps, pc = [], [], time.perf_counter
            fence()
            t0 = pc()
            for k in range(steps):             # the timed region holds the K blocking steps and a clock read each; checks come after
                results.append(step(base + k))
                stamps.append(pc())
            torch.cuda.synchronize()
            dt = max_over_ranks(pc() - t0)
            fence()
            for res in results:
                check(res)
            if not instrumented:
                calls.extend(b - a for a, b in zip([t0] + stamps[:-1], stamps))
            if not passes and not instrumented:
                first.extend(res[0] for res in results)
            (inst_passes if instrumented else passes).append(dt)
            if instrumented and len(inst_passes) % 16 == 0:
                _hip.kernel_time()      # drain the event pool as the passes go (4096 pairs)
        launches, kernel_ms = _hip.kernel_time()
        _hip.profile_enable(False)
        avg_kernel_s = kernel_ms / 1e3 / launches if launches else None
        return passes, inst_passes, calls, avg_kernel_s, launches, worst, first

    # Before anything is timed the device gets ~150 ms of the same work (untimed, `pre_warm_ms`): an idle MI355X needs tens
    # of milliseconds of load to reach its sustained clocks.  A FIXED count, not a clock: every rank enters the same
    # collectives the same number of times.
    n_global = paths_per_gpu * world
    step_main = make_step(n_global)
    pre_warm = max(64, PRE_WARM_PASSES * 32 * PATHS_PER_GPU // paths_per_gpu)
    t_pre = time.perf_counter()
    for k in range(pre_warm):
        step_main(500_000 + k)
    pre_warm_ms = (time.perf_counter() - t_pre) * 1e3

    ranks_seen = 1
    if use_dist:
        ones = torch.ones(1, dtype=torch.float64, device="cuda")
        if rehearsal:
            torch.cuda.synchronize()
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        ranks_seen = int(ones.item())

    passes, inst_passes, calls, avg_kernel_s, launches, worst, first_prices = timed_passes(step_main, n_global, K_steps, W,
                                                                                           min_total_s=args.min_seconds)     # THE measured passes
    assert worst <= 5.5, f"a step's price is {worst:.2f} sigma from Black-Scholes"     # max of up to ~50,000 draws of |N(0,1)|: P(> 5.5) ~ 2e-3
    pass_s = statistics.median(passes)
    path_steps = n_global * N_STEPS
    value = path_steps * K_steps / pass_s

    # calibration of the roofline: the shader clock held under this kernel's load and the per-class issue times, measured by the
    # INSTRUMENTED build (include/olmc_probe.h, tools/probe/libolmc_probe.so) -- the product library carries no measurement kernels
    clock = None
    try:
        from tools.probe import binding as probe
        for _ in range(3):
            clock = probe.clock_probe(PATHS_PER_GPU, N_STEPS, SEED)
    except Exception as e:
        clock = {"error": f"{type(e).__name__}: {e}"}
    clock_ghz = clock.get("ghz") if clock else None
    costs, mixes = None, load_isa_mix()
    try:
        costs = probe.issue_probe(8)         # ns per wave64 instruction per SIMD, per class, on this device, now
    except Exception as e:
        print(f"[bench] issue probes unavailable: {type(e).__name__}: {e}", file=sys.stderr)
    if "c5_shard" not in mixes and "c2_european" in mixes:
        mixes = dict(mixes, c5_shard=mixes["c2_european"])          # the same kernel, another launch shape

    out = None
    if rank == 0:
        local_paths = sharding.shard_bounds(n_global, 0, world)[1]
        roof = roofline_for(pmc, pmc_key, avg_kernel_s, N_STEPS, paths_per_gpu, costs, mixes, clock_ghz) if pmc_key else None
        if roof is None:
            roof = {"bound": "valu", "achieved": None, "peak": N_SIMD * PEAK_GHZ, "unit": "G VALU issue-cycles/s (peak = 1024 SIMDs x 2.4 GHz)", "frac": None,
                    "traffic": None, "avg_kernel_ms": avg_kernel_s * 1e3 if avg_kernel_s else None, "clock_ghz_under_load": clock_ghz,
                    "why_null": (pmc or {}).get("error", "no PMC counters for this launch size (only 1M and 8M paths per launch are profiled)")}
        roof.update({
            "kernel": "european_path_kernel<1,true,kReduce,false>", "launches_timed": launches,
            "pmc_source": (pmc or {}).get("source"), "issue_costs_ns": costs,
            "note": "VALU-issue bound (SURVEY 8d: neither HBM nor MFMA). frac = issue cycles the kernel's VALU instruction stream needs (live "
                    "SQ_INSTS_VALU / SQ_WAVES, the step loop's static mix from the ISA, 2 / 4 / 8 issue passes per class) / (1024 SIMDs x kernel "
                    "time x 2.4 GHz): <= 1 by construction. Kernel time = HIP events attached to the dispatches of the instrumented passes (the "
                    "kernel's own begin / end timestamps). issue_costs_ns = the per-class issue times measured in isolation in this run; "
                    "frac_vs_isolated_rates and frac_valu_active_pmc are secondary readings, neither bounded by 1 (see bench.py roofline_for).",
            "lane_op_model": {"lane_ops_per_path_step": LANE_OPS_PER_PATH_STEP, "peak_tlaneops": PEAK_TLANEOPS,
                              "achieved_tlaneops": (local_paths * N_STEPS * LANE_OPS_PER_PATH_STEP / avg_kernel_s / 1e12) if avg_kernel_s else None,
                              "frac": (local_paths * N_STEPS * LANE_OPS_PER_PATH_STEP / avg_kernel_s / 1e12 / PEAK_TLANEOPS) if avg_kernel_s else None,
                              "what": "SURVEY 8(d)'s algorithmic count (32 lane-ops per path-step) over 256 CU x 4 SIMD-32 x 2.4 GHz = 78.6 T "
                                      "(the guide's 157.3 TF fp32 / 2; SURVEY's own 39.3 T assumed SIMD-16). A MODEL, not a bound: the kernel "
                                      "issues 14.5 instructions per path-step, so this ratio can exceed 1"}})
        which = "configs[1]" if (world == 1 and paths_per_gpu == PATHS_PER_GPU) else ("configs[4]" if paths_per_gpu == C5_PATHS_PER_GPU else "non-default size")
        out = {
            "metric": "MC path-steps/sec (1M paths × 252 steps Euro call); price vs BS |err|/σ",      # BASELINE.json, verbatim
            "value": value, "unit": "path-steps/s", "n_gpus": world, "steps": K_steps, "warmup": W,
            "ms_per_step": pass_s / K_steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 normals / f64 prices", "data": "synthetic",
            "config": {"workload": f"BASELINE {which}: European call S0=100 K=100 sigma=0.2 r=0.05 T=1, {paths_per_gpu:,} paths x 252 steps per GPU, "
                                   "antithetic, on-device reduction; step = one blocking price(return_error=True)",
                       "paths_per_gpu": paths_per_gpu, "n_steps": N_STEPS, "global_paths": n_global,
                       "parallelism": f"path-sharded x{world}" + (", 1 RCCL all-reduce of (sum,sumsq,n) per step" if world > 1 else "")
                                      + (" [REHEARSAL: all ranks on one GPU, gloo]" if rehearsal else "")},
            "passes": {"n": len(passes), "seconds": passes, "what": f"each = exactly {K_steps} blocking steps between fences, max over ranks; "
                                                                    "value and ms_per_step use the median pass"},
            "instrumented_passes": {"n": len(inst_passes), "ms_per_step": statistics.median(inst_passes) / K_steps * 1e3,
                                    "what": "the same loop, same seeds, interleaved with the plain passes, with a HIP event pair attached to every "
                                            "dispatch -> roofline.avg_kernel_ms. The pair's start marker is an extra packet in front of the kernel: "
                                            "the difference to ms_per_step is its price, which is why `value` is not taken from these passes"},
            "per_call": {"median_ms": statistics.median(calls) * 1e3, "min_ms": min(calls) * 1e3, "p90_ms": sorted(calls)[int(0.9 * (len(calls) - 1))] * 1e3,
                         "n": len(calls), "what": "wall of the individual blocking calls of the timed passes (rank 0)"},
            "pre_warm_ms": pre_warm_ms, "pre_warm_pricings": pre_warm,
            "ranks_seen": ranks_seen,
            "payoff_samples_per_s": 2 * n_global * K_steps / pass_s,      # the antithetic mirror doubles the payoff samples, not the path-steps
            "accuracy": {"bs_price": bs, "max_abs_err_over_sigma": worst, "payoffs_per_step": 2 * n_global, "steps_checked": len(calls) + len(inst_passes) * K_steps + W},
            "roofline": roof,
            "clock": clock,
            "device": info,
            "isa_mix": {k: mixes.get(k) for k in set(PMC_KERNELS.values()) if k in mixes},
            "issue_passes": ISSUE_PASSES,
        }
        if cpu:
            out["cpu_baseline"] = cpu
            out["gpu_over_cpu"] = value / cpu["value"]

    # ---- everything below is secondary: it must never cost the line.  A watchdog prints what exists if a section hangs --
    # and then exits NON-ZERO, naming the section, so a stuck GPU section is not reported as success.
    printed = threading.Event()
    current = {"section": None}

    def emit():
        if rank == 0 and not printed.is_set():
            printed.set()
            sys.stdout.flush()
            detail = write_detail(out)
            os.write(json_fd, (json.dumps(compact_line(out, detail), separators=(",", ":")) + "\n").encode())

    def watchdog_fire():
        if sp_child is not None and sp_child.poll() is None:
            sp_child.kill()
        if rank == 0 and out is not None:
            out.setdefault("errors", []).append(f"section `{current['section']}` did not finish within its time limit; line printed by the watchdog, exit 3")
        emit()
        os._exit(3)

    dog = threading.Timer(600.0 if world == 1 else 300.0, watchdog_fire)
    dog.daemon = True
    dog.start()

    def section(name, fn):
        current["section"] = name
        try:
            res = fn()
        except Exception as e:          # recorded, not raised
            res = {"error": f"{type(e).__name__}: {e}"}
        if rank == 0 and res is not None:
            out[name] = res

    def side(n_glob, steps, what, sharded=None, key=None, min_total_s=0.1):
        """A secondary workload through the same timed_passes: {value, ms_per_step, avg_kernel_ms, roofline, ...}."""
        step = make_step(n_glob, sharded)
        ps, _ips, cs, ks, ln, wst, _first = timed_passes(step, n_glob, steps, 2, min_total_s=min_total_s, max_passes=12, seed0=7_000_000)
        med = statistics.median(ps)
        per_gpu = n_glob // world if (use_dist if sharded is None else sharded) else n_glob
        d = {"value": n_glob * N_STEPS * steps / med, "unit": "path-steps/s", "ms_per_step": med / steps * 1e3, "steps": steps, "passes": len(ps),
             "global_paths": n_glob, "paths_per_gpu": per_gpu, "n_gpus": world, "avg_kernel_ms": ks * 1e3 if ks else None,
             "max_abs_err_over_sigma": wst, "dtype": "f32 normals / f64 prices", "workload": what}
        if rank == 0 and key:
            d["roofline"] = roofline_for(pmc, key, ks, N_STEPS, per_gpu, costs, mixes, clock_ghz)
        return d

    if not args.no_extras:
        if world > 1:
            # the same 8M-path workload on ONE GPU with no collective, measured by every rank in this very run: the N = 1 basis the
            # scaling efficiency of `value` is computable from (value / (N x n1_basis.value)) out of one record
            section("n1_basis", lambda: side(paths_per_gpu, 10, f"one-GPU basis of the headline workload: European call, {paths_per_gpu:,} paths x 252 steps on this "
                                             "rank alone, blocking price(), no collective; max over ranks", sharded=False, key=pmc_key))
            section("c2_1m_per_gpu", lambda: side(PATHS_PER_GPU * world, 50, f"configs[1] sharded: {PATHS_PER_GPU:,} paths x 252 steps per GPU x {world} GPUs, blocking "
                                                  "sharded price(), one all-reduce of the triple per step", key="c2_european"))
        else:
            # BASELINE configs[4] on one GPU: its per-GPU shard (8M paths), and all 64M paths on this one device
            section("c5_weak", lambda: side(C5_PATHS_PER_GPU, 10, f"configs[4] weak: European call, {C5_PATHS_PER_GPU:,} paths x 252 steps per GPU x {world} GPU(s), "
                                            "blocking price()", key="c5_shard"))
        section("c5_strong", lambda: side(C5_TOTAL, 3, f"configs[4] strong: European call, {C5_TOTAL:,} paths x 252 steps in total split over {world} GPU(s), "
                                          "blocking sharded price(), one all-reduce of the triple per step", min_total_s=0.05))
        section("pipelined", lambda: pipelined(args, torch, dist, _hip, sharding, use_dist, rehearsal, world, rank, main_stream, fence, max_over_ranks,
                                               n_global, K_steps, W, bs, first_prices))
        if world == 1 and not use_dist:
            section("c3_greeks", lambda: c3_greeks(ol, _hip, pmc, costs, mixes, clock_ghz))
            section("c4_asian", lambda: c4_asian(ol, _hip, pmc, costs, mixes, clock_ghz, cpu))
            section("f_kernels", lambda: f_kernels(ol, _hip, pmc, costs, mixes, clock_ghz))
            section("c2_f64_normals", lambda: c2_f64_normals(ol, _hip, bs))
        # the torch-free form north_star describes: ONE process, all N devices, olmc_multi_gpu_european (its own streams, one grouped RCCL
        # all-reduce, no torch.distributed).  Measured by a fresh child of rank 0 while every rank of THIS job idles on a CPU barrier.
        section("c5_single_process", lambda: c5_single_process(sp_child, dist, use_dist, rehearsal, world, rank, gloo_group))
    dog.cancel()
    emit()
    if sp_child is not None and sp_child.poll() is None:
        sp_child.kill()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ secondary sections
def pipelined(args, torch, dist, _hip, sharding, use_dist, rehearsal, world, rank, main_stream, fence, max_over_ranks, n_global, K_steps, W, bs,
              blocking_prices):
    """The same K pricings as independent requests dealt round-robin over --streams HIP streams: the ~15 us tail of one
    launch (its last lone workgroup + the reduction chain + the host round trip) hides under the head of the next.
    Triples land in per-step device slots; the timed region ends with their D2H copy and a full synchronise."""
    S, K, T, r, sigma, q = (WORK[k] for k in ("S", "K", "T", "r", "sigma", "q"))
    lo, hi = sharding.shard_bounds(n_global, rank, world)
    streams = [torch.cuda.Stream() for _ in range(max(1, args.streams))]
    slots = torch.zeros((max(K_steps, W, 1), 3), dtype=torch.float64, device="cuda")
    host_slots = torch.zeros(slots.shape, dtype=torch.float64).pin_memory()

    def run_pass(steps, seed0):
        def enq(k):
            st = streams[k % len(streams)]
            with torch.cuda.stream(st):
                _hip.european_shard_dev(S, K, T, r, sigma, q, True, lo, hi - lo, N_STEPS, seed0 + k, True, slots[k].data_ptr(), st.cuda_stream)
                if use_dist:
                    if rehearsal:
                        st.synchronize()
                    return dist.all_reduce(slots[k], op=dist.ReduceOp.SUM, async_op=True)
            return None

        fence()
        t0 = time.perf_counter()
        pend = [enq(k) for k in range(steps)]
        for w in pend:
            if w is not None:
                w.wait()
        for st in streams:
            main_stream.wait_stream(st)
        host_slots[:steps].copy_(slots[:steps], non_blocking=True)
        torch.cuda.synchronize()
        dt = max_over_ranks(time.perf_counter() - t0)
        fence()
        return dt, host_slots[:steps].clone()

    if W:
        run_pass(W, SEED + 1000)
    times, res = [], None
    n_pass = None
    while n_pass is None or len(times) < n_pass:
        dt, res = run_pass(K_steps, SEED)        # the seeds of the blocking pass's first K steps
        times.append(dt)
        if n_pass is None:
            n_pass = max(3, min(25, int(math.ceil(0.05 / max(dt, 1e-9)))))
    worst, drift = 0.0, 0.0
    for k, row in enumerate(res.tolist()):
        price, se = sharding.finalize(row[0], row[1], int(row[2]), r, T)
        assert int(row[2]) == 2 * n_global, row
        worst = max(worst, abs(price - bs) / se)
        if k < len(blocking_prices):            # same seeds as the blocking pass: the overlapped launches must reproduce it
            drift = max(drift, abs(price - blocking_prices[k]) / abs(blocking_prices[k]))
    med = statistics.median(times)
    return {"value": n_global * N_STEPS * K_steps / med, "unit": "path-steps/s", "ms_per_step": med / K_steps * 1e3, "streams": len(streams),
            "passes": len(times), "max_abs_err_over_sigma": worst, "max_rel_diff_vs_blocking": drift, "consistent_with_blocking": drift <= 1e-12,
            "what": "K independent pricings in flight over several HIP streams (results stay on the device until one D2H at the end); "
                    "NOT the blocking call SURVEY 8(d) defines the metric on -- `value` is"}


def _timed_calls(_hip, fn, reps, warm=3):
    """(median wall of a call, kernel seconds per launch, launches per call).  Two loops: the wall is taken with the dispatch events
    OFF (an event pair puts a marker packet in front of the kernel: +9 us on a blocking call), the kernel time with them on."""
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    _hip.profile_enable(True)
    _hip.profile_reset()
    for _ in range(reps):
        fn()
    launches, kernel_ms = _hip.kernel_time()
    _hip.profile_enable(False)
    return statistics.median(ts), (kernel_ms / 1e3 / launches if launches else None), launches // reps


def c3_greeks(ol, _hip, pmc, costs, mixes, clock_ghz):
    """BASELINE configs[2]: finite-difference Greeks at 1M paths x 252 steps, common Philox key.  fused = the 8 / 14 bumped
    contracts of unified_greeks.py:295-358 priced on the SAME normals in ONE launch; literal = the 8 price() calls."""
    p = ol.MonteCarloPricer(PATHS_PER_GPU, N_STEPS, SEED)
    ps = PATHS_PER_GPU * N_STEPS
    out = {"dtype": "f32 normals / f64 prices and differences"}
    for key, second, pk in (("fused_8", False, "c3_fused8"), ("fused_14", True, "c3_fused14")):
        med, ks, per = _timed_calls(_hip, lambda: p.greeks(*ATM, "call", include_second_order=second), 20)
        g = p.greeks(*ATM, "call", include_second_order=second)
        out[key] = {"ms_per_call": med * 1e3, "path_steps_per_s": ps / med, "contract_path_steps_per_s": (14 if second else 8) * ps / med,
                    "avg_kernel_ms": ks * 1e3 if ks else None, "launches_per_call": per, "roofline": roofline_for(pmc, pk, ks, N_STEPS, PATHS_PER_GPU, costs, mixes, clock_ghz),
                    "delta": g["delta"], "gamma": g["gamma"], "vega": g["vega"], "theta": g["theta"], "rho": g["rho"]}
    med, ks, per = _timed_calls(_hip, lambda: ol.compute_greeks_unified(p, *ATM, "call", include_second_order=False, fused=False), 10)
    out["literal_8"] = {"ms_per_call": med * 1e3, "path_steps_per_s": 8 * ps / med, "avg_kernel_ms": ks * 1e3 if ks else None, "launches_per_call": per,
                        "what": "8 blocking price() calls, path-steps counted 8x (SURVEY 8d)"}
    out["workload"] = "configs[2]: FD Greeks (delta gamma vega theta rho), 2-sided bump-and-reprice, 1,000,000 paths x 252 steps, antithetic, common key"
    return out


def c4_asian(ol, _hip, pmc, costs, mixes, clock_ghz, cpu=None):
    """BASELINE configs[3]: arithmetic Asian call, 1M paths x 1024 dates.  fp64 = the reference's arithmetic (the default);
    fp32 = the opt-in fast kernel, reported beside it, never instead of it."""
    a = ol.AsianOption(*ATM, seed=SEED)
    ps = PATHS_PER_GPU * ASIAN_STEPS
    out = {"workload": "configs[3]: arithmetic Asian call, 1,000,000 paths x 1024 monitoring dates (t = 1..M), running sum in registers"}
    for key, precision, anti, pk in (("fp64", "fp64", False, "c4_asian_fp64"), ("fp64_antithetic", "fp64", True, "c4_asian_fp64_antithetic"),
                                     ("fp32_fast", "fp32", False, "c4_asian_fp32"), ("fp32_fast_antithetic", "fp32", True, "c4_asian_fp32_antithetic")):
        fn = lambda: a.price(PATHS_PER_GPU, ASIAN_STEPS, "arithmetic", "call", antithetic=anti, return_error=True, precision=precision)
        med, ks, per = _timed_calls(_hip, fn, 16, warm=6)
        price, se = fn()
        out[key] = {"ms_per_call": med * 1e3, "path_steps_per_s": ps / med, "avg_kernel_ms": ks * 1e3 if ks else None, "price": float(price), "std_error": se,
                    "dtype": "f32 normals / f64 cumulative log-return, f64 exp per date, f64 sums" if precision == "fp64"
                             else "f32 normals / f32 exponent + v_exp_f32 per date inside groups of 16 dates, f64 across groups",
                    "roofline": roofline_for(pmc, pk, ks, ASIAN_STEPS, PATHS_PER_GPU, costs, mixes, clock_ghz)}
    # finite-difference Greeks over the same option (compute_greeks_unified through ExoticAdapter, unified_greeks.py:177-227): the 8 / 14
    # evaluations as six path recursions in ONE launch (olmc_asian_greeks_fd) against the 8 / 14 launches of the kernel above
    ad = ol.ExoticAdapter(ol.AsianOption(*ATM, seed=SEED), n_paths=PATHS_PER_GPU, n_steps=ASIAN_STEPS)
    greeks = {}
    for key, second, fused, reps in (("fused_8", False, True, 8), ("fused_14", True, True, 8), ("literal_8", False, False, 3), ("literal_14", True, False, 3)):
        med, ks, per = _timed_calls(_hip, lambda: ol.compute_greeks_unified(ad, *ATM, "call", include_second_order=second, fused=fused), reps, warm=2)
        greeks[key] = {"ms": med * 1e3, "launches_per_call": per, "kernel_ms_per_launch": ks * 1e3 if ks else None}
        if key == "fused_14":
            greeks[key]["roofline"] = roofline_for(pmc, "c4_asian_greeks14", ks, ASIAN_STEPS, PATHS_PER_GPU, costs, mixes, clock_ghz)
    greeks["speedup_14"] = greeks["literal_14"]["ms"] / greeks["fused_14"]["ms"]
    greeks["speedup_8"] = greeks["literal_8"]["ms"] / greeks["fused_8"]["ms"]
    # the geometric average: its six recursions share the whole date loop (asian_geometric_greeks_kernel) -- detail file only
    geo = ol.ExoticAdapter(ol.AsianOption(*ATM, seed=SEED), n_paths=PATHS_PER_GPU, n_steps=ASIAN_STEPS, avg_type="geometric")
    for key, fused, reps in (("geometric_fused_14", True, 8), ("geometric_literal_14", False, 3)):
        med, ks, per = _timed_calls(_hip, lambda: ol.compute_greeks_unified(geo, *ATM, "call", include_second_order=True, fused=fused), reps, warm=2)
        greeks[key] = {"ms": med * 1e3, "launches_per_call": per, "kernel_ms_per_launch": ks * 1e3 if ks else None}
    greeks["geometric_speedup_14"] = greeks["geometric_literal_14"]["ms"] / greeks["geometric_fused_14"]["ms"]
    out["greeks"] = greeks
    out["headline"] = "fp64"
    if cpu and isinstance(cpu.get("c4_asian"), dict) and cpu["c4_asian"].get("value"):
        out["cpu_baseline"] = cpu["c4_asian"]
        out["gpu_over_cpu"] = out["fp64"]["path_steps_per_s"] / cpu["c4_asian"]["value"]
    return out


def f_kernels(ol, _hip, pmc, costs, mixes, clock_ghz):
    """SURVEY 8(f) kernels: one workload each, the kernel's own duration (dispatch events), the issue-cycle fraction
    of the same model as the headline kernel's -- live SQ_INSTS_VALU of this kernel, the static mix of its hot loop, 2 / 4 / 8 passes
    per class -- and, for the American option's chain of per-date launches, an HBM + dependent-launch bound."""
    out = {}
    for key, (fn, what, n_units, n_steps) in f_workloads(ol, _hip).items():
        try:
            med, ks, per = _timed_calls(_hip, fn, 12, warm=4)
            roof = roofline_for(pmc, key, ks, n_steps, n_units, costs, mixes, clock_ghz)
            out[key] = {"ms_per_call": med * 1e3, "avg_kernel_ms": ks * 1e3 if ks else None, "launches_per_call": per, "workload": what,
                        "roofline": roof, "vgprs": (mixes.get(key) or {}).get("vgprs")}
        except Exception as e:
            out[key] = {"error": f"{type(e).__name__}: {e}"}
    # American LSM: 1 path kernel + one launch per exercise date, each dependent on the one before through five coefficients
    try:
        from tools.probe import binding as probe
        gap_us = probe.launch_gap_us(2000)
        chain = {}
        for n, m in ((50_000, 50), (1_000_000, 50)):
            fn = lambda: _hip.american_lsm(*ATM, 0.0, False, n, m, 3, SEED)
            med, ks, per = _timed_calls(_hip, fn, 12, warm=4)
            bytes_dates = 32.0 * n * m                   # per date and path: two prices + the cash flow read, the cash flow written
            bytes_paths = 8.0 * n * (m + 1)              # the path matrix, written once
            bound_s = (m + 1) * gap_us * 1e-6 + (bytes_dates + bytes_paths) / 8e12
            chain[f"{n}x{m}"] = {"ms_per_call": med * 1e3, "call_kernel_ms": ks * 1e3 if ks else None, "launches": m + 1, "bound_ms": bound_s * 1e3,
                                 "frac": bound_s / med, "algorithmic_bytes": bytes_dates + bytes_paths,
                                 "hbm_gbps_algorithmic": (bytes_dates + bytes_paths) / med / 1e9}
        out["f_american_lsm"] = {"bound": "hbm + dependent-launch chain", "dependent_launch_gap_us": gap_us, "sizes": chain,
                                 "what": "bound = (M + 1) launches x the measured gap between two dependent EMPTY kernels on one stream + (32 B per "
                                         "path and date + the 8 B path matrix) / 8 TB/s; frac = bound / blocking call"}
    except Exception as e:
        out["f_american_lsm"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def c2_f64_normals(ol, _hip, bs):
    """What the reference's own width would cost: the European call at 1M x 252 with fp64 normals -- same Philox
    stream, two 53-bit uniforms per block, library-precision fp64 Box-Muller, fp64 sum -- by the INSTRUMENTED build's
    european_f64_normals_kernel.  NOT a product path: a labelled number beside `dtype`: "f32 normals"."""
    from tools.probe import binding as probe
    fn = lambda: probe.european_f64_normals(*ATM, 0.0, True, PATHS_PER_GPU, N_STEPS, SEED)
    med, ks, per = _timed_calls(probe.hip, fn, 10, warm=3)
    f64 = fn()
    f32 = _hip.european(*ATM, 0.0, True, PATHS_PER_GPU, N_STEPS, SEED, True)
    ks32 = _timed_calls(_hip, lambda: _hip.european(*ATM, 0.0, True, PATHS_PER_GPU, N_STEPS, SEED, True), 10, warm=3)[1]
    return {"value": PATHS_PER_GPU * N_STEPS / med, "unit": "path-steps/s", "ms_per_call": med * 1e3, "avg_kernel_ms": ks * 1e3 if ks else None,
            "product_kernel_ms": ks32 * 1e3 if ks32 else None, "slowdown_vs_product_kernel": (ks / ks32) if ks and ks32 else None,
            "price": f64.price, "std_error": f64.std_error, "product_price": f32.price, "product_std_error": f32.std_error,
            "abs_diff_over_se": abs(f64.price - f32.price) / math.hypot(f64.std_error, f32.std_error),
            "err_over_se_vs_bs": abs(f64.price - bs) / f64.std_error, "product_err_over_se_vs_bs": abs(f32.price - bs) / f32.std_error,
            "dtype": "f64 normals (2 per Philox block, 53-bit uniforms, libm log / sincospi) / f64 prices",
            "what": "instrumented build only (include/olmc_probe.h: olmc_european_f64_normals); the two prices use different normals of the same "
                    "counter stream, so they differ by sampling noise: abs_diff_over_se is that difference in units of its own standard error"}


def single_process_child(args):
    """`bench.py --single-process-child --gpus N`: ONE process prices with all N devices through olmc_multi_gpu_european -- no torch,
    no torch.distributed -- and prints one JSON object.  Started by rank 0 of the benchmark while the ranks idle."""
    if sys.stdin.readline().strip() != "go":          # started before the parent touched a GPU; waits here, GPU untouched, until told
        return 0
    import optionslab_amd as ol
    from optionslab_amd import _hip
    n_gpus = args.gpus
    per_gpu = args.paths_per_gpu or (PATHS_PER_GPU if n_gpus == 1 else C5_PATHS_PER_GPU)
    n_global = per_gpu * n_gpus
    S, K, T, r, sigma, q = (WORK[k] for k in ("S", "K", "T", "r", "sigma", "q"))
    bs = ol.black_scholes(S, K, T, r, sigma, "call", q)
    step = lambda k: _hip.multi_gpu_european(S, K, T, r, sigma, q, True, n_global, N_STEPS, SEED + k, True, n_gpus)
    # N > 1 has never run on real multi-GPU hardware, so the child reports in STAGES, a JSON line after each (the parent takes the
    # last one, also from a child it had to kill): first the serial form of the launch phase (round 4's: the calling thread queues the
    # ranks one after the other), then the launcher-thread form (the default), then the other payloads.  With one rank the two
    # forms are the same code.
    out = {"n_gpus": n_gpus, "paths_per_gpu": per_gpu, "global_paths": n_global}
    if n_gpus > 1:
        try:
            _hip.tune(_hip.TUNE_MULTI_LAUNCH, -1)
            t0 = time.perf_counter()
            step(2000)
            serial_setup_s = time.perf_counter() - t0
            for k in range(10):
                step(2001 + k)
            sp, ssp = [], []
            for p_ in range(5):
                t0 = time.perf_counter()
                for k in range(args.steps):
                    step(3000 + p_ * args.steps + k)
                sp.append(time.perf_counter() - t0)
                ssp.append(_hip.multi_gpu_spans())
            out["serial_launch_form"] = {"ms_per_step": statistics.median(sp) / args.steps * 1e3, "value": n_global * N_STEPS * args.steps / statistics.median(sp),
                                         "first_call_s": serial_setup_s, "spans_us": {k: round(statistics.median(x[k] for x in ssp), 2) for k in ssp[0]}}
        except Exception as e:
            out["serial_launch_form"] = {"error": f"{type(e).__name__}: {e}"}
        finally:
            _hip.tune(_hip.TUNE_MULTI_LAUNCH, 0)
        print(json.dumps({**out, "partial": "stage 1 of 3: the launcher-thread form had not reported when this line was taken"}), flush=True)
    try:                                                # the threaded form has never run on more than one real GPU: if it fails, the serial form below still reports
        t0 = time.perf_counter()
        first = step(0)
        setup_s = time.perf_counter() - t0                  # contexts, rank streams, ncclCommInitAll
        for k in range(max(args.warmup, 20)):
            step(1000 + k)
        passes, worst, spans = [], 0.0, []
        while len(passes) < 3 or (sum(passes) < 0.5 and len(passes) < 200):
            t0 = time.perf_counter()
            res = [step(len(passes) * args.steps + k) for k in range(args.steps)]
            passes.append(time.perf_counter() - t0)
            spans.append(_hip.multi_gpu_spans())              # host spans of the pass's last call (olmc_multi_gpu_spans)
            for st in res:
                assert st.n == 2 * n_global
                worst = max(worst, abs(st.price - bs) / st.std_error)
        med = statistics.median(passes)
        span = {k: round(statistics.median(sp[k] for sp in spans), 2) for k in spans[0]}
        out.update({"value": n_global * N_STEPS * args.steps / med, "unit": "path-steps/s", "ms_per_step": med / args.steps * 1e3, "n_gpus": n_gpus, "paths_per_gpu": per_gpu,
               "global_paths": n_global, "steps": args.steps, "passes": len(passes), "max_abs_err_over_sigma": worst, "first_call_s": setup_s, "price": first.price,
               "enqueue_us": span["launch_us"], "spans_us": span,
               "what": "olmc_multi_gpu_european: one process, a launcher thread and a stream per device (every rank's kernel queued at once), ONE grouped "
                       "RCCL all-reduce of (sum, sumsq, n) per blocking pricing, result by rank 0's polled completion word; no torch.  enqueue_us = host "
                       "time from posting the launch to the last rank's kernel being queued"})
    except Exception as e:
        out.setdefault("errors", []).append(f"launcher-thread form: {type(e).__name__}: {e}")
    if n_gpus > 1:
        print(json.dumps({**out, "partial": "stage 2 of 3: the Greeks / control-variate payloads had not reported when this line was taken"}), flush=True)
    # the other two payloads of the same engine (count 33 and 6), at 1M paths per GPU
    try:
        g = lambda: _hip.multi_gpu_greeks_fd(S, K, T, r, sigma, q, True, PATHS_PER_GPU * n_gpus, N_STEPS, SEED, True, n_gpus, want_evals=False)
        c = lambda: _hip.multi_gpu_european_cv(S, K, T, r, sigma, q, True, PATHS_PER_GPU * n_gpus, N_STEPS, SEED, True, n_gpus)
        for name, fn in (("greeks14_1m_per_gpu", g), ("control_variate_1m_per_gpu", c)):
            for _ in range(5):
                fn()
            ts = []
            for _ in range(20):
                t0 = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t0)
            out[name] = {"ms_per_call": statistics.median(ts) * 1e3, "path_steps_per_s": PATHS_PER_GPU * n_gpus * N_STEPS / statistics.median(ts)}
    except Exception as e:
        out.setdefault("errors", []).append(f"{type(e).__name__}: {e}")
    print(json.dumps(out), flush=True)
    _hip.shutdown()
    return 0


def start_single_process_child(args, world):
    """Rank 0, BEFORE it touches a GPU (no process that has initialised HIP forks here): `bench.py --single-process-child --gpus N`
    as a child that waits on its stdin, GPU untouched, until c5_single_process tells it to go."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "OLMC_DEVICE", "MASTER_ADDR", "MASTER_PORT",
                                                              "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE", "TORCHELASTIC_RUN_ID", "OLMC_BENCH_DETAIL")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.abspath(__file__), "--single-process-child", "--gpus", str(world), "--steps", str(min(args.steps, 20)),
           "--warmup", str(args.warmup)] + (["--paths-per-gpu", str(args.paths_per_gpu)] if args.paths_per_gpu else [])
    try:
        return subprocess.Popen(cmd, env=env, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    except OSError as e:
        print(f"[bench] cannot start the single-process child: {e}", file=sys.stderr)
        return None


def c5_single_process(child, dist, use_dist, rehearsal, world, rank, gloo_group):
    """The child rank 0 started at the very beginning (a fresh process: its own HIP contexts on all N devices) measures between two
    CPU barriers; the other ranks wait there with idle GPUs.  In a rehearsal (all ranks on one GPU) the child is not run."""
    def cpu_barrier():
        if use_dist and world > 1:
            dist.barrier(group=None if rehearsal else gloo_group)
    if rehearsal:
        return {"error": "rehearsal: every rank sits on one GPU, the single-process form needs N devices"}
    cpu_barrier()
    res = None
    if rank == 0:
        if child is None:
            res = {"error": "the child was not started"}
        else:
            try:
                try:
                    out, err = child.communicate("go\n", timeout=240)
                    timed_out = False
                except subprocess.TimeoutExpired:
                    child.kill()                                    # this child, by its handle
                    out, err = child.communicate()                  # what it had printed: the child reports in stages
                    timed_out = True
                line = next((l for l in reversed(out.splitlines()) if l.startswith("{")), None)
                if line and (timed_out or child.returncode == 0):
                    res = json.loads(line)
                    if timed_out:
                        res.setdefault("errors", []).append("the child did not finish within 240 s and was killed; " + str(res.get("partial", "")))
                else:
                    res = {"error": f"child rc {child.returncode}{' (killed after 240 s)' if timed_out else ''}: {(err or out)[-300:]}"}
            except (OSError, ValueError) as e:
                child.kill()
                res = {"error": f"{type(e).__name__}: {e}"}
    cpu_barrier()
    return res


if __name__ == "__main__":
    main()
