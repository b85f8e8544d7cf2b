#!/usr/bin/env python3
"""The (n_paths x n_steps) grid north_star names: blocking MonteCarloPricer.price(return_error=True) on one GPU at
N in {1e4, 1e5, 1e6, 8e6, 6.4e7} x M in {1, 16, 64, 252, 1024} (the UI's range, streamlit_app/pages/1_MonteCarlo_Basic.py:111-126, is
10k-200k paths x 1-100 steps; BASELINE's configs sit at 1M / 8M / 64M x 252).  Per cell: wall per call, path-steps/s, kernel
microseconds (the dispatch's own timestamps), and the issue-cycle roofline fraction bench.py defines -- with SQ_INSTS_VALU of THAT
launch shape from a live rocprofv3 PMC pass of this script (`--pmc-child`), because the launch shape changes over the grid: split
workgroups at 1M, grid-striding short-path launches at M <= 128, sub-chip launches at 10k.  For cells below half of the roofline
the row says which floor the call sits on.

    python tools/measure_grid.py --out profiles/r03_grid.jsonl          (GPU box)
"""
import argparse
import csv
import glob
import importlib.util
import json
import os
import shutil
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PATHS = (10_000, 100_000, 1_000_000, 8_000_000, 64_000_000)
STEPS = (1, 16, 64, 252, 1024)
ATM = (100.0, 100.0, 1.0, 0.05, 0.2)
SEED = 42
PMC_LAUNCHES = 2


def cells():
    return [(n, m) for n in PATHS for m in STEPS]


def load_bench():
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec = importlib.util.spec_from_file_location("olmc_bench_for_grid", os.path.join(ROOT, "bench.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def pmc_child():
    """PMC_LAUNCHES blocking pricings per cell, cells in the fixed order of cells(): the parent maps dispatches back by order."""
    import optionslab_amd as ol
    for n, m in cells():
        p = ol.MonteCarloPricer(n, m, SEED)
        for i in range(PMC_LAUNCHES):
            p.price(*ATM, "call", seed=SEED + i, return_error=True)
    print("grid-pmc-child done", flush=True)


def collect_pmc():
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    base = tempfile.mkdtemp(prefix="olmc_grid_pmc_", dir="/tmp")
    cmd = [exe, "--kernel-trace", "--pmc", "SQ_INSTS_VALU", "SQ_WAVES", "--output-format", "csv", "-d", base, "--",
           sys.executable, os.path.abspath(__file__), "--pmc-child"]
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=300)
    except (OSError, subprocess.TimeoutExpired) as e:
        return {"error": f"{type(e).__name__}: {e}"}
    if r.returncode != 0 or "grid-pmc-child done" not in r.stdout:
        return {"error": f"rc {r.returncode}: {(r.stderr or r.stdout)[-400:]}"}
    rows = {}
    for f in glob.glob(os.path.join(base, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if "european_path_kernel<1, true, 0" in row["Kernel_Name"]:
                    d = rows.setdefault(int(row["Dispatch_Id"]), {"kernel": row["Kernel_Name"], "grid_threads": int(float(row["Grid_Size"]))})
                    d[row["Counter_Name"]] = float(row["Counter_Value"])
    shutil.rmtree(base, ignore_errors=True)
    order = [rows[k] for k in sorted(rows)]
    if len(order) != PMC_LAUNCHES * len(cells()):
        return {"error": f"{len(order)} european dispatches seen, {PMC_LAUNCHES * len(cells())} expected"}
    out = {}
    for i, cell in enumerate(cells()):
        mine = order[PMC_LAUNCHES * i: PMC_LAUNCHES * (i + 1)]
        out[cell] = {"SQ_INSTS_VALU": statistics.mean(d["SQ_INSTS_VALU"] for d in mine), "SQ_WAVES": statistics.mean(d["SQ_WAVES"] for d in mine),
                     "SQ_ACTIVE_INST_VALU": 1.0,      # roofline_for wants the key; this tool reports the issue-cycle fraction only
                     "grid_threads": mine[0]["grid_threads"], "strided": "true>" in mine[0]["kernel"].replace(" ", "")}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--pmc-child", action="store_true")
    ap.add_argument("--no-pmc", action="store_true")
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child()
    bench = load_bench()
    pmc = {"error": "skipped"} if args.no_pmc else collect_pmc()        # before this process touches the GPU
    if "error" in pmc:
        print(f"[grid] PMC pass unavailable: {pmc['error']}", file=sys.stderr)

    import optionslab_amd as ol
    from optionslab_amd import _hip

    info = _hip.device_info()
    mix = bench.load_isa_mix()
    bs = ol.black_scholes(*ATM, "call")
    sink = open(args.out, "w") if args.out else sys.stdout
    print(json.dumps({"device": info, "what": "blocking MonteCarloPricer.price(return_error=True), antithetic on; frac = issue cycles needed "
                      "(live SQ_INSTS_VALU of this launch shape, loop mix from the ISA, 2/4/8 passes per class) / (1024 SIMDs x kernel time x 2.4 GHz)",
                      "pmc": "live rocprofv3 pass of tools/measure_grid.py --pmc-child" if "error" not in pmc else pmc["error"]}), file=sink, flush=True)
    for k in range(2500):                                   # ~0.3 s of load first: sustained clocks
        _hip.european(*ATM, 0.0, True, 1_000_000, 252, k)
    for n, m in cells():
        p = ol.MonteCarloPricer(n, m, SEED)
        fn = lambda k=0: p.price(*ATM, "call", seed=SEED + k, return_error=True)
        t_end, warm = time.perf_counter() + 0.08, 0
        while warm < 3 or time.perf_counter() < t_end:      # >= 80 ms of THIS workload before it is timed
            fn(warm)
            warm += 1
        est = max(1e-6, (time.perf_counter() - (t_end - 0.08)) / warm)
        reps = int(min(200, max(7, 0.25 / est)))
        ts, worst = [], 0.0
        for k in range(reps):
            t0 = time.perf_counter()
            res = fn(k)
            ts.append(time.perf_counter() - t0)
            worst = max(worst, abs(res.price - bs) / res.std_error)
        _hip.profile_enable(True)
        _hip.profile_reset()
        for k in range(min(reps, 50)):
            fn(k)
        launches, kms = _hip.kernel_time()
        _hip.profile_enable(False)
        call_us, kernel_us = statistics.median(ts) * 1e6, kms / max(launches, 1) * 1e3
        row = {"n_paths": n, "n_steps": m, "path_steps": n * m, "call_us": call_us, "call_us_min": min(ts) * 1e6, "kernel_us": kernel_us, "reps": reps,
               "path_steps_per_s": n * m / (call_us * 1e-6), "paths_per_s": n / (call_us * 1e-6), "max_abs_err_over_sigma": worst}
        c = pmc.get((n, m)) if "error" not in pmc else None
        if c:
            r = bench.roofline_for({"c2_european": c}, "c2_european", kernel_us * 1e-6, m, n, None, mix)
            sol_us = r["speed_of_light_kernel_ms"] * 1e3
            row.update({"frac": r["frac"], "speed_of_light_us": sol_us, "valu_insts_per_launch": c["SQ_INSTS_VALU"], "waves_per_launch": c["SQ_WAVES"],
                        "workgroups": c["grid_threads"] // 256, "grid_striding_kernel": c["strided"],
                        "frac_of_call": sol_us / call_us})
            if r["frac"] < 0.5:
                inside, outside = kernel_us - sol_us, call_us - kernel_us
                row["floor"] = (f"kernel-internal fixed cost {inside:.1f} us (dispatch ramp of {c['grid_threads'] // 256} workgroups, per-path fp64 exp epilogue latency, "
                                f"ticket chain of the two-level reduction) against {sol_us:.1f} us of issue work; "
                                f"around the kernel {outside:.1f} us of host launch + PCIe completion word")
        print(json.dumps(row), file=sink, flush=True)
    if args.out:
        sink.close()


if __name__ == "__main__":
    main()
