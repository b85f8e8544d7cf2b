#!/usr/bin/env python3
"""Prints the figures of a bench.py JSON line that one looks at first.  Usage: python tools/show_bench.py file.json"""
import json
import sys

d = json.load(open(sys.argv[1]))
print(f"value {d['value']:.4g} {d['unit']}  ms_per_step {d['ms_per_step']:.4f}  instrumented {d.get('instrumented_passes', {}).get('ms_per_step', float('nan')):.4f}  n_gpus {d['n_gpus']}")
r = d["roofline"]
keys = ("achieved", "peak", "frac", "avg_kernel_ms", "speed_of_light_kernel_ms", "frac_vs_isolated_rates", "frac_valu_active_pmc", "clock_ghz_under_load", "traffic", "hbm_gbps")
print("roofline", {k: (round(r[k], 4) if isinstance(r.get(k), float) else r.get(k)) for k in keys})
if d.get("clock"):
    print("clock", d["clock"])
if "cpu_baseline" in d:
    c = d["cpu_baseline"]
    print("cpu", f"{c['value']:.4g}", "cores", c["cores"], "all_cores", c.get("all_cores", {}).get("value"))
for k in ("c5_weak", "c5_strong", "pipelined"):
    if k in d:
        print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in d[k].items() if a not in ("workload", "what", "dtype", "unit")})
for sec in ("c3_greeks", "c4_asian"):
    for k, v in d.get(sec, {}).items():
        if isinstance(v, dict) and "ms_per_call" in v:
            rr = v.get("roofline") or {}
            print(sec, k, f"ms/call {v['ms_per_call']:.4f} kernel {v['avg_kernel_ms']:.4f} path-steps/s {v['path_steps_per_s']:.4g}",
                  "frac", rr.get("frac") and round(rr["frac"], 3), "iso", rr.get("frac_vs_isolated_rates") and round(rr["frac_vs_isolated_rates"], 3),
                  "pmc", rr.get("frac_valu_active_pmc") and round(rr["frac_valu_active_pmc"], 3))
if d.get("errors"):
    print("ERRORS", d["errors"])
