#!/usr/bin/env python3
"""Summarise the output of tools/profile_gpu.sh into one text file.
Usage: tools/summarize_pmc.py gpurun_out/prof_<tag> > profiles/<name>.txt"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    return name.split("(")[0].replace("void ", "")[:78]


def newest(pattern):
    """gpurun merges every call's output into the same local directory: keep the latest run of a pass only."""
    files = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return files[-1:]


def bench_line(path):
    """The record of a bench.py pass: the full one (<pass>.detail.json, written next to the one-line JSON) when it is there,
    else the stdout line."""
    stem = path[:-len(".log")] if path.endswith(".log") else path[:-len(".json")]
    try:
        with open(stem + ".detail.json") as f:
            return json.load(f)
    except (OSError, ValueError):
        pass
    try:
        for l in open(path):
            if l.startswith('{"metric"'):
                return json.loads(l)
    except OSError:
        pass
    return None


def main(root):
    print(f"# rocprofv3 summary of {root} (tools/profile_gpu.sh)")
    for sub, title in (("stats", "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 100 --warmup 10 --no-pmc --no-cpu-baseline --no-extras"),
                       ("stats_full", "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline   (all sections)")):
        for f in newest(os.path.join(root, sub, "**", "*_kernel_stats.csv")):
            print(f"\n## {title}")
            for row in csv.DictReader(open(f)):
                print(f"{short(row['Name']):78s} calls {row['Calls']:>5s}  avg_ns {float(row['AverageNs']):>12.1f}  min {row['MinNs']:>9s}  max {row['MaxNs']:>9s}  {row['Percentage']}%")
        line = bench_line(os.path.join(root, sub + ".log"))
        if line:
            r = line["roofline"]
            print(f"bench.py of that (profiled) run: value {line['value']:.4g} {line['unit']}, ms_per_step {line['ms_per_step']:.4f}, "
                  f"roofline.avg_kernel_ms {r['avg_kernel_ms']:.4f} (HIP events attached to {r.get('launches_timed')} dispatches)")
        # the headline kernel's dispatches by bench.py phase: pre-warm / warm-up / timed passes
        for f in newest(os.path.join(root, sub, "**", "*_kernel_trace.csv")):
            rows = sorted((r for r in csv.DictReader(open(f)) if "european_path_kernel<1, true, 0, false>" in r["Kernel_Name"]), key=lambda r: int(r["Dispatch_Id"]))
            dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
            if line and sub == "stats" and dur:
                pre = line.get("pre_warm_pricings", 0)
                timed = dur[pre + line["warmup"]:]
                print(f"european_path_kernel<1> dispatches of that run: {len(dur)} in all; the {pre} pre-warm ones avg {sum(dur[:pre]) / max(pre, 1):.2f} us; "
                      f"the {len(timed)} of the timed passes (plain + instrumented) avg {sum(timed) / max(len(timed), 1):.2f} us, min {min(timed):.2f}, max {max(timed):.2f}")
    line = bench_line(os.path.join(root, "bench.json"))
    if line:
        r = line["roofline"]
        print("\n## python3 bench.py --steps 20 --warmup 5 (unprofiled; its live PMC passes are the tables below)")
        print(f"value {line['value']:.4g} {line['unit']}  ms_per_step {line['ms_per_step']:.4f}  avg_kernel_ms {r['avg_kernel_ms']:.4f}  frac {r['frac'] if r.get('frac') is None else format(r['frac'], '.4f')}  "
              f"frac_vs_isolated_rates {r.get('frac_vs_isolated_rates')}  frac_valu_active_pmc {r.get('frac_valu_active_pmc')}  traffic {r.get('traffic')} B/launch")
    for p in ("sq", "fetch", "write"):
        files = newest(os.path.join(root, "pmc", p, "**", "*_counter_collection.csv"))
        if not files:
            continue
        agg = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(files[0])):
            agg[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print(f"\n## live --pmc pass '{p}' of that bench.py run (rocprofv3 --kernel-trace --pmc ... -- python3 bench.py --pmc-child; mean per dispatch)")
        for k, ctrs in agg.items():
            if "olmc::" not in k:
                continue
            for c, v in sorted(ctrs.items()):
                print(f"{k:78s} {c:22s} n={len(v):4d} mean={sum(v) / len(v):.6g}")


if __name__ == "__main__":
    main(sys.argv[1])
