#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output of tools/profile_gpu.sh into one text table per kernel.
Usage: tools/summarize_pmc.py gpurun_out/prof_<tag> > profiles/<name>.txt"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    return name.split("(")[0].replace("void ", "")[:70]


def newest(pattern):
    """gpurun merges every call's output into the same local directory: keep the latest run of a pass only."""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:]


def main(root):
    print(f"# rocprofv3 summary of {root}")
    for sub, title in (("stats", "bench.py --streams 1 (launches back to back, no overlap)"), ("stats_default", "bench.py default (8 streams; the serial pass is included, so durations mix overlapped and serial launches)")):
      for f in newest(os.path.join(root, sub, "*", "*_kernel_stats.csv")):
        print(f"\n## kernel-trace --stats: {title}")
        for row in csv.DictReader(open(f)):
            print(f"{short(row['Name']):70s} calls {row['Calls']:>5s}  avg_ns {float(row['AverageNs']):>12.1f}  min {row['MinNs']:>8s}  max {row['MaxNs']:>8s}  {row['Percentage']}%")
    for p in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write"):
        files = newest(os.path.join(root, p, "*", "*_counter_collection.csv"))
        if not files:
            continue
        agg = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(files[0])):
            agg[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print(f"\n## --pmc pass {p} (mean per dispatch)")
        for k, ctrs in agg.items():
            for c, v in sorted(ctrs.items()):
                print(f"{k:70s} {c:24s} n={len(v):4d} mean={sum(v)/len(v):.6g}")


if __name__ == "__main__":
    main(sys.argv[1])
