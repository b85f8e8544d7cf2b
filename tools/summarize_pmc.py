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
    # the default command by bench.py phase: only the single-stream pass (what bench.py's roofline.avg_kernel_ms times with HIP
    # events) has one kernel on the device at a time; in the 8-stream pass a dispatch's duration includes its co-residents'
    for f in newest(os.path.join(root, "stats_default", "*", "*_kernel_trace.csv")):
        rows = sorted((r for r in csv.DictReader(open(f)) if "european_path_kernel" in r["Kernel_Name"]), key=lambda r: int(r["Dispatch_Id"]))
        dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
        log = os.path.join(root, "stats_default.log")
        steps, warm, line = None, None, None
        if os.path.exists(log):
            import json
            for l in open(log):
                if l.startswith('{"metric"'):
                    line = json.loads(l)
                    steps, warm = line["steps"], line["warmup"]
        pre = line.get("pre_warm_pricings", 0) if line else 0
        dur = dur[pre:]                                   # untimed clock pre-warm of bench.py
        if steps and len(dur) >= warm + 2 * steps:
            print(f"\n## the default command's european_path_kernel dispatches by bench.py phase (us), after {pre} pre-warm dispatches")
            for name, a, b in (("warm-up, 8 streams", 0, warm), ("timed K steps, 8 streams (-> value)", warm, warm + steps),
                               ("same K steps, 1 stream (-> serial, roofline)", warm + steps, warm + 2 * steps), ("blocking price() calls (-> sync_call)", warm + 2 * steps, len(dur))):
                seg = dur[a:b]
                if seg:
                    print(f"{name:55s} n={len(seg):4d}  avg {sum(seg) / len(seg):8.1f}  min {min(seg):8.1f}  max {max(seg):8.1f}")
            print(f"bench.py of that run reported roofline.avg_kernel_ms = {line['roofline']['avg_kernel_ms']:.4f} (HIP events, {line['roofline']['launches_timed']} launches), value = {line['value']:.4g} {line['unit']}")
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
