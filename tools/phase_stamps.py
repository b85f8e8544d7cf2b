#!/usr/bin/env python3
"""Apportions the non-amortised part of a headline launch (VERDICT r2 #4a): one instrumented launch of the European kernel
(olmc_phase_stamps: same device functions, same launch shape, s_memrealtime stamps of wave 0 of every workgroup) -> where the
microseconds between the dispatch's begin and end timestamps go.

    python tools/phase_stamps.py [n_paths] [n_steps] [repeats]        (GPU box; prints one JSON object per launch size)
"""
import json
import os
import statistics
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optionslab_amd import _hip  # noqa: E402
from tools.probe import binding as probe  # noqa: E402  (the instrumented build: include/olmc_probe.h)

SIZES = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(65_536, 252), (458_752, 252), (1_000_000, 252), (8_000_000, 252), (1_000_000, 1024)]
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 15
TICK_US = 0.01          # s_memrealtime: 100 MHz


def one(n, m, seed):
    # 30 identical launches queued back to back in front of the recorded one: it runs at the clock the production kernel runs at
    st, final, split_from, disp_ns = probe.phase_stamps(n, m, seed, lead_launches=30)
    where = st[:, 4]
    st = st[:, :4]
    t0 = int(st[:, 0].min())
    rel = (st - t0) * TICK_US
    fin = (final - t0) * TICK_US
    full, split = rel[:split_from], rel[split_from:]
    row = {
        "dispatch_us": disp_ns / 1e3,                                   # the kernel's own begin / end timestamps (what rocprofv3 and bench.py report)
        "shader_span_us": fin,                                          # first wave's entry -> totals written
        "outside_shader_us": disp_ns / 1e3 - fin,                       # command processor: begin stamp -> first wave, last store -> end stamp
        "ramp_last_entry_of_first_round_us": float(np.sort(rel[:, 0])[min(len(rel), 256 * 7) - 1]),   # when the 1792nd workgroup (7 per CU) has entered
        "last_entry_us": float(rel[:, 0].max()),
        "first_loop_done_us": float(rel[:, 1].min()),
        "last_loop_done_us": float(rel[:, 1].max()),
        "last_full_loop_done_us": float(full[:, 1].max()) if len(full) else None,
        "last_split_loop_done_us": float(split[:, 1].max()) if len(split) else None,
        "last_sums_done_us": float(rel[:, 2].max()),                    # payoffs (2 fp64 exp) + workgroup reduction of the last workgroup
        "last_ticket_us": float(rel[:, 3].max()),
        "totals_written_us": fin,
        "median_loop_us": float(np.median(full[:, 1] - full[:, 0])) if len(full) else None,
        "median_split_loop_us": float(np.median(split[:, 1] - split[:, 0])) if len(split) else None,
        "median_epilogue_us": float(np.median(rel[:, 2] - rel[:, 1])),
        "median_ticket_us": float(np.median(rel[:, 3] - rel[:, 2])),
        "chain_after_last_sums_us": fin - float(rel[:, 2].max()),       # two dependent ticket / row-sum round trips + the store of the totals
        "workgroups": int(len(rel)), "split_from": int(split_from),
    }
    # where the workgroups ran: (xcc, se, sh, cu) from HW_REG_XCC_ID / HW_REG_HW_ID
    hw = where & 0xFFFFFFFF
    xcc = (where >> 32) & 0xF
    cu_key = (xcc << 16) | (hw & 0xFF00)                  # se_id 15:13, sh_id 12, cu_id 11:8
    cus = np.unique(cu_key)
    busy_end = np.array([rel[cu_key == c, 1].max() for c in cus])          # when each CU's last step loop ended
    first_in = np.array([rel[cu_key == c, 0].min() for c in cus])
    per_cu = np.array([(cu_key == c).sum() for c in cus])
    loop = rel[:, 1] - rel[:, 0]
    is_full = np.arange(len(rel)) < split_from
    row.update({
        "cus_seen": int(len(cus)), "xccs_seen": int(len(np.unique(xcc))),
        "cu_first_entry_us": {"min": float(first_in.min()), "median": float(np.median(first_in)), "max": float(first_in.max())},
        "cu_last_loop_done_us": {"min": float(busy_end.min()), "p10": float(np.percentile(busy_end, 10)), "median": float(np.median(busy_end)),
                                 "p90": float(np.percentile(busy_end, 90)), "max": float(busy_end.max())},
        # CU-microseconds between a CU's last loop end and the launch's last loop end, averaged per CU: the drain
        "mean_cu_idle_before_last_loop_done_us": float((busy_end.max() - busy_end).mean()),
        "workgroups_per_cu": {"min": int(per_cu.min()), "median": float(np.median(per_cu)), "max": int(per_cu.max())},
        "xcc_median_full_loop_us": {int(x): float(np.median(loop[(xcc == x) & is_full])) for x in np.unique(xcc) if ((xcc == x) & is_full).any()},
        "xcc_last_loop_done_us": {int(x): float(rel[xcc == x, 1].max()) for x in np.unique(xcc)},
        "xcc_workgroups": {int(x): int((xcc == x).sum()) for x in np.unique(xcc)},
    })
    return row


def main():
    _hip.lib()
    if os.environ.get("OLMC_AB_TUNE"):                      # "knob=value,knob=value", as tools/ab_libs.py
        for kv in os.environ["OLMC_AB_TUNE"].split(","):
            k, v = kv.split("=")
            _hip.tune(int(k), int(v))
    for k in range(2500):
        _hip.european(100.0, 100.0, 1.0, 0.05, 0.2, 0.0, True, 1_000_000, 252, k, True)
    print(json.dumps({"device": _hip.device_info(), "what": "medians over launches of olmc_phase_stamps; microseconds from the first wave's entry stamp"}), flush=True)
    for n, m in SIZES:
        for k in range(200):
            _hip.european(100.0, 100.0, 1.0, 0.05, 0.2, 0.0, True, n, m, k, True)
        rows = [one(n, m, 42 + k) for k in range(REPS)]
        def med_of(vals):
            if vals[0] is None:
                return None
            if isinstance(vals[0], dict):
                return {kk: statistics.median(v[kk] for v in vals if kk in v) for kk in vals[0]}
            return statistics.median(vals)
        med = {k: med_of([r[k] for r in rows]) for k in rows[0]}
        # the same launch uninstrumented, by its dispatch timestamps (after load again: the instrumented launches idle the device
        # between their memset, kernel and 300 KB copy, and the clock sags within milliseconds of idling)
        for k in range(300):
            _hip.european(100.0, 100.0, 1.0, 0.05, 0.2, 0.0, True, n, m, k, True)
        _hip.profile_enable(True)
        _hip.profile_reset()
        for k in range(50):
            _hip.european(100.0, 100.0, 1.0, 0.05, 0.2, 0.0, True, n, m, k, True)
        launches, ms = _hip.kernel_time()
        _hip.profile_enable(False)
        med.update({"n_paths": n, "n_steps": m, "launches": REPS, "production_kernel_us": ms / launches * 1e3})
        print(json.dumps(med), flush=True)


if __name__ == "__main__":
    main()
