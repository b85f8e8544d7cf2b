"""Host spans of the single-process multi-GPU engine (olmc_multi_gpu_european), n ranks REHEARSED on the one device of the box in the
instrumented build: how long it takes to queue every rank's path kernel (launch_us: first rank's launch begun -> last rank's kernel
queued), the collective, the fetch of the result and the drain of the other ranks -- with one launcher thread per rank (default) and
with round 4's serial form (OLMC_TUNE_MULTI_LAUNCH = -1).  One JSON line per (ranks, paths per rank, form).

    python3 tools/measure_multi_enqueue.py > profiles/r05_multi_enqueue.jsonl

The stagger of the ranks' kernel STARTS is what costs weak-scaling efficiency on real devices (on one device the kernels queue
behind each other anyway, so `total_us` here says nothing about 8 GPUs; `launch_us` does: it is host time)."""
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from tools.probe import binding as probe  # noqa: E402

hip = probe.hip
S, K, T, r, v = 100.0, 100.0, 1.0, 0.05, 0.2


def main():
    info = hip.device_info()
    probe.tune(probe.TUNE_MULTI_REHEARSAL, 1)
    reps = int(os.environ.get("OLMC_ENQUEUE_REPS", "60"))
    for per_rank in (1_000_000, 8_000_000):
        for n_ranks in (1, 2, 4, 8):
            if per_rank * n_ranks > (1 << 26):
                continue
            for form, knob in (("launcher threads", 0), ("serial", -1)):
                hip.tune(hip.TUNE_MULTI_LAUNCH, knob)
                n = per_rank * n_ranks
                for k in range(5):
                    hip.multi_gpu_european(S, K, T, r, v, 0.0, True, n, 252, 42 + k, True, n_ranks)
                spans, wall = [], []
                for k in range(reps if per_rank == 1_000_000 else max(10, reps // 4)):
                    t0 = time.perf_counter()
                    st = hip.multi_gpu_european(S, K, T, r, v, 0.0, True, n, 252, 100 + k, True, n_ranks)
                    wall.append((time.perf_counter() - t0) * 1e6)
                    spans.append(hip.multi_gpu_spans())
                    assert st.n == 2 * n
                med = {key: statistics.median(s[key] for s in spans) for key in spans[0]}
                p90 = sorted(s["launch_us"] for s in spans)[int(0.9 * (len(spans) - 1))]
                print(json.dumps({"ranks": n_ranks, "paths_per_rank": per_rank, "n_steps": 252, "form": form, "calls": len(spans),
                                  "launch_us": round(med["launch_us"], 2), "launch_us_p90": round(p90, 2), "collective_us": round(med["collective_us"], 2),
                                  "fetch_us": round(med["fetch_us"], 2), "drain_us": round(med["drain_us"], 2), "total_us": round(med["total_us"], 2),
                                  "wake_us_max": round(med["wake_us_max"], 2), "rank_launch_us_max": round(med["rank_launch_us_max"], 2), "rank_launch_us_min": round(med["rank_launch_us_min"], 2),
                                  "wall_us_python": round(statistics.median(wall), 2), "device": info["name"], "rehearsal": True}), flush=True)
    hip.tune(hip.TUNE_MULTI_LAUNCH, 0)
    probe.tune(probe.TUNE_MULTI_REHEARSAL, 0)
    hip.shutdown()


if __name__ == "__main__":
    main()
