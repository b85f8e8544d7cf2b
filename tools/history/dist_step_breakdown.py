#!/usr/bin/env python3
"""Where the sharded blocking step's time goes with ONE RCCL rank (GPU box): kernel on a torch stream -> all_reduce -> D2H -> sync,
adding one stage at a time.  Usage: python tools/dist_step_breakdown.py"""
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import optionslab_amd as ol  # noqa: E402
from optionslab_amd import _hip  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
dbuf = torch.zeros(3, dtype=torch.float64, device="cuda")
hbuf = torch.zeros(3, dtype=torch.float64).pin_memory()
P = (100.0, 100.0, 1.0, 0.05, 0.2, 0.0)
N, M = 1_000_000, 252


def launch(k):
    _hip.european_shard_dev(*P, True, 0, N, M, 42 + k, True, dbuf.data_ptr(), st.cuda_stream)


def v_kernel(k):
    launch(k); st.synchronize()


def v_allreduce(k):
    launch(k); dist.all_reduce(dbuf); st.synchronize()


def v_full(k):
    launch(k); dist.all_reduce(dbuf); hbuf.copy_(dbuf, non_blocking=True); st.synchronize(); return hbuf.tolist()


def v_full_item(k):
    launch(k); dist.all_reduce(dbuf); return dbuf.tolist()


def v_fetch(k):
    launch(k); dist.all_reduce(dbuf); return _hip.fetch_dev(dbuf.data_ptr(), 3, st.cuda_stream)


pricer = ol.MonteCarloPricer(N, M, 42)


def v_blocking_api(k):
    return pricer.price(*P[:5], "call", seed=42 + k, return_error=True)


def med(fn, reps=300, warm=300):
    for k in range(warm):
        fn(k)
    ts = []
    for k in range(reps):
        t0 = time.perf_counter(); fn(k); ts.append(time.perf_counter() - t0)
    return statistics.median(ts) * 1e6


for name, fn in (("blocking price() [no dist]", v_blocking_api), ("shard_dev on a torch stream + stream sync", v_kernel), ("+ dist.all_reduce (1 RCCL rank)", v_allreduce),
                 ("+ pinned D2H copy + sync + tolist (bench.py's step)", v_full), ("all_reduce then dbuf.tolist()", v_full_item), ("all_reduce then olmc_fetch_dev (publish kernel + polled word)", v_fetch)):
    print(json.dumps({"variant": name, "us_per_step": med(fn)}), flush=True)
dist.destroy_process_group()
