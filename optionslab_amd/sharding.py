"""Sharding independent path batches over GPUs.

Global path index g in [0, N); rank k of P owns the contiguous block
[k*N/P, (k+1)*N/P).  The Philox counter carries g, so every path's normals are
the same whatever P is.  The only exchange is one all-reduce (sum) of the
triple (sum_x, sum_xx, n) as 3 x fp64 = 24 bytes -- RCCL over xGMI when the
process group's backend is "nccl", gloo on CPU in the tests.

torch is imported lazily and used only as plumbing (device buffer, stream,
process group); single-GPU pricing never touches it.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

from . import _hip


def shard_bounds(n_paths: int, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) of rank `rank`; the P blocks tile [0, n_paths) without gaps."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    if n_paths < world:
        raise ValueError("fewer paths than ranks")
    return n_paths * rank // world, n_paths * (rank + 1) // world


def finalize(sum_x: float, sum_xx: float, n: int, r: float, T: float) -> Tuple[float, float]:
    """(price, std_error) from the reduced triple; identical on every rank
    (formulae of src/pricing_models/monte_carlo.py:145-150)."""
    disc, mean = math.exp(-r * T), sum_x / n
    var = max(sum_xx / n - mean * mean, 0.0)
    return disc * mean, disc * math.sqrt(var) / math.sqrt(n)


def allreduce_triple(triple, group=None):
    """Sum a (3,) float64 tensor over the process group in place and return it.
    CUDA tensor + nccl backend = RCCL over xGMI; CPU tensor + gloo in tests."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(triple, op=dist.ReduceOp.SUM, group=group)
    return triple


def price_european_sharded(S, K, T, r, sigma, option_type, q, n_paths_global: int, n_steps: int, seed: int,
                           antithetic: bool = True, group=None, device_buffer=None):
    """One rank's part of a sharded price(): local HIP kernel on this rank's path
    block -> all-reduce of the triple -> same (price, std_error, n) on every rank."""
    import torch
    import torch.distributed as dist

    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    lo, hi = shard_bounds(n_paths_global, rank, world)
    buf = device_buffer if device_buffer is not None else torch.empty(3, dtype=torch.float64, device="cuda")
    _hip.european_shard_dev(S, K, T, r, sigma, q, option_type == "call", lo, hi - lo, n_steps, seed, antithetic,
                            buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
    allreduce_triple(buf, group)
    s, ss, n = buf.cpu().tolist()
    price, se = finalize(s, ss, int(n), r, T)
    return price, se, int(n)
