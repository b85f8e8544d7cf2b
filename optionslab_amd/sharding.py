"""Sharding independent path batches over GPUs.

    price_european_sharded      one contract: all-reduce of (sum, sumsq, n)                 count = 3
    greeks_sharded              the 8 / 14 bumped contracts of compute_greeks_unified on the
                                SAME normals, one launch per rank, one all-reduce            count = 2k + 1
    control_variate_sharded     the five moments + n                                        count = 6
    price_sharded               any entry point that takes (path_offset, n_local): Asian, barrier, lookback,
                                autocallable, cliquet, Heston, jump diffusion               count = 3
    qmc_sharded                 MCMethod.QMC: contiguous blocks of Sobol POINTS (gbm_qmc.py:14-46)        count = 3
    qmc_greeks_sharded, qmc_control_variate_sharded   the same two on blocks of Sobol points (cut by qmc_shard_bounds)   count = 2k + 1, 6

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
from typing import Tuple

from . import _hip


def shard_bounds(n_paths: int, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) of rank `rank`; the P blocks tile [0, n_paths) without gaps."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    if n_paths < world:
        raise ValueError("fewer paths than ranks")
    return n_paths * rank // world, n_paths * (rank + 1) // world


QMC_SHARD_ALIGN, QMC_SHARD_MIN_POINTS = 512, 4096


def qmc_shard_bounds(n_points: int, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) of rank `rank` over Sobol POINTS: shard_bounds with every inner boundary rounded down to a multiple of 512 points
    where a rank then still owns 4,096 or more, so that every rank's point offset is one the aligned Sobol kernels take (an
    unaligned offset costs a rank 1.5 x the time).  The cut olmc_multi_gpu_european_qmc makes (olmc_host_math.h qmc_shard_range)."""
    lo, hi = shard_bounds(n_points, rank, world)
    if n_points // world < QMC_SHARD_MIN_POINTS:
        return lo, hi
    return lo // QMC_SHARD_ALIGN * QMC_SHARD_ALIGN, (n_points if rank + 1 == world else hi // QMC_SHARD_ALIGN * QMC_SHARD_ALIGN)


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
    if buf.is_cuda and (not dist.is_initialized() or dist.get_backend(group) == "nccl"):
        # behind the RCCL all-reduce on the caller's stream: hand the triple over through the library's pinned buffer and
        # completion word instead of a D2H copy + stream synchronise
        s, ss, n = _hip.fetch_dev(buf.data_ptr(), 3, torch.cuda.current_stream().cuda_stream)
    else:
        s, ss, n = buf.cpu().tolist()
    price, se = finalize(s, ss, int(n), r, T)
    return price, se, int(n)


# --------------------------------------------------------------------------------------------------
def _group_info(group=None):
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _allreduce_list(values, group=None):
    """Sum a list of floats over the group (fp64).  nccl backend -> a CUDA tensor (RCCL), otherwise CPU."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [float(v) for v in values]
    device = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor(values, dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().tolist()


def price_sharded(shard_fn, n_paths_global: int, r_discount: float, T: float, group=None, bounds=shard_bounds):
    """Generic sharded pricing.  `shard_fn(path_offset, n_local)` prices this rank's block and returns an
    object with (.sum, .sumsq, .n) -- e.g. ``lambda lo, n: _hip.asian(S, K, T, r, v, q, True, False, n, M, seed,
    False, path_offset=lo)``.  Returns (price, std_error, n) identical on every rank; pass r_discount = 0 for
    payoffs that carry their own discounting (autocallable, American)."""
    rank, world = _group_info(group)
    lo, hi = bounds(n_paths_global, rank, world)
    st = shard_fn(lo, hi - lo)
    s, ss, n = _allreduce_list([st.sum, st.sumsq, float(st.n)], group)
    price, se = finalize(s, ss, int(n), r_discount, T)
    return price, se, int(n)


def qmc_sharded(S, K, T, r, sigma, option_type, q, n_points_global: int, n_steps: int, seed: int, group=None, shard_fn=None):
    """MonteCarloPricer(method=MCMethod.QMC).price over sharded POINTS (src/simulation/gbm_qmc.py:14-46): rank k prices points
    [k N / P, (k + 1) N / P) of the one scrambled Sobol sequence (inner boundaries on multiples of 512 points: qmc_shard_bounds) through
    the kernels' point offset -- the same points whatever P is --
    and the ranks' (sum, sumsq, n) meet in one all-reduce.  `shard_fn(lo, n_local)` -> an object with (.sum, .sumsq, .n) defaults to
    the device kernel (tests inject the oracle).  The single-process form is olmc_multi_gpu_european_qmc."""
    if shard_fn is None:
        from .monte_carlo import sobol_tables

        sv, shift = sobol_tables(n_steps, seed, n_points_global)
        shard_fn = lambda lo, n: _hip.european_qmc(S, K, T, r, sigma, q, option_type == "call", n, sv, shift, point_offset=lo)   # noqa: E731
    return price_sharded(shard_fn, n_points_global, r, T, group, bounds=qmc_shard_bounds)


class _Recorder:
    """Stands in for a pricer while compute_greeks_unified lists the contracts it wants."""

    def __init__(self):
        self.contracts = []

    def price(self, S, K, T, r, sigma, option_type, q=0.0, **kw):
        self.contracts.append((S, K, T, r, sigma, q))
        return 1.0 + 0.001 * len(self.contracts)      # any finite number: the formulas are evaluated again on replay


class _Replayer:
    def __init__(self, table):
        self.table = table

    def price(self, S, K, T, r, sigma, option_type, q=0.0, **kw):
        return self.table[(S, K, T, r, sigma, q)]


def _device_batch(contracts, is_call, lo, n_local, n_steps, seed, antithetic):
    options = [(S, K, T, r, v, q, is_call) for (S, K, T, r, v, q) in contracts]
    return [(st.sum, st.sumsq, st.n) for st in _hip.european_batch(options, n_local, n_steps, seed, antithetic, path_offset=lo)]


def greeks_sharded(S, K, T, r, sigma, option_type, q, n_paths_global: int, n_steps: int, seed: int,
                   include_second_order: bool = True, antithetic: bool = True, group=None, batch_fn=None, bounds=shard_bounds):
    """compute_greeks_unified (src/greeks/unified_greeks.py:235-367) over sharded paths: every rank prices the
    8 / 14 bumped contracts on ITS block of the common normals in one launch, ONE all-reduce carries the 2k sums
    and n, and every rank evaluates the same finite differences.  `batch_fn(contracts, is_call, lo, n_local,
    n_steps, seed, antithetic) -> [(sum, sumsq, n)]` defaults to the device batch kernel (tests inject the CPU
    checker).  Returns the reference's OrderedDict."""
    from .greeks import compute_greeks_unified

    rec = _Recorder()
    compute_greeks_unified(rec, S, K, T, r, sigma, option_type, q, include_second_order, fused=False)
    contracts = list(dict.fromkeys(rec.contracts))              # the reference memoises on the same key
    rank, world = _group_info(group)
    lo, hi = bounds(n_paths_global, rank, world)
    local = (batch_fn or _device_batch)(contracts, option_type == "call", lo, hi - lo, n_steps, seed, antithetic)
    flat = [v for (sx, sxx, _n) in local for v in (sx, sxx)] + [float(local[0][2])]
    red = _allreduce_list(flat, group)
    n = int(red[-1])
    table = {c: finalize(red[2 * i], red[2 * i + 1], n, c[3], c[2])[0] for i, c in enumerate(contracts)}
    return compute_greeks_unified(_Replayer(table), S, K, T, r, sigma, option_type, q, include_second_order, fused=False)


def control_variate_sharded(S, K, T, r, sigma, option_type, q, n_paths_global: int, n_steps: int, seed: int,
                            antithetic: bool = True, group=None, shard_fn=None, bounds=shard_bounds):
    """MonteCarloPricer.price_with_control_variate (monte_carlo.py:154-186) over sharded paths: the five moments
    and n are summed over ranks, beta and the estimate follow on every rank.  `shard_fn(lo, n_local)` -> an object
    with the olmc_cv_moments fields (default: the device kernel)."""
    rank, world = _group_info(group)
    lo, hi = bounds(n_paths_global, rank, world)
    fn = shard_fn or (lambda lo_, n_: _hip.european_cv_shard(S, K, T, r, sigma, q, option_type == "call", lo_, n_, n_steps, seed, antithetic))
    m = fn(lo, hi - lo)
    sd, ss, sdd, sss, sds, n = _allreduce_list([m.sum_d, m.sum_s, m.sum_dd, m.sum_ss, m.sum_ds, float(m.n)], group)
    n = float(int(n))
    mean_d, mean_s = sd / n, ss / n
    cov_ds = (sds - n * mean_d * mean_s) / (n - 1.0)            # np.cov, ddof = 1 (:181)
    var_s = (sss - n * mean_s * mean_s) / (n - 1.0)
    beta = cov_ds / var_s if (n > 1.0 and var_s > 1e-10) else 0.0          # :182
    return mean_d - beta * (mean_s - S * math.exp((r - q) * T))              # :184


def qmc_greeks_sharded(S, K, T, r, sigma, option_type, q, n_points_global: int, n_steps: int, seed: int,
                       include_second_order: bool = True, group=None, batch_fn=None):
    """compute_greeks_unified on a MCMethod.QMC pricer over sharded POINTS: every rank prices the 8 / 14 bumped contracts on its block
    of the one Sobol sequence (qmc_shard_bounds) in one launch, one all-reduce of the 2k sums and n.  `batch_fn` as in greeks_sharded
    (its `antithetic` argument is False: no mirror, gbm_qmc.py:44-46).  The single-process form is olmc_multi_gpu_european_qmc_greeks_fd."""
    if batch_fn is None:
        from .monte_carlo import sobol_tables

        sv, shift = sobol_tables(n_steps, seed, n_points_global)

        def batch_fn(contracts, is_call, lo, n_local, _n_steps, _seed, _antithetic):
            options = [(S_, K_, T_, r_, v_, q_, is_call) for (S_, K_, T_, r_, v_, q_) in contracts]
            return [(st.sum, st.sumsq, st.n) for st in _hip.european_qmc_batch(options, n_local, sv, shift, point_offset=lo)]
    return greeks_sharded(S, K, T, r, sigma, option_type, q, n_points_global, n_steps, seed, include_second_order, False, group, batch_fn,
                          bounds=qmc_shard_bounds)


def qmc_control_variate_sharded(S, K, T, r, sigma, option_type, q, n_points_global: int, n_steps: int, seed: int, group=None, shard_fn=None):
    """price_with_control_variate on a MCMethod.QMC pricer over sharded POINTS (qmc_shard_bounds): the five moments of every rank's
    block and n meet in one all-reduce.  The single-process form is olmc_multi_gpu_european_qmc_cv."""
    if shard_fn is None:
        from .monte_carlo import sobol_tables

        sv, shift = sobol_tables(n_steps, seed, n_points_global)
        shard_fn = lambda lo, n: _hip.european_qmc_cv(S, K, T, r, sigma, q, option_type == "call", n, sv, shift, point_offset=lo)   # noqa: E731
    return control_variate_sharded(S, K, T, r, sigma, option_type, q, n_points_global, n_steps, seed, False, group, shard_fn,
                                   bounds=qmc_shard_bounds)
