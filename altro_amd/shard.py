"""Multi-GPU partitioning of the batch: problem instances are independent, so
GPU g simply owns the contiguous range [g*B/G, (g+1)*B/G) of a global batch -- no halo, no exchange
during sweeps.  The ONLY collective is the reduction of solver statistics (a few scalars, RCCL over
xGMI when the backend is "nccl"; gloo in the CPU tests).  torch.distributed is plumbing here."""
import numpy as np


def shard_range(global_batch, rank, world):
    """Contiguous, balanced [lo, hi) of the global problem index for `rank`."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def reduce_stats(stats, device=None, group=None):
    """All-reduce {problems, cholesky_failures, sum_delta_V0, sum_delta_V1} (SUM) and {max_abs_xN} (MAX).
    `stats` is anything with those attributes (altro_amd.Stats).  Returns a dict valid on every rank."""
    import torch
    import torch.distributed as dist
    ssum = torch.tensor([float(stats.problems), float(stats.cholesky_failures), float(stats.sum_delta_V0),
                         float(stats.sum_delta_V1)], dtype=torch.float64, device=device)
    smax = torch.tensor([float(stats.max_abs_xN)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(ssum, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(smax, op=dist.ReduceOp.MAX, group=group)
    return {"problems": int(round(ssum[0].item())), "cholesky_failures": int(round(ssum[1].item())),
            "sum_delta_V0": ssum[2].item(), "sum_delta_V1": ssum[3].item(), "max_abs_xN": smax[0].item()}


def max_over_ranks(value, device=None, group=None):
    """bench.py's timing rule: the slowest rank defines the step time."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
