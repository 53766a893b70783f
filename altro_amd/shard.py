"""Multi-GPU partitioning of the batch: problem instances are independent, so
GPU g simply owns the contiguous range [g*B/G, (g+1)*B/G) of a global batch -- no halo, no exchange
during sweeps.  The ONLY collective is the reduction of solver statistics -- altro_hip_stats_allreduce in the C ABI
(device-side reduction + two ncclAllReduce on RCCL over xGMI); `reduce_stats` is the same reduction through
torch.distributed for the places RCCL cannot run (gloo in the CPU tests).  torch.distributed is plumbing here."""
import numpy as np


def shard_range(global_batch, rank, world):
    """Contiguous, balanced [lo, hi) of the global problem index for `rank`."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def reduce_stats(stats, device=None, group=None):
    """All-reduce an altro_amd.Stats (or anything with its fields) through torch.distributed: the SUM fields as one
    float64 vector, the MAX fields as another -- the same two calls altro_hip_stats_allreduce makes on RCCL from C.
    Used where the C entry cannot be (gloo on CPU in the tests; bench.py's single-box gloo hook).  Returns a dict
    valid on every rank."""
    import torch
    import torch.distributed as dist
    from altro_amd import Stats
    ssum = torch.tensor([float(getattr(stats, k, 0)) for k in Stats.SUM_FIELDS], dtype=torch.float64, device=device)
    smax = torch.tensor([float(getattr(stats, k, 0.0)) for k in Stats.MAX_FIELDS], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(ssum, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(smax, op=dist.ReduceOp.MAX, group=group)
    out = {}
    for k, v in zip(Stats.SUM_FIELDS, ssum.tolist()):
        out[k] = int(round(v)) if k in ("problems", "cholesky_failures", "converged", "iterations", "non_finite") else v
    out.update(zip(Stats.MAX_FIELDS, smax.tolist()))
    return out


def make_comm(device, rank, world):
    """One RCCL communicator rank for altro_hip_stats_allreduce: rank 0 draws the unique id, torch.distributed
    (already initialised by the launcher's rendezvous) carries its 128 bytes to the other ranks."""
    import torch
    import torch.distributed as dist
    import altro_amd
    uid = altro_amd.Comm.unique_id() if rank == 0 else bytes(altro_amd.COMM_ID_BYTES)
    if world > 1:
        on = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor(list(uid), dtype=torch.uint8, device=on)
        dist.broadcast(t, src=0)
        uid = bytes(t.cpu().tolist())
    return altro_amd.Comm(device, rank, world, uid)


def max_over_ranks(value, device=None, group=None):
    """bench.py's timing rule: the slowest rank defines the step time."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
