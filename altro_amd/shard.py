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
    (already initialised by the launcher's rendezvous) carries its 128 bytes to the other ranks.  Every step that can
    fail on one rank only is followed by an agreement over torch.distributed, so that either ALL ranks hold a
    communicator or all of them raise -- never some inside ncclCommInitRank while others have given up."""
    import torch
    import torch.distributed as dist
    import altro_amd
    multi = world > 1
    on = None
    if multi:
        on = "cuda" if dist.get_backend() == "nccl" else "cpu"
    ok, uid, err = 1, bytes(altro_amd.COMM_ID_BYTES), None
    if rank == 0:
        try:
            uid = altro_amd.Comm.unique_id()
        except Exception as e:   # noqa: BLE001 -- e.g. librccl cannot be loaded: tell the other ranks instead of leaving them waiting
            ok, err = 0, e
    if multi:
        t = torch.tensor([ok] + list(uid), dtype=torch.uint8, device=on)
        dist.broadcast(t, src=0)
        vals = t.cpu().tolist()
        ok, uid = vals[0], bytes(vals[1:])
    if not ok:
        raise altro_amd.AltroHipError("rank 0 could not draw an RCCL unique id%s" % ((": %s" % err) if err else ""))
    comm, err = None, None
    try:
        comm = altro_amd.Comm(device, rank, world, uid)
    except Exception as e:   # noqa: BLE001
        err = e
    if multi:
        flag = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=on)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if comm is not None:
                comm.close()
            raise altro_amd.AltroHipError("ncclCommInitRank failed on at least one rank%s" % ((": %s" % err) if err else ""))
    elif comm is None:
        raise err
    return comm


def max_over_ranks(value, device=None, group=None):
    """bench.py's timing rule: the slowest rank defines the step time."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
