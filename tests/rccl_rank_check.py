"""Helper of tests/test_gpu_parity.py::test_rccl_and_the_hip_library_share_a_process: one rank of bench.py's launch form
(`python -m torch.distributed.run ...`, backend "nccl" == RCCL) that FORCES the collectives to run even at world size 1,
so that RCCL communicator setup, an all-reduce and a barrier happen in the same process -- and on the same device -- as
libaltro_hip.so's streams and kernels.  (The 8-GPU runs belong to the driver; this is what one 1-GPU box can prove.)"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import altro_amd  # noqa: E402
from altro_amd import shard  # noqa: E402
from tests import problems  # noqa: E402


def main():
    rank, world, local_rank = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    N, n, m, batch = 32, 12, 4, 256
    lo, hi = shard.shard_range(batch * world, rank, world)
    pr = problems.random_ltv(batch * world, N, n, m)
    bt = altro_amd.Batch(N, n, m, batch, device=local_rank)
    sl = slice(lo, hi)
    bt.set_dynamics(pr["A"][sl], pr["B"][sl], pr["f"][sl])
    bt.set_cost(pr["Q"][sl], pr["R"][sl], pr["H"][sl], pr["q"][sl], pr["r"][sl])
    bt.set_initial_state(pr["x0"][sl])
    bt.sweep()
    before = bt.get("K").copy()
    st = bt.stats()
    # collectives on the device, interleaved with sweeps on the handle's own stream
    t = torch.tensor([float(st.problems), float(st.sum_delta_V0)], dtype=torch.float64, device="cuda")
    for _ in range(3):
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        bt.sweep()
        dist.barrier()
    torch.cuda.synchronize()
    assert t[0].item() == batch and t[1].item() == st.sum_delta_V0
    red = shard.reduce_stats(st, device="cuda")
    assert red["problems"] == batch * world and red["cholesky_failures"] == 0
    # the C ABI's own collective: a communicator built from an id that travelled over torch.distributed, two
    # ncclAllReduce calls on the handle's stream (altro_hip_stats_allreduce)
    comm = shard.make_comm(local_rank, rank, world)
    glob = bt.stats(comm)
    assert glob.problems == batch * world and glob.cholesky_failures == 0
    assert abs(glob.sum_delta_V0 - red["sum_delta_V0"]) <= 1e-12 * abs(red["sum_delta_V0"])
    assert glob.max_abs_xN == red["max_abs_xN"]
    comm.close()
    assert shard.max_over_ranks(1.5 + rank, device="cuda") == 1.5 + (world - 1)
    assert np.array_equal(bt.get("K"), before)
    bt.close()
    dist.destroy_process_group()
    if rank == 0:
        print("rccl rank check OK")


if __name__ == "__main__":
    main()
