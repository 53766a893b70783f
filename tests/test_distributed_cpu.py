"""N > 1 path on CPU (gloo, world_size 2): batch sharding regenerates exactly the global problem set
with no communication, and the statistics reduction -- the path's only collective -- matches a
serial reduction.  (On the GPU the same functions run under the "nccl" = RCCL backend: bench.py.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from altro_amd import shard
from tests import problems


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _Stats:
    def __init__(self, x0):
        self.problems = x0.shape[0]
        self.cholesky_failures = int((x0[:, 0] > 0.9).sum())
        self.sum_delta_V0 = float(x0[:, 1].sum())
        self.sum_delta_V1 = float((x0[:, 2] ** 2).sum())
        self.max_abs_xN = float(np.abs(x0).max())
        # the quantities SolverImpl::Solve reports, synthesised from the same slice
        self.converged = int((x0[:, 3] > 0).sum())
        self.iterations = int((np.abs(x0[:, 4]) * 10).astype(int).sum())
        self.sum_cost = float((x0[:, 5] ** 2).sum())
        self.max_stationarity = float(np.abs(x0[:, 6]).max())
        self.max_feasibility = float(np.abs(x0[:, 7]).max())


def _worker(rank, world, port, global_batch, n, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.shard_range(global_batch, rank, world)
    x0 = 2.0 * problems.uniform01((hi - lo, n), 21, lo * n) - 1.0     # bench.py's per-rank generation
    red = shard.reduce_stats(_Stats(x0))
    tmax = shard.max_over_ranks(1.0 + rank)
    gathered = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(gathered, torch.tensor([x0.sum()], dtype=torch.float64))
    out.put((rank, lo, hi, red, tmax, [g.item() for g in gathered], x0))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_sharding_and_stats_reduction_gloo():
    world, global_batch, n = 2, 1001, 12          # ragged: 501 + 500
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, global_batch, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=100) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert [(r[1], r[2]) for r in res] == [(0, 501), (501, 1001)]
    full = 2.0 * problems.uniform01((global_batch, n), 21) - 1.0
    assert np.array_equal(np.concatenate([r[6] for r in res]), full)          # shards tile the global set
    serial = _Stats(full)
    for r in res:
        red = r[3]
        assert red["problems"] == global_batch
        assert red["cholesky_failures"] == serial.cholesky_failures
        assert abs(red["sum_delta_V0"] - serial.sum_delta_V0) < 1e-9
        assert abs(red["sum_delta_V1"] - serial.sum_delta_V1) < 1e-9
        assert red["max_abs_xN"] == serial.max_abs_xN
        assert red["converged"] == serial.converged and red["iterations"] == serial.iterations
        assert abs(red["sum_cost"] - serial.sum_cost) < 1e-9
        assert red["max_stationarity"] == serial.max_stationarity
        assert red["max_feasibility"] == serial.max_feasibility
        assert r[4] == 2.0                                                       # max over ranks


def test_shard_range_covers_everything():
    for world in (1, 2, 3, 4, 8):
        for gb in (1, 7, 8, 4096, 65536 + 3):
            spans = [shard.shard_range(gb, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


# ---- shard.make_comm: either ALL ranks hold a communicator or all of them raise ---------------------------------
class _FakeComm:
    """Stands in for altro_amd.Comm (an RCCL communicator rank) on a box without GPUs: records what make_comm hands it."""
    fail_id = False          # rank 0 cannot draw a unique id (librccl missing, ...)
    fail_init_on = None      # this rank's ncclCommInitRank fails (wrong device, ...)
    closed = 0

    @staticmethod
    def unique_id():
        if _FakeComm.fail_id:
            raise RuntimeError("librccl.so not found")
        return bytes(range(128))

    def __init__(self, device, rank, world, uid):
        if _FakeComm.fail_init_on == rank:
            raise RuntimeError("ncclCommInitRank: invalid device on rank %d" % rank)
        self.device, self.rank, self.world, self.uid = device, rank, world, uid

    def close(self):
        _FakeComm.closed += 1


def _comm_worker(rank, world, port, scenario, out):
    import altro_amd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    altro_amd.Comm = _FakeComm
    _FakeComm.fail_id = scenario == "id"
    _FakeComm.fail_init_on = 1 if scenario == "init" else None
    try:
        c = shard.make_comm(rank, rank, world)
        out.put((rank, "ok", c.world, c.uid == bytes(range(128)), _FakeComm.closed))
    except altro_amd.AltroHipError as e:
        out.put((rank, "raised", str(e), None, _FakeComm.closed))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("scenario", ["fine", "id", "init"])
def test_make_comm_all_ranks_agree(scenario):
    """The unique-id broadcast and the agreement after ncclCommInitRank of shard.make_comm over gloo, world 2, with a stand-in for
    the RCCL communicator: every rank gets rank 0's 128 bytes; when rank 0 cannot draw an id, or ONE rank's init fails, BOTH ranks
    raise (and the rank whose init succeeded closes its communicator) -- nobody is left inside a collective the other never joins."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_comm_worker, args=(r, world, port, scenario, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=150) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    if scenario == "fine":
        assert [r[1] for r in res] == ["ok", "ok"] and all(r[2] == world and r[3] for r in res)
    else:
        assert [r[1] for r in res] == ["raised", "raised"], res
        if scenario == "id":
            assert all("unique id" in r[2] for r in res)
        else:
            assert all("ncclCommInitRank failed on at least one rank" in r[2] for r in res)
            assert res[0][4] == 1 and res[1][4] == 0        # rank 0's communicator (which did come up) was closed again
