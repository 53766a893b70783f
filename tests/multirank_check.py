"""Helper of tests/test_gpu_multirank.py: ONE RANK of a world-size-2 run on a one-GPU box, with REAL handles.

Launched as `python -m torch.distributed.run --nproc-per-node 2 tests/multirank_check.py` with the gloo backend (RCCL
refuses two ranks on one device; the 8-GPU RCCL runs belong to the driver).  Every rank owns its contiguous shard of a
global batch of steering-bounded bicycle tracking problems (BASELINE.json configs[3] at reduced size, ragged split),
solves it on the GPU through the C ABI, reduces its statistics on the device (altro_hip_stats_reduce) and all-reduces
the two vectors exactly as bench.py's multi-rank path does.  Rank 0 then solves the WHOLE global batch in one handle
and checks that (a) the reduced vector equals the single-process one -- the quantities SolverImpl::Solve reports,
solver.cpp:464-469, 492-509 -- and (b) every problem's result is the same bit for bit wherever it was solved."""
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

N, n, m = 50, 4, 2
GLOBAL = int(os.environ.get("ALTRO_MULTIRANK_GLOBAL", "601"))


def solve_range(lo, hi, device):
    """The problems [lo, hi) of the global batch on one handle: bench.py --config c3's set-up."""
    batch = hi - lo
    x_ref, u_ref = problems.bicycle_reference(N + 1)
    bt = altro_amd.Batch(N, n, m, batch, device=device)
    bt.set_model(altro_amd.MODEL_BICYCLE, np.float32(0.1))
    bt.set_tracking_cost(np.full((1, N + 1, n), 1e-2), np.full((1, N, m), 1e-3), x_ref[None, :N + 1], u_ref[None, :N],
                         batch_stride_zero=True)
    G = np.zeros((2, n + m)); G[0, 3] = 1.0; G[1, 3] = -1.0
    bt.add_linear_constraint(0, N, altro_amd.CONE_INEQUALITY, G, np.full(2, np.pi / 3))
    x0 = x_ref[0] + (problems.uniform01((batch, n), 23, lo * n) - 0.5) * 0.4     # counter-based: the global stream's slice
    bt.set_initial_state(x0)
    bt.set_input_guess(np.array([[[u_ref[0][0], 0.0]]]), k_stride_zero=True, batch_stride_zero=True)
    res = bt.ilqr_solve(iterations_max=40, use_backtracking=True)
    return bt, res


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    device = int(os.environ["LOCAL_RANK"]) % torch.cuda.device_count()
    torch.cuda.set_device(device)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.shard_range(GLOBAL, rank, world)
    bt, res = solve_range(lo, hi, device)
    local = bt.stats()
    red = shard.reduce_stats(local, device="cpu")           # 2 all-reduces: sums, maxima
    mine = {k: res[k] for k in ("status", "iterations", "phi", "stationarity", "feasibility", "alpha")}
    mine["xN"] = bt.get_knot(N, want_u=False)[0]
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, hi, mine, local.as_dict()))
    tmax = shard.max_over_ranks(1.0 + rank, device="cpu")
    assert tmax == float(world)
    bt.close()
    if rank == 0:
        full, fres = solve_range(0, GLOBAL, device)
        ser = full.stats().as_dict()
        xN = full.get_knot(N, want_u=False)[0]
        assert sorted((g[0], g[1]) for g in gathered) == [shard.shard_range(GLOBAL, r, world) for r in range(world)]
        for lo_r, hi_r, part, _ in gathered:           # (b) a problem does not care which rank / lane it rode
            for k in ("status", "iterations", "phi", "stationarity", "feasibility", "alpha"):
                assert np.array_equal(part[k], fres[k][lo_r:hi_r]), (k, lo_r)
            assert np.array_equal(part["xN"], xN[lo_r:hi_r])
        # (a) the reduced vector against the single-process one
        assert red["problems"] == ser["problems"] == GLOBAL
        for k in ("cholesky_failures", "converged", "iterations", "non_finite"):
            assert red[k] == ser[k], (k, red[k], ser[k])
        assert ser["converged"] > 0 and ser["iterations"] > GLOBAL        # a real solve happened
        for k in ("sum_cost", "sum_delta_V0", "sum_delta_V1"):             # same addends, a different tree
            assert abs(red[k] - ser[k]) <= 1e-12 * max(1.0, abs(ser[k])), (k, red[k], ser[k])
        for k in ("max_stationarity", "max_feasibility", "max_abs_xN"):
            assert red[k] == ser[k], (k, red[k], ser[k])
        # and against numpy over the gathered per-problem results
        it = np.concatenate([g[2]["iterations"] for g in sorted(gathered, key=lambda g: g[0])])
        assert red["iterations"] == int(it.sum())
        assert sum(g[3]["problems"] for g in gathered) == GLOBAL
        full.close()
        print("multirank check OK: world %d, global batch %d, converged %d, iterations %d, max feasibility %.3e"
              % (world, GLOBAL, red["converged"], red["iterations"], red["max_feasibility"]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
