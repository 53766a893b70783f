"""Statistics reduction: the device-side reduction of altro_hip_stats_reduce against a numpy
reduction of the per-problem results, on every plan, and the RCCL all-reduce entry points of the C ABI
(altro_hip_comm_*, altro_hip_stats_allreduce, altro_hip_stats_allreduce_multi) on one GPU: a one-rank communicator
still goes through ncclCommInitRank / ncclAllReduce on the handle's stream.  Needs an MI355X."""
import ctypes as C

import os

import numpy as np
import pytest

import altro_amd
from tests import problems

pytestmark = pytest.mark.gpu


def _ltv_batch(plan, batch, N, n, m, dtype=altro_amd.F64, break_some=False):
    pr = problems.random_ltv(batch, N, n, m)
    if break_some:   # an indefinite R makes the Cholesky of a few problems fail
        pr["R"] = pr["R"].copy()
        pr["R"][::7] *= -1.0
    bt = altro_amd.Batch(N, n, m, batch, plan=plan, dtype=dtype)
    bt.set_dynamics(pr["A"], pr["B"], pr["f"])
    bt.set_cost(pr["Q"], pr["R"], pr["H"], pr["q"], pr["r"])
    bt.set_initial_state(pr["x0"])
    return bt


@pytest.mark.parametrize("plan,n,m,batch", [(altro_amd.PLAN_MFMA16, 12, 4, 777), (altro_amd.PLAN_LANE, 4, 2, 1000),
                                            (altro_amd.PLAN_GENERIC, 5, 3, 300)])
def test_sweep_statistics_match_numpy(plan, n, m, batch):
    bt = _ltv_batch(plan, batch, 20, n, m, break_some=True)
    bt.sweep()
    st, dv, x = bt.get("status"), bt.get("delta_V"), bt.get("x")
    s = bt.stats()
    ok = st == -1
    assert s.problems == batch
    assert s.cholesky_failures == int((~ok).sum()) and s.cholesky_failures > 0
    assert abs(s.sum_delta_V0 - dv[ok, 0].sum()) <= 1e-12 * np.abs(dv[ok, 0]).sum()
    assert abs(s.sum_delta_V1 - dv[ok, 1].sum()) <= 1e-12 * np.abs(dv[ok, 1]).sum()
    assert s.max_abs_xN == np.abs(x[:, -1]).max()
    assert s.converged == 0 and s.iterations == 0 and s.sum_cost == 0.0     # no solve has run on this handle
    s2 = bt.stats()                                                          # deterministic: bit-identical again
    assert (s2.sum_delta_V0, s2.sum_delta_V1) == (s.sum_delta_V0, s.sum_delta_V1)
    bt.close()


def test_fp32_handle_statistics():
    bt = _ltv_batch(altro_amd.PLAN_MFMA16, 500, 16, 12, 4, dtype=altro_amd.F32)
    bt.sweep()
    dv, x = bt.get("delta_V"), bt.get("x")
    s = bt.stats()
    assert s.problems == 500 and s.cholesky_failures == 0
    assert abs(s.sum_delta_V0 - dv[:, 0].sum()) <= 1e-12 * np.abs(dv[:, 0]).sum()
    assert s.max_abs_xN == np.abs(x[:, -1]).max()
    bt.close()


def _pendulum_solved(batch):
    N, n, m = 40, 2, 1
    bt = altro_amd.Batch(N, n, m, batch)
    bt.set_model(altro_amd.MODEL_PENDULUM, np.float32(0.05))
    xf = np.array([np.pi, 0.0])
    bt.set_tracking_cost(np.array([[1e-2, 1e-2], [1.0, 1.0]]), np.array([[1e-3]]), np.stack([xf, xf]),
                         np.zeros((1, m)), k_stride_zero=True, batch_stride_zero=True)
    x0 = np.zeros((batch, n)); x0[:, 0] = problems.uniform01((batch,), 31) - 0.5
    bt.set_initial_state(x0)
    bt.set_input_guess(np.array([[[0.1]]]), k_stride_zero=True, batch_stride_zero=True)
    res = bt.ilqr_solve(iterations_max=30)
    return bt, res


def test_solve_statistics_match_the_per_problem_results():
    """{sum cost, sum iterations, #converged, #chol-fail; max stationarity, max feasibility} -- what
    SolverImpl::Solve reports (solver.cpp:464-469, 492-509) -- from IlqrProb on the device."""
    batch = 300
    bt, res = _pendulum_solved(batch)
    s = bt.stats()
    assert s.problems == batch
    assert s.converged == int((res["status"] == 0).sum()) and s.converged > 0
    assert s.iterations == int(res["iterations"].sum())
    assert abs(s.sum_cost - res["phi"].sum()) <= 1e-12 * np.abs(res["phi"]).sum()
    assert s.max_stationarity == np.abs(res["stationarity"]).max()
    assert s.max_feasibility == res["feasibility"].max()
    assert s.max_abs_xN == np.abs(bt.get("x")[:, -1]).max()
    bt.close()


def test_rccl_allreduce_entry_points_on_one_rank():
    """altro_hip_comm_create (ncclCommInitRank) + altro_hip_stats_allreduce (2 x ncclAllReduce on the handle's stream)
    with world = 1: the reduced vector equals the local one, and sweeps interleave with the collective."""
    import torch  # noqa: F401  (its librccl / HIP runtime are the ones this process shares)
    bt = _ltv_batch(altro_amd.PLAN_MFMA16, 512, 24, 12, 4)
    bt.sweep()
    local = bt.stats().as_dict()
    comm = altro_amd.Comm(0, 0, 1, altro_amd.Comm.unique_id())
    for _ in range(3):
        red = bt.stats(comm).as_dict()
        assert red == local
        bt.sweep()
    # the single-process multi-device form (ncclCommInitAll + grouped calls), with the one device this box has
    L = altro_amd.lib()
    cm = (C.c_void_p * 1)()
    assert L.altro_hip_comm_create_all(cm, 1, None) == 0, L.altro_hip_last_error()
    hs = (C.c_void_p * 1)(bt.h)
    out = altro_amd.Stats()
    assert L.altro_hip_stats_allreduce_multi(hs, cm, 1, C.byref(out)) == 0, L.altro_hip_last_error()
    assert out.as_dict() == local
    L.altro_hip_comm_destroy(cm[0])
    comm.close()
    bt.close()


def test_rccl_all_devices_of_the_box_in_one_process():
    """altro_hip_comm_create_all (ncclCommInitAll) + altro_hip_stats_allreduce_multi (grouped all-reduces: int64 counts, double sums,
    double maxima) over EVERY device this box has, one handle per device: the reduced vector is the sum / maximum of the local
    ones.  The first execution of the multi-rank RCCL code on a multi-GPU node must not be the driver's scaling run -- on the
    1-GPU test boxes this is skipped."""
    import torch  # noqa: F401
    L = altro_amd.lib()
    ndev = L.altro_hip_device_count()
    if ndev < 2:
        pytest.skip("one HIP device: the one-rank form is covered by test_rccl_allreduce_entry_points_on_one_rank")
    bts, locals_ = [], []
    for d in range(ndev):
        pr = problems.random_ltv(96 + 32 * d, 24, 12, 4, first=1000 * d)
        bt = altro_amd.Batch(24, 12, 4, 96 + 32 * d, device=d)
        bt.set_dynamics(pr["A"], pr["B"], pr["f"]); bt.set_cost(pr["Q"], pr["R"], pr["H"], pr["q"], pr["r"])
        bt.set_initial_state(pr["x0"]); bt.sweep()
        bts.append(bt); locals_.append(bt.stats().as_dict())
    cm = (C.c_void_p * ndev)()
    assert L.altro_hip_comm_create_all(cm, ndev, None) == 0, L.altro_hip_last_error()
    hs = (C.c_void_p * ndev)(*[bt.h for bt in bts])
    out = altro_amd.Stats()
    assert L.altro_hip_stats_allreduce_multi(hs, cm, ndev, C.byref(out)) == 0, L.altro_hip_last_error()
    red = out.as_dict()
    for k in ("problems", "cholesky_failures", "converged", "iterations", "non_finite"):
        assert red[k] == sum(l[k] for l in locals_), k
    for k in ("sum_cost", "sum_delta_V0", "sum_delta_V1"):
        ref = sum(l[k] for l in locals_)
        assert abs(red[k] - ref) <= 1e-12 * max(1.0, abs(ref)), k
    for k in ("max_stationarity", "max_feasibility", "max_abs_xN"):
        assert red[k] == max(l[k] for l in locals_), k
    for i in range(ndev):
        L.altro_hip_comm_destroy(cm[i])
    for bt in bts:
        bt.close()


def test_comm_create_gives_up_when_a_rank_never_joins():
    """ncclCommInitRank waits for every rank of the world; altro_hip_comm_create gives up after ALTRO_HIP_COMM_TIMEOUT_S with a
    message instead of hanging (here: rank 0 of a world of 2 whose rank 1 never starts).  Run in a child process: the helper
    thread stays blocked inside RCCL until the process ends."""
    import subprocess
    import sys
    code = ("import os, torch, altro_amd\n"
            "os.environ['ALTRO_HIP_COMM_TIMEOUT_S'] = '3'\n"
            "try:\n"
            "    altro_amd.Comm(0, 0, 2, altro_amd.Comm.unique_id())\n"
            "    print('JOINED', flush=True)\n"
            "except Exception as e:\n"
            "    print('ERR', e, flush=True)\n"
            "os._exit(0)\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert "ERR" in r.stdout and "did not return within 3 s" in r.stdout, r.stdout + r.stderr


def test_a_diverged_problem_shows_in_the_statistics():
    """fmax drops a NaN; the reduction must not: one problem with a NaN trajectory makes max_abs_xN NaN (as the
    per-problem results say) and is counted in non_finite -- the count is what survives an ncclMax across GPUs."""
    batch = 700
    pr = problems.random_ltv(batch, 12, 4, 2)
    x0 = pr["x0"].copy()
    x0[333, 1] = np.nan
    bt = altro_amd.Batch(12, 4, 2, batch)
    bt.set_dynamics(pr["A"], pr["B"], pr["f"])
    bt.set_cost(pr["Q"], pr["R"], pr["H"], pr["q"], pr["r"])
    bt.set_initial_state(x0)
    bt.sweep()
    s = bt.stats()
    assert np.isnan(bt.get("x")[333, -1]).any()
    assert np.isnan(s.max_abs_xN) and s.non_finite == 1 and s.problems == batch
    bt.set_initial_state(pr["x0"])
    bt.sweep()
    s = bt.stats()
    assert s.non_finite == 0 and s.max_abs_xN == np.abs(bt.get("x")[:, -1]).max()
    bt.close()
