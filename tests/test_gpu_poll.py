"""Results while the batch still solves (altro_hip_ilqr_solve_async / _poll / _wait): the one-launch solve
kernel publishes every problem into pinned host memory the moment it stops.  The records must be exactly what the blocking
solve returns, the first input must be the solution's u_0, and most of a straggler-tailed batch must be available long
before the launch ends.  Needs an MI355X."""
import time

import numpy as np
import pytest

import altro_amd
from tests import problems

pytestmark = pytest.mark.gpu


def _bicycles(batch, N=50):
    n, m = 4, 2
    x_ref, u_ref = problems.bicycle_reference(N + 1)
    bt = altro_amd.Batch(N, n, m, batch)
    bt.set_model(altro_amd.MODEL_BICYCLE, np.float32(0.1))
    bt.set_tracking_cost(np.full((1, N + 1, n), 1e-2), np.full((1, N, m), 1e-3), x_ref[None, :N + 1], u_ref[None, :N],
                         batch_stride_zero=True)
    G = np.zeros((2, n + m)); G[0, 3] = 1.0; G[1, 3] = -1.0
    bt.add_linear_constraint(0, N, altro_amd.CONE_INEQUALITY, G, np.full(2, np.pi / 3))
    bt.set_initial_state(x_ref[0] + (problems.uniform01((batch, n), 23, 0) - 0.5) * 0.4)
    bt.set_input_guess(np.array([[[u_ref[0][0], 0.0]]]), k_stride_zero=True, batch_stride_zero=True)
    return bt


def test_polled_records_equal_the_blocking_solve():
    batch = 1500
    a, b = _bicycles(batch), _bicycles(batch)
    ref = a.ilqr_solve(iterations_max=80, use_backtracking=True)
    u0_ref = a.get_knot(0)[1]
    b.ilqr_solve_async(iterations_max=80, use_backtracking=True)
    seen = []
    while True:
        n_done, rec = b.poll()
        seen.append(n_done)
        if n_done == batch:
            break
        time.sleep(0.0005)
    assert all(x <= y for x, y in zip(seen, seen[1:]))          # the count only grows
    assert (rec["done"] == 1).all()
    res = b.wait()
    for k, f in (("status", "status"), ("iterations", "iterations"), ("stationarity", "stationarity"), ("alpha", "final_alpha"),
                 ("phi", "final_phi"), ("feasibility", "primal_feasibility"), ("penalty", "penalty"), ("dual_updates", "dual_updates")):
        assert np.array_equal(rec["result"][f], ref[k]), k       # published == blocking solve, bit for bit
        assert np.array_equal(res[k], ref[k]), k                 # and so is what wait() returns
    assert np.array_equal(rec["u0"][:, :2], u0_ref) and (rec["u0"][:, 2:] == 0).all()
    assert np.array_equal(b.get_nominal()[0], a.get_nominal()[0])
    a.close(); b.close()


def test_most_of_a_straggler_batch_is_there_early():
    """8192 steering-bounded bicycles: the mean is ~3 sweeps, a few problems run all 80 (bench.py --config c3's solve).  Time to
    99 % of the records against the time to the last one."""
    batch = 8192
    bt = _bicycles(batch)
    bt.ilqr_solve(iterations_max=80, use_backtracking=True)      # (first launch of a process loads the code object)
    bt.reset_duals(1.0)
    x_ref, u_ref = problems.bicycle_reference(51)
    bt.set_input_guess(np.array([[[u_ref[0][0], 0.0]]]), k_stride_zero=True, batch_stride_zero=True)
    bt.synchronize()
    t0 = time.perf_counter()
    bt.ilqr_solve_async(iterations_max=80, use_backtracking=True)
    t99 = None
    while True:
        n_done, rec = bt.poll()
        if t99 is None and n_done >= 0.99 * batch:
            t99 = time.perf_counter() - t0
        if n_done == batch:
            t_all = time.perf_counter() - t0
            break
    res = bt.wait()
    print("poll: 99 %% of %d problems after %.2f ms, all after %.2f ms (%d sweeps)" % (batch, t99 * 1e3, t_all * 1e3, res["sweeps"]))
    assert (res["status"] == 0).sum() > 0.99 * batch
    assert t99 < 0.6 * t_all
    bt.close()


def test_async_needs_the_one_launch_path():
    p = problems.ilqr12x4_problem(8, 12, True)
    bt = altro_amd.Batch(12, 12, 4, 8)
    bt.set_dynamics(p["A"], p["B"], p["f"]); bt.set_tracking_cost(p["Qd"], p["Rd"], p["xref"], p["uref"])
    bt.set_initial_state(p["x0"]); bt.set_input_guess(p["u0"])
    with pytest.raises(altro_amd.AltroHipError):
        bt.ilqr_solve_async(iterations_max=5)
    res = bt.ilqr_solve(iterations_max=5)                        # the handle is still usable
    assert (res["status"] == 0).all()
    bt.close()
