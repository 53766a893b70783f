"""Batched receding-horizon MPC on the device: the caller pattern of
test/bicycle_test.cpp:266-337 -- solve, take u_0, move the reference (UpdateLinearCosts), SetInitialState,
ShiftTrajectory -- for a whole batch of vehicles at once, against the CPU oracle running each vehicle alone."""
import numpy as np
import pytest

import altro_amd
from tests import mpc_common as M
from tests import problems

pytestmark = pytest.mark.gpu


def make_hip(x_ref, u_ref, x0s):
    batch = x0s.shape[0]
    bt = altro_amd.Batch(M.N, M.n, M.m, batch)
    assert bt.plan == altro_amd.PLAN_LANE
    bt.set_model(altro_amd.MODEL_BICYCLE, M.H)
    Qd = np.full((1, M.N + 1, M.n), M.QD); Rd = np.full((1, M.N, M.m), M.RD)
    bt.set_tracking_cost(Qd, Rd, x_ref[None, :M.N + 1], u_ref[None, :M.N], batch_stride_zero=True)
    cone, G, g = M.steering_block()
    bt.add_linear_constraint(0, M.N, cone, G, g)
    bt.set_initial_state(x0s)
    u0 = np.array([u_ref[0][0], 0.0])
    bt.set_input_guess(u0[None, None], k_stride_zero=True, batch_stride_zero=True)
    return bt, u0


def test_shift_and_linear_cost_update_on_device():
    x_ref, u_ref = problems.bicycle_reference(M.N + 3)
    batch = 67
    x0s = np.tile(x_ref[0], (batch, 1)) + 0.01 * np.arange(batch)[:, None]
    bt, u0 = make_hip(x_ref, u_ref, x0s)
    bt.open_loop_rollout()
    xc0, uc0 = bt.get("x"), bt.get("u")
    bt.shift_trajectory()
    xc, uc = bt.get("x"), bt.get("u")
    assert np.array_equal(xc[:, :M.N], xc0[:, 1:]) and np.array_equal(xc[:, M.N], xc0[:, M.N])
    assert np.array_equal(uc[:, :M.N - 1], uc0[:, 1:]) and np.array_equal(uc[:, M.N - 1], uc0[:, M.N - 1])
    # UpdateLinearCosts: phi(0) after the update == the oracle's cost with the same update
    q, c = M.linear_costs(x_ref, 1, u0)
    bt.update_linear_costs(q[None], None, c[None], 0, M.N, batch_stride_zero=True)
    with pytest.raises(altro_amd.AltroHipError):
        bt.update_linear_costs(q[None], np.zeros((1, M.N + 1, M.m)), c[None], 0, M.N, batch_stride_zero=True)   # r at k = N
    with pytest.raises(altro_amd.AltroHipError):
        bt.update_linear_costs(q[None], None, c[None], 0, M.N + 1, batch_stride_zero=True)                       # BadIndex
    bt.accept(); bt.expand(); bt.backward()
    phi, _ = bt.merit(0.0)
    for b in [0, 66]:
        s, _ = M.make_oracle(x_ref, u_ref, x0s[b])
        s.L.oracle_ilqr_open_loop_rollout(s.h)
        s.L.oracle_ilqr_shift_trajectory(s.h)
        for k in range(M.N + 1):
            s.L.oracle_ilqr_update_linear_costs(s.h, k, q[k].ctypes.data, None, float(c[k]))
        np.testing.assert_allclose(xc[b], s.get("x_cand"), rtol=1e-12, atol=1e-12)
        s.L.oracle_ilqr_copy_trajectory(s.h)
        s.L.oracle_ilqr_calc_cost(s.h)   # CalcCost refreshes constraint values and projected duals (solver.cpp:424)
        s.L.oracle_ilqr_calc_dynamics_expansions(s.h); s.L.oracle_ilqr_calc_cost_gradient(s.h)
        s.L.oracle_ilqr_calc_expansions(s.h)
        s.L.oracle_ilqr_backward_pass(s.h)
        p_ref, _ = s.merit(0.0)
        # merit(0) re-rolls the shifted inputs from x0: same closed-loop trajectory, same cost
        assert abs(phi[b] - p_ref) <= 1e-9 * max(1.0, abs(p_ref))


def test_batched_bicycle_mpc_matches_per_vehicle_oracle():
    nsim = 8
    batch = 70
    x_ref, u_ref = problems.bicycle_reference(M.N + nsim + 1)
    off = problems.uniform01((batch, 4), 33) - 0.5
    x0s = x_ref[0] + off * np.array([0.2, 0.2, 0.04, 0.0])
    bt, u0 = make_hip(x_ref, u_ref, x0s)
    # SetState(x_ref) of the fixture is overwritten by Solve's initial rollout, so it is not needed here
    xs = x0s.copy()
    hist_iters, hist_u, hist_x = [], [], []
    for it in range(nsim):
        res = bt.ilqr_solve(iterations_max=80, use_backtracking=True)
        assert (res["status"] == 0).all(), (it, res["status"])
        _, u = bt.get_knot(0)
        xs = np.stack([M.plant(xs[b], u[b]) for b in range(batch)])
        hist_iters.append(res["iterations"].copy()); hist_u.append(u.copy()); hist_x.append(xs.copy())
        q, c = M.linear_costs(x_ref, it + 1, u0)
        bt.update_linear_costs(q[None], None, c[None], 0, M.N, batch_stride_zero=True)
        bt.set_initial_state(xs)
        bt.shift_trajectory()
    err0 = np.linalg.norm(hist_x[0] - x_ref[1], axis=1)
    errN = np.linalg.norm(hist_x[-1] - x_ref[nsim], axis=1)
    assert (errN < np.maximum(err0, 1e-3)).all()
    same_iters = 0
    for b in [0, 13, 69]:
        steps = M.oracle_mpc(x_ref, u_ref, x0s[b], nsim)
        for it, (iters, u, xn, status) in enumerate(steps):
            assert status == 0
            same_iters += int(iters == hist_iters[it][b])
            np.testing.assert_allclose(hist_u[it][b], u, rtol=0, atol=5e-5)
            np.testing.assert_allclose(hist_x[it][b], xn, rtol=0, atol=5e-5)
    assert same_iters >= 3 * nsim - 2   # sin/cos last-ulp differences may move a convergence test by one sweep
