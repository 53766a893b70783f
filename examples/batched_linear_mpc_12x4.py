#!/usr/bin/env python3
"""Batched linear MPC at the quadrotor-sized shape (n, m) = (12, 4) on one MI355X: `batch` four-axis triple integrators
(position, velocity, acceleration per axis; jerk inputs) follow a moving set-point under input bounds.  Dynamics are
given as data (the reference's SetLinearDynamics path), so the whole AL-iLQR loop runs on the matrix-core plan MFMA16.

    python examples/batched_linear_mpc_12x4.py [batch] [steps] [stop-when-running-at-most]

The third argument is altro_hip_solve_options::stop_when_running_at_most (default 0 = every problem to its own end, like the
reference): a batched call lasts as long as its slowest problem, and here 99.9 % of the systems need one sweep per step.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import altro_amd  # noqa: E402


# the reference's MPC caller uses the backtracking line search (test/bicycle_test.cpp:291); BACKTRACK=0 selects the cubic one
BACKTRACK = bool(int(os.environ.get("BACKTRACK", "1")))


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    stop_at = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    N, n, m, h = 40, 12, 4, 0.05
    I4, Z4 = np.eye(4), np.zeros((4, 4))
    A = np.block([[I4, h * I4, 0.5 * h * h * I4], [Z4, I4, h * I4], [Z4, Z4, I4]])
    B = np.vstack([h ** 3 / 6 * I4, 0.5 * h * h * I4, h * I4])
    colmajor = lambda M: np.ascontiguousarray(M.T).reshape(1, 1, -1)   # noqa: E731  (reference layout: column-major blocks)
    Qd = np.concatenate([10.0 * np.ones(4), np.ones(4), 0.1 * np.ones(4)])
    Rd, umax = 1e-2 * np.ones(m), 6.0
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-1, 1, (batch, 4)), np.zeros((batch, 8))], axis=1)
    goal = lambda t: np.concatenate([np.array([np.sin(0.3 * t), np.cos(0.3 * t), 0.5 * np.sin(0.15 * t), 0.2 * t]), np.zeros(8)])  # noqa: E731

    bt = altro_amd.Batch(N, n, m, batch)
    assert bt.plan == altro_amd.PLAN_MFMA16
    bt.set_dynamics(colmajor(A), colmajor(B), None, k_stride_zero=True, batch_stride_zero=True)
    xref = np.stack([goal(k * h) for k in range(N + 1)])
    bt.set_tracking_cost(np.tile(Qd, (1, N + 1, 1)), np.tile(Rd, (1, N, 1)), xref[None], np.zeros((1, N, m)),
                         batch_stride_zero=True)
    G = np.zeros((2 * m, n + m)); G[:m, n:] = np.eye(m); G[m:, n:] = -np.eye(m)          # |u| <= umax
    bt.add_linear_constraint(0, N - 1, altro_amd.CONE_INEQUALITY, G, np.full(2 * m, umax))
    bt.set_initial_state(x)
    bt.set_input_guess(np.zeros((1, 1, m)), k_stride_zero=True, batch_stride_zero=True)

    t_solve = 0.0
    for t in range(steps):
        t0 = time.perf_counter()
        res = bt.ilqr_solve(iterations_max=60, use_backtracking=BACKTRACK, stop_when_running_at_most=stop_at)
        t_solve += time.perf_counter() - t0
        _, u = bt.get_knot(0)
        x = x @ A.T + u @ B.T                                                    # the plant: the same linear model
        if t % 10 == 0 or t == steps - 1:
            err = np.linalg.norm(x[:, :4] - goal((t + 1) * h)[:4], axis=1)
            print("step %3d: %5d/%d converged, iterations mean %.2f max %d (%d sweeps, %d merit launches), max |u| %.3f, position error mean %.3e"
                  % (t, int((res["status"] == 0).sum()), batch, res["iterations"].mean(), res["iterations"].max(), res["sweeps"],
                     res["merit_launches"], np.abs(u).max(), err.mean()))
        xr = np.stack([goal((t + 1 + k) * h) for k in range(N + 1)])             # the set-point moves on
        bt.update_linear_costs(-(Qd * xr)[None], None, (0.5 * (Qd * xr * xr).sum(1))[None], 0, N, batch_stride_zero=True)
        bt.set_initial_state(x)
        bt.shift_trajectory()
    print("%d MPC steps x %d systems (n=12, m=4, N=%d%s): %.2f ms per step in the solver (%.0f solves/s)"
          % (steps, batch, N, ", return when <= %d still run" % stop_at if stop_at else "", t_solve / steps * 1e3, steps * batch / t_solve))


if __name__ == "__main__":
    main()
