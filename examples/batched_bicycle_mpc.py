#!/usr/bin/env python3
"""Batched receding-horizon MPC on one MI355X: `batch` kinematic bicycles track a reference path with a steering
bound, all solved together by the device AL-iLQR loop (the caller pattern of the reference's
test/bicycle_test.cpp:266-337, for many vehicles at once).

    python examples/batched_bicycle_mpc.py [batch] [steps]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import altro_amd  # noqa: E402


def reference_path(steps, h=0.1, v=6.3, L=2.7, lr=1.5):
    def f(x, u):
        beta = np.arctan2(lr * x[3], L)
        return np.array([u[0] * np.cos(x[2] + beta), u[0] * np.sin(x[2] + beta), u[0] * np.cos(beta) * np.tan(x[3]) / L, u[1]])
    x, xs, us = np.zeros(4), [], []
    for i in range(steps):
        u = np.array([v, 0.05 * np.sin(0.2 * i * h)])
        xs.append(x.copy()); us.append(u)
        x = x + h * f(x + 0.5 * h * f(x, u), u)
    xs.append(x.copy())
    return np.array(xs), np.array(us)


def plant(x, u, h=0.1, L=2.7, lr=1.5):
    """Vectorised explicit-midpoint step of the same model, standing in for the real vehicles."""
    def f(x, u):
        beta = np.arctan2(lr * x[:, 3], L)
        return np.stack([u[:, 0] * np.cos(x[:, 2] + beta), u[:, 0] * np.sin(x[:, 2] + beta),
                         u[:, 0] * np.cos(beta) * np.tan(x[:, 3]) / L, u[:, 1]], 1)
    return x + h * f(x + 0.5 * h * f(x, u), u)


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    N, n, m, h = 30, 4, 2, np.float32(0.1)
    Qd, Rd = 1e-2, 1e-3
    x_ref, u_ref = reference_path(N + steps + 1)
    rng = np.random.default_rng(0)
    x = x_ref[0] + rng.uniform(-0.3, 0.3, size=(batch, n)) * np.array([1.0, 1.0, 0.2, 0.0])

    bt = altro_amd.Batch(N, n, m, batch)
    bt.set_model(altro_amd.MODEL_BICYCLE, h)
    bt.set_tracking_cost(np.full((1, N + 1, n), Qd), np.full((1, N, m), Rd), x_ref[None, :N + 1], u_ref[None, :N],
                         batch_stride_zero=True)
    G = np.zeros((2, n + m)); G[0, 3] = 1.0; G[1, 3] = -1.0                      # |steering angle| <= 60 deg
    bt.add_linear_constraint(0, N, altro_amd.CONE_INEQUALITY, G, np.full(2, np.pi / 3))
    bt.set_initial_state(x)
    u0 = np.array([u_ref[0][0], 0.0])
    bt.set_input_guess(u0[None, None], k_stride_zero=True, batch_stride_zero=True)
    c_u = 0.5 * float(u0 @ (Rd * u0))

    t_solve = 0.0
    for t in range(steps):
        t0 = time.perf_counter()
        res = bt.ilqr_solve(iterations_max=20, use_backtracking=True)
        t_solve += time.perf_counter() - t0
        _, u = bt.get_knot(0)
        x = plant(x, u)
        err = np.linalg.norm(x - x_ref[t + 1], axis=1)
        if t % 10 == 0 or t == steps - 1:
            print("step %3d: %5d/%d converged, mean iterations %.2f, tracking error mean %.3e max %.3e"
                  % (t, int((res["status"] == 0).sum()), batch, res["iterations"].mean(), err.mean(), err.max()))
        xr = x_ref[t + 1:t + N + 2]                                             # the reference moves on: new linear terms
        q = -(Qd * xr)
        c = 0.5 * Qd * (xr * xr).sum(1) + np.where(np.arange(N + 1) < N, c_u, 0.0)
        bt.update_linear_costs(q[None], None, c[None], 0, N, batch_stride_zero=True)
        bt.set_initial_state(x)
        bt.shift_trajectory()
    print("%d MPC steps x %d vehicles: %.1f ms per step in the solver (%.0f vehicle-solves/s)"
          % (steps, batch, t_solve / steps * 1e3, steps * batch / t_solve))


if __name__ == "__main__":
    main()
