#!/usr/bin/env python3
"""Batched NONLINEAR MPC at (n, m) = (12, 4) on one MI355X: `batch` quadrotors (the 12-state rigid-body model of
altro_amd/csrc/models.h; thrust and three body torques) fly a moving set-point under thrust and torque bounds.  The dynamics
are a device model (altro_hip_set_model on plan MFMA16): every rollout, merit evaluation and expansion steps the model on the
GPU in the tile plan's row layout, the backward sweeps are the benchmarked matrix-core kernel.

    python examples/batched_quadrotor_nmpc.py [batch] [steps] [sweeps-per-step]

`sweeps-per-step` caps iLQR sweeps per MPC step (real-time iteration; default 3): a batch waits for its slowest problem, and a
receding-horizon controller re-plans 50 times a second anyway.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import altro_amd  # noqa: E402

MASS, GRAV, IX, IY, IZ = 0.5, 9.81, 0.0023, 0.0023, 0.004


def f_cont(x, u):
    """the model's continuous dynamics for the plant (the same equations, numpy, whole batch)"""
    sp, cp, st, ct, ss, cs = np.sin(x[:, 3]), np.cos(x[:, 3]), np.sin(x[:, 4]), np.cos(x[:, 4]), np.sin(x[:, 5]), np.cos(x[:, 5])
    tt = st / ct
    wx, wy, wz = x[:, 9], x[:, 10], x[:, 11]
    a = u[:, 0] / MASS
    return np.stack([x[:, 6], x[:, 7], x[:, 8],
                     wx + sp * tt * wy + cp * tt * wz, cp * wy - sp * wz, (sp * wy + cp * wz) / ct,
                     a * (cp * st * cs + sp * ss), a * (cp * st * ss - sp * cs), a * (cp * ct) - GRAV,
                     (u[:, 1] - (IZ - IY) * wy * wz) / IX, (u[:, 2] - (IX - IZ) * wz * wx) / IY, (u[:, 3] - (IY - IX) * wx * wy) / IZ], axis=1)


def plant(x, u, h):   # explicit midpoint, like the solver's discretisation (test/test_utils.cpp:84-132)
    return x + h * f_cont(x + 0.5 * h * f_cont(x, u), u)


def main():
    argv = [a for a in sys.argv if not a.startswith("--")]
    batch = int(argv[1]) if len(argv) > 1 else 4096
    steps = int(argv[2]) if len(argv) > 2 else 50
    sweeps = int(argv[3]) if len(argv) > 3 else 3
    from_source = "--source" in sys.argv          # the same model handed over as HIP source (altro_hip_set_model_source: hiprtc)
    N, n, m, h = 40, 12, 4, 0.02
    hover = np.array([MASS * GRAV, 0.0, 0.0, 0.0])
    Qd = np.concatenate([np.full(3, 4.0), np.full(3, 1.0), np.full(3, 0.5), np.full(3, 0.1)])
    Rd = np.array([0.05, 20.0, 20.0, 20.0])
    rng = np.random.default_rng(3)
    x = np.zeros((batch, n))
    x[:, :3] = rng.uniform(-1.0, 1.0, (batch, 3))
    x[:, 3:6] = rng.uniform(-0.2, 0.2, (batch, 3))
    x[:, 6:9] = rng.uniform(-0.5, 0.5, (batch, 3))
    goal = lambda t: np.concatenate([np.array([np.sin(0.8 * t), np.cos(0.8 * t) - 1.0, 0.3 * np.sin(0.4 * t)]), np.zeros(9)])  # noqa: E731

    bt = altro_amd.Batch(N, n, m, batch)
    assert bt.plan == altro_amd.PLAN_MFMA16
    if from_source:
        from tests.test_gpu_tile_model import QUADROTOR_SRC
        bt.set_model_source(QUADROTOR_SRC, np.float32(h))
    else:
        bt.set_model(altro_amd.MODEL_QUADROTOR, np.float32(h))
    xref = np.stack([goal(k * h) for k in range(N + 1)])
    Q = np.tile(Qd, (1, N + 1, 1)); Q[0, N] *= 10.0
    bt.set_tracking_cost(Q, np.tile(Rd, (1, N, 1)), xref[None], np.tile(hover, (1, N, 1)), batch_stride_zero=True)
    G = np.zeros((2 * m, n + m)); G[:m, n:] = np.eye(m); G[m:, n:] = -np.eye(m)
    bnd = np.array([3.0, 0.05, 0.05, 0.05])                       # |F - m g| <= 3 N, |tau| <= 0.05 N m
    bt.add_linear_constraint(0, N - 1, altro_amd.CONE_INEQUALITY, G[:m], hover + bnd)
    bt.add_linear_constraint(0, N - 1, altro_amd.CONE_INEQUALITY, G[m:], -(hover - bnd))
    bt.set_initial_state(x)
    bt.set_input_guess(hover[None, None], k_stride_zero=True, batch_stride_zero=True)

    t_solve = 0.0
    for t in range(steps):
        t0 = time.perf_counter()
        res = bt.ilqr_solve(iterations_max=(20 if t == 0 else sweeps), tol_stationarity=1e-3, use_backtracking=True)
        dt = time.perf_counter() - t0
        if t > 0:
            t_solve += dt
        _, u = bt.get_knot(0)
        x = plant(x, u, h)
        if t % 10 == 0 or t == steps - 1:
            err = np.linalg.norm(x[:, :3] - goal((t + 1) * h)[:3], axis=1)
            print("step %3d: %.2f ms, %5d/%d converged, sweeps %d, thrust in [%.2f, %.2f], max |tau| %.3f, position error mean %.3f max %.3f"
                  % (t, dt * 1e3, int((res["status"] == 0).sum()), batch, res["sweeps"], u[:, 0].min(), u[:, 0].max(), np.abs(u[:, 1:]).max(),
                     err.mean(), err.max()))
        xr = np.stack([goal((t + 1 + k) * h) for k in range(N + 1)])
        Qk = np.tile(Qd, (N + 1, 1)); Qk[N] *= 10.0
        bt.update_linear_costs(-(Qk * xr)[None], None, (0.5 * (Qk * xr * xr).sum(1))[None], 0, N, batch_stride_zero=True)
        bt.set_initial_state(x)
        bt.shift_trajectory()
    print("%d NMPC steps x %d quadrotors (n=12, m=4, N=%d, h=%.2f, <= %d sweeps per step): %.2f ms per step in the solver (%.0f solves/s)"
          % (steps - 1, batch, N, h, sweeps, t_solve / (steps - 1) * 1e3, (steps - 1) * batch / t_solve))


if __name__ == "__main__":
    main()
