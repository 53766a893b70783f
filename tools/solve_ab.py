#!/usr/bin/env python3
"""Fused vs launch-sequenced batched solve on the C2 / C3 batches, alternating in ONE process (same box, same
buffers): wall time of altro_hip_ilqr_solve, min / median over the repeats.    python tools/solve_ab.py [repeats]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import altro_amd                      # noqa: E402
from tests import problems            # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 and __name__ == "__main__" else 7


def c2(backtracking=False):
    N, batch = 100, 8192
    bt = altro_amd.Batch(N, 2, 1, batch)
    bt.set_model(altro_amd.MODEL_PENDULUM, np.float32(0.03))
    xf = np.array([np.pi, 0.0])
    bt.set_tracking_cost(np.array([[1e-2, 1e-2], [1.0, 1.0]]), np.array([[1e-3]]), np.stack([xf, xf]), np.zeros((1, 1)),
                         k_stride_zero=True, batch_stride_zero=True)
    x0 = np.zeros((batch, 2)); x0[:, 0] = problems.uniform01((batch,), 22) - 0.5
    bt.set_initial_state(x0)
    guess = lambda: bt.set_input_guess(np.array([[[0.1]]]), k_stride_zero=True, batch_stride_zero=True)   # noqa: E731
    return bt, guess, dict(iterations_max=80, use_backtracking=backtracking)


def c3(backtracking=True, N=50, batch=8192):
    n, m = 4, 2
    x_ref, u_ref = problems.bicycle_reference(N + 1)
    bt = altro_amd.Batch(N, n, m, batch)
    bt.set_model(altro_amd.MODEL_BICYCLE, np.float32(0.1))
    bt.set_tracking_cost(np.full((1, N + 1, n), 1e-2), np.full((1, N, m), 1e-3), x_ref[None, :N + 1], u_ref[None, :N],
                         batch_stride_zero=True)
    G = np.zeros((2, n + m)); G[0, 3] = 1.0; G[1, 3] = -1.0
    bt.add_linear_constraint(0, N, altro_amd.CONE_INEQUALITY, G, np.full(2, np.pi / 3))
    bt.set_initial_state(x_ref[0] + (problems.uniform01((batch, n), 23, 0) - 0.5) * 0.4)

    def guess():
        bt.reset_duals(1.0)
        bt.set_input_guess(np.array([[[u_ref[0][0], 0.0]]]), k_stride_zero=True, batch_stride_zero=True)
    return bt, guess, dict(iterations_max=80, use_backtracking=backtracking)


if __name__ == "__main__":
    CASES = (("C2 pendulum N=100 batch=8192, cubic search", c2, False), ("C2 pendulum, backtracking search", c2, True),
             ("C3 bicycle N=50 batch=8192, backtracking search", c3, True), ("C3 bicycle, cubic search", c3, False))
    for name, make, backtracking in CASES:
        bt, guess, opts = make(backtracking)
        times = {"fused": [], "sequenced": []}
        for rep in range(R + 1):
            for mode in ("fused", "sequenced"):
                os.environ["ALTRO_HIP_FUSED"] = "0" if mode == "sequenced" else "1"
                guess()
                bt.synchronize()
                t0 = time.perf_counter()
                res = bt.ilqr_solve(**opts)
                dt = time.perf_counter() - t0
                if rep:                      # the first pair loads the kernels' code objects
                    times[mode].append(dt)
        os.environ.pop("ALTRO_HIP_FUSED", None)
        for mode, ts in times.items():
            ts = np.sort(ts) * 1e3
            print("%-50s %-9s  min %.3f ms  median %.3f ms  max %.3f ms   (sweeps %d, converged %d)" % (
                name, mode, ts[0], ts[len(ts) // 2], ts[-1], int(res["sweeps"]), int((res["status"] == 0).sum())))
