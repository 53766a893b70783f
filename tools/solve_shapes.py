#!/usr/bin/env python3
"""Whole iLQR solves of random LTV tracking problems at a given (n, m) on the plan ALTRO_HIP_PLAN_AUTO picks -- for rocprofv3
--kernel-trace: which kernels a solve's time goes to past the (12, 4) tile (plan MFMA32's sweeps + plan GENERIC's loop kernels).

    python tools/solve_shapes.py n m [batch] [horizon] [solves] [--bounds]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import altro_amd  # noqa: E402
from tests import problems  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n, m = int(args[0]), int(args[1])
    batch = int(args[2]) if len(args) > 2 else 4096
    N = int(args[3]) if len(args) > 3 else 128
    solves = int(args[4]) if len(args) > 4 else 3
    p = problems.ilqr12x4_problem(batch, N, True, n=n, m=m)
    bt = altro_amd.Batch(N, n, m, batch)
    bt.set_dynamics(p["A"], p["B"], p["f"])
    bt.set_tracking_cost(p["Qd"], p["Rd"], p["xref"], p["uref"])
    bt.set_initial_state(p["x0"])
    if "--bounds" in sys.argv:
        G = np.zeros((2 * m, n + m)); G[:m, n:] = np.eye(m); G[m:, n:] = -np.eye(m)
        bt.add_linear_constraint(0, N - 1, altro_amd.CONE_INEQUALITY, G, np.full(2 * m, 0.5))
    ts = []
    for i in range(solves + 1):
        bt.set_input_guess(p["u0"])
        if "--bounds" in sys.argv:
            bt.reset_duals(1.0)
        bt.synchronize()
        t0 = time.perf_counter()
        res = bt.ilqr_solve(iterations_max=40)
        bt.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts = sorted(ts[1:])
    print("(%d, %d) x %d problems, N = %d, plan %d%s: solve median %.2f ms; sweeps %d, merit launches %d, converged %d"
          % (n, m, batch, N, bt.plan, " + input bounds" if "--bounds" in sys.argv else "", ts[len(ts) // 2], res["sweeps"], res["merit_launches"],
             int((res["status"] == 0).sum())))
    bt.close()


if __name__ == "__main__":
    main()
