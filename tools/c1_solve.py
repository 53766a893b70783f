#!/usr/bin/env python3
"""C1 (BASELINE.json configs[1]: N = 256, n = 12, m = 4, 4096 problems) as whole iLQR solves on plan MFMA16: wall time per
altro_hip_ilqr_solve (host clock, stream drained), for A/B runs (ALTRO_HIP_MERIT2=0/1) and for rocprofv3 --kernel-trace.

    python tools/c1_solve.py [solves] [batch] [horizon] [--al] [--boxes] [--generic] [--alternate=ENV_SWITCH]

--boxes: the input box of --al AND a state box |x| <= 1.2 (24 rows) at every running knot point -- four of plan MFMA16's six slots
(kernels/al_types.h); --generic: the same problem on plan GENERIC (where it had to live up to round 5).
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
    solves = int(args[0]) if len(args) > 0 else 9
    batch = int(args[1]) if len(args) > 1 else 4096
    N = int(args[2]) if len(args) > 2 else 256
    boxes = "--boxes" in sys.argv
    soc = "--soc" in sys.argv                     # ||u[:3]|| <= 2 as a second-order cone instead of the input box (+ the state box with --boxes)
    al = "--al" in sys.argv or boxes or soc
    n, m = 12, 4
    one = problems.c1_double_integrator(1, N=N)
    bt = altro_amd.Batch(N, n, m, batch, plan=altro_amd.PLAN_GENERIC if "--generic" in sys.argv else altro_amd.PLAN_AUTO)
    bt.set_dynamics(one["A"][0, :1], one["B"][0, :1], None, k_stride_zero=True, batch_stride_zero=True)
    bt.set_tracking_cost(np.stack([np.ones(n), 100.0 * np.ones(n)]), np.full((1, m), 1e-2), np.zeros((2, n)), np.zeros((1, m)),
                         k_stride_zero=True, batch_stride_zero=True)
    bt.set_initial_state(2.0 * problems.uniform01((batch, n), 21) - 1.0)
    if soc:
        Gs = np.zeros((4, n + m)); Gs[0, n] = Gs[1, n + 1] = Gs[2, n + 2] = 1.0
        bt.add_linear_constraint(0, N - 1, altro_amd.CONE_SOC, Gs, np.array([0.0, 0.0, 0.0, -2.0]))
    elif al:   # input bounds as an INEQUALITY block: the AL path of the same shape
        G = np.zeros((2 * m, n + m)); G[:m, n:] = np.eye(m); G[m:, n:] = -np.eye(m)
        bt.add_linear_constraint(0, N - 1, altro_amd.CONE_INEQUALITY, G, np.full(2 * m, 2.0))
    if boxes:
        Gx = np.zeros((2 * n, n + m)); Gx[:n, :n] = np.eye(n); Gx[n:, :n] = -np.eye(n)
        bt.add_linear_constraint(int(os.environ.get("C1_XBOX_K0", "0")), N - 1, altro_amd.CONE_INEQUALITY, Gx, np.full(2 * n, float(os.environ.get("C1_XBOX", "1.2"))))
    alternate = [a.split("=")[1] for a in sys.argv if a.startswith("--alternate=")]   # an environment switch flipped 1 / 0 per solve
    ts = []
    for i in range(solves + 1):
        if alternate:
            os.environ[alternate[0]] = str(i & 1)
        bt.set_input_guess(np.zeros((1, 1, m)), k_stride_zero=True, batch_stride_zero=True)
        if al:
            bt.reset_duals(1.0)
        bt.synchronize()
        t0 = time.perf_counter()
        res = bt.ilqr_solve(iterations_max=40 if al else 10)
        bt.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts = sorted(ts[1:])
    print("C1 solve%s, %d problems, N = %d, ALTRO_HIP_MERIT2=%s: median %.3f ms (min %.3f, max %.3f) over %d solves; sweeps %d, "
          "merit launches %d, converged %d, max stationarity %.2e"
          % ((" + %s + state box, plan %d" % ("input cone" if soc else "input box", bt.plan)) if boxes else " + input cone" if soc else " + input bounds" if al else "", batch, N, os.environ.get("ALTRO_HIP_MERIT2", "unset"), ts[len(ts) // 2], ts[0], ts[-1],
             len(ts), res["sweeps"], res["merit_launches"], int((res["status"] == 0).sum()), float(np.abs(res["stationarity"]).max())))
    bt.close()


if __name__ == "__main__":
    main()
