#!/usr/bin/env python3
"""Soak: constrained (12, 4) AL-iLQR solves on plan MFMA16 against the oracle, EVERY problem of every batch, once with the affine
line-search rounds and once with every trial a rollout (ALTRO_HIP_AFFINE=0): how many problems end with the oracle's
status and iteration count in either form, and how far the converged ones are from the oracle's trajectory.

    python tests/soak/fuzz_tile_al_oracle.py [cases] [seed]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import altro_amd  # noqa: E402
from oracle import oracle  # noqa: E402
from tests import problems  # noqa: E402

n, m = 12, 4
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
tot = {"affine": [0, 0, 0.0], "rollout": [0, 0, 0.0]}
for it in range(cases):
    N = int(rng.integers(8, 70)); batch = int(rng.integers(4, 24))
    backtracking = bool(rng.random() < 0.4)
    p = problems.ilqr12x4_problem(batch, N, True)
    p["x0"] = p["x0"] * float(rng.uniform(0.5, 1.5))
    blocks = problems.ilqr12x4_constraint_blocks(N)
    ref = []
    for b in range(batch):
        s = oracle.ILQR(N, n, m, 0.01, oracle.DYN_LINEAR, cost_kind=oracle.COST_DIAGONAL)
        fb = np.ascontiguousarray(p["f"][b])
        s.L.oracle_ilqr_set_linear_dynamics(s.h, np.ascontiguousarray(p["A"][b]), np.ascontiguousarray(p["B"][b]), fb.ctypes.data)
        for k in range(N + 1):
            s.L.oracle_ilqr_set_lqr_cost(s.h, k, np.ascontiguousarray(p["Qd"][b, k]), np.ascontiguousarray(p["Rd"][b, min(k, N - 1)]),
                                         np.ascontiguousarray(p["xref"][b, k]), np.ascontiguousarray(p["uref"][b, min(k, N - 1)]))
        s.L.oracle_ilqr_set_initial_state(s.h, np.ascontiguousarray(p["x0"][b]))
        for (k0, k1, cone, G, g) in blocks:
            for k in range(k0, k1 + 1):
                s.add_linear_constraint(k, cone, G, g)
        s.L.oracle_ilqr_initialize(s.h)
        for k in range(N):
            s.L.oracle_ilqr_set_input(s.h, k, np.ascontiguousarray(p["u0"][b, k]))
        s.set_penalty(1.0, 10.0)
        s.L.oracle_ilqr_set_options(s.h, 60, 1e-4, 1e-4, 1e-8, 1 if backtracking else 0)
        status, iters, _ = s.solve()
        ref.append((status, iters, s.get("x").reshape(N + 1, n).copy()))
    line = []
    for name, env in (("affine", "1"), ("rollout", "0")):
        os.environ["ALTRO_HIP_AFFINE"] = env
        bt = altro_amd.Batch(N, n, m, batch)
        bt.set_dynamics(p["A"], p["B"], p["f"]); bt.set_tracking_cost(p["Qd"], p["Rd"], p["xref"], p["uref"])
        bt.set_initial_state(p["x0"]); bt.set_input_guess(p["u0"])
        for (k0, k1, cone, G, g) in blocks:
            bt.add_linear_constraint(k0, k1, cone, G, g)
        res = bt.ilqr_solve(iterations_max=60, penalty_initial=1.0, penalty_scaling=10.0, use_backtracking=backtracking)
        x, _ = bt.get_nominal()
        off, worst = 0, 0.0
        for b in range(batch):
            st, its, xr = ref[b]
            if res["status"][b] != st or res["iterations"][b] != its:
                off += 1
            elif st == 0:
                worst = max(worst, float(np.abs(x[b] - xr).max()))
        tot[name][0] += batch; tot[name][1] += off; tot[name][2] = max(tot[name][2], worst)
        line.append("%s: %d of %d off the oracle, |dx| %.1e" % (name, off, batch, worst))
        bt.close()
    print("case %2d: N = %2d batch = %2d backtracking %d  |  %s" % (it, N, batch, backtracking, "  |  ".join(line)), flush=True)
for name in tot:
    print("%s rounds: %d of %d problems end with another status / iteration count than the oracle; converged rest within %.1e" % (
        name, tot[name][1], tot[name][0], tot[name][2]))
