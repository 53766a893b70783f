#!/usr/bin/env python3
"""Soak: whole AL-iLQR solves on plan GENERIC (any n, m up to 32; random constraint blocks in every cone, shared or per-problem
right-hand sides, terminal blocks, both line searches) against the oracle's restatement of SolverImpl, problem by problem: status,
iterations, dual updates; trajectories where both converged.

    python tests/soak/fuzz_generic_al.py [cases] [seed] [--big-soc]

--big-soc: second-order cones of 2 .. 12 rows (over all inputs and some states; GEN_MAXSOC = 32 on this plan) instead of 2 .. 4 -- another
random stream than the default's, whose counts tests/soak/README.md quotes.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import altro_amd  # noqa: E402
from oracle import oracle  # noqa: E402
from tests import problems  # noqa: E402

BIG_SOC = "--big-soc" in sys.argv
argv = [a for a in sys.argv if not a.startswith("--")]
cases = int(argv[1]) if len(argv) > 1 else 30
rng = np.random.default_rng(int(argv[2]) if len(argv) > 2 else 0)
bad = 0
for it in range(cases):
    n = int(rng.integers(13, 25)) if rng.random() < 0.7 else int(rng.integers(5, 13))
    m = int(rng.integers(5, 9))                      # (m > 4 keeps the shape off the tile plan)
    N = int(rng.integers(3, 20)); batch = int(rng.integers(1, 12))
    dense = bool(rng.random() < 0.4); backtracking = bool(rng.random() < 0.4)
    p = problems.ilqr12x4_problem(batch, N, True, n=n, m=m)
    if dense:
        p.update(problems.quadratic_cost(batch, N, n, m))
    bt = altro_amd.Batch(N, n, m, batch, plan=altro_amd.PLAN_GENERIC)
    assert bt.plan == altro_amd.PLAN_GENERIC
    bt.set_dynamics(p["A"], p["B"], p["f"])
    if dense:
        bt.set_quadratic_cost(p["Q"], p["R"], p["H"], p["q"], p["r"], p["c"])
    else:
        bt.set_tracking_cost(p["Qd"], p["Rd"], p["xref"], p["uref"])
    bt.set_initial_state(p["x0"]); bt.set_input_guess(p["u0"])
    w = n + m
    blocks, cones = [], []
    used = np.zeros(N + 1, dtype=int)
    for _ in range(int(rng.integers(1, 4))):
        cone = int(rng.choice([altro_amd.CONE_EQUALITY, altro_amd.CONE_INEQUALITY, altro_amd.CONE_INEQUALITY, altro_amd.CONE_SOC]))
        pp = int(rng.integers(2, 13 if BIG_SOC else 5)) if cone == altro_amd.CONE_SOC else int(rng.integers(1, 9 if cone != altro_amd.CONE_EQUALITY else 3))
        k0 = int(rng.integers(0, N + 1)); k1 = int(rng.integers(k0, N + 1))
        if rng.random() < 0.5:
            k0, k1 = 0, N - 1
        if (used[k0:k1 + 1] >= 2).any():
            continue
        used[k0:k1 + 1] += 1
        G = np.zeros((pp, w))
        if rng.random() < 0.6 and cone != altro_amd.CONE_SOC:          # bound-type rows on inputs
            for r in range(pp):
                G[r, n + int(rng.integers(0, m))] = 1.0 if rng.random() < 0.5 else -1.0
            g = np.full(pp, 0.4)
        elif cone == altro_amd.CONE_SOC:
            for r in range(pp - 1):
                if r < m or not BIG_SOC:
                    G[r, n + r % m] = 1.0
                else:
                    G[r, (r - m) % n] = 0.2                         # (rows past the inputs: states, scaled down)
            g = np.zeros(pp); g[-1] = -0.5
        else:
            G = 0.3 * rng.standard_normal((pp, w)); g = 0.5 + 0.2 * rng.random(pp)
            if cone == altro_amd.CONE_EQUALITY:
                G[:, :n] = 0.0; g = 0.05 * rng.standard_normal(pp)
        per_problem = bool(rng.random() < 0.3) and cone != altro_amd.CONE_SOC
        gg = np.tile(g, (batch, 1)) + (0.01 * rng.standard_normal((batch, pp)) if per_problem else 0.0) if per_problem else g
        blocks.append((k0, k1, cone, G, gg, per_problem)); cones.append(cone)
        bt.add_linear_constraint(k0, k1, cone, G, gg)
    res = bt.ilqr_solve(iterations_max=40, penalty_initial=1.0, penalty_scaling=10.0, use_backtracking=backtracking)
    x, u = bt.get_nominal()
    ok = True
    for b in sorted(set([0, batch - 1])):
        s = oracle.ILQR(N, n, m, 0.01, oracle.DYN_LINEAR, cost_kind=oracle.COST_QUADRATIC if dense else oracle.COST_DIAGONAL)
        s.L.oracle_ilqr_set_linear_dynamics(s.h, np.ascontiguousarray(p["A"][b]), np.ascontiguousarray(p["B"][b]), np.ascontiguousarray(p["f"][b]).ctypes.data)
        for k in range(N + 1):
            kk = min(k, N - 1)
            if dense:
                s.L.oracle_ilqr_set_quadratic_cost(s.h, k, np.ascontiguousarray(p["Q"][b, k]), np.ascontiguousarray(p["R"][b, kk]).ctypes.data,
                                                   np.ascontiguousarray(p["H"][b, kk]).ctypes.data, np.ascontiguousarray(p["q"][b, k]),
                                                   np.ascontiguousarray(p["r"][b, kk]).ctypes.data, float(p["c"][b, k]))
            else:
                s.L.oracle_ilqr_set_lqr_cost(s.h, k, np.ascontiguousarray(p["Qd"][b, k]), np.ascontiguousarray(p["Rd"][b, kk]),
                                             np.ascontiguousarray(p["xref"][b, k]), np.ascontiguousarray(p["uref"][b, kk]))
        s.L.oracle_ilqr_set_initial_state(s.h, np.ascontiguousarray(p["x0"][b]))
        for (k0, k1, cone, G, gg, per_problem) in blocks:
            for k in range(k0, k1 + 1):
                s.add_linear_constraint(k, cone, G, gg[b] if per_problem else gg)
        s.L.oracle_ilqr_initialize(s.h)
        for k in range(N):
            s.L.oracle_ilqr_set_input(s.h, k, np.ascontiguousarray(p["u0"][b, k]))
        s.set_penalty(1.0, 10.0)
        s.L.oracle_ilqr_set_options(s.h, 40, 1e-4, 1e-4, 1e-8, 1 if backtracking else 0)
        status, iters, log = s.solve()
        same = res["status"][b] == status and res["iterations"][b] == iters
        if same and status == 0:
            same = np.abs(x[b] - s.get("x")).max() <= 1e-6 * max(1.0, np.abs(x[b]).max()) and np.abs(u[b] - s.get("u")).max() <= 1e-5 * max(1.0, np.abs(u[b]).max())
        ok = ok and same
        if not same:
            print("   problem %d: device status %d iterations %d, oracle %d %d" % (b, res["status"][b], res["iterations"][b], status, iters))
    bad += 0 if ok else 1
    print("%s case %3d: (n, m) = (%2d, %d) N = %2d batch = %2d %s cones %s backtracking %d sweeps %d converged %d/%d"
          % ("ok " if ok else "BAD", it, n, m, N, batch, "dense" if dense else "diag ", cones, backtracking, res["sweeps"], int((res["status"] == 0).sum()), batch), flush=True)
    bt.close()
print("%d of %d cases differ" % (bad, cases))
