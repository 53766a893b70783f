#!/usr/bin/env python3
"""Random batches through plan MFMA16's iLQR solve with every DPP-form kernel on, and again with the LDS forms they replace
(kernels/ilqr_merit2_dpp.hip): statuses, iteration counts, phi, stationarity, feasibility, trajectories and gains must be equal
bit for bit.  Shapes (padded ones too), horizons, batch sizes, dense random constraint Jacobians, cones, line searches, storage type.

    python tools/fuzz_dpp.py [cases] [seed]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import altro_amd  # noqa: E402
from tests import problems  # noqa: E402

SWITCHES = ("ALTRO_HIP_MERIT_DPP", "ALTRO_HIP_EXPAND_DPP", "ALTRO_HIP_ALROWS_DPP")
os.environ["ALTRO_HIP_AFFINE"] = "0"   # the kernel FORMS against each other: every line-search trial a rollout in both (the affine rounds exist in the row layout only)
KEYS = ("status", "iterations", "dual_updates", "phi", "stationarity", "feasibility", "alpha", "penalty")


def blocks_for(rng, N, n, m):
    w = n + m
    out = []
    kind = rng.integers(0, 5)
    if kind == 0:
        return out
    if kind in (1, 3, 4):   # input bounds
        Gb = np.zeros((2 * m, w)); Gb[:m, n:] = np.eye(m); Gb[m:, n:] = -np.eye(m)
        out.append((0, N - 1, altro_amd.CONE_INEQUALITY, Gb, np.full(2 * m, 0.25 + 0.2 * rng.random())))
    if kind in (2, 3):      # a dense random half-space / equality block on states and inputs
        p = int(rng.integers(1, 5))
        G = rng.normal(size=(p, w)) * 0.5
        cone = altro_amd.CONE_INEQUALITY if rng.random() < 0.7 else altro_amd.CONE_EQUALITY
        k0 = int(rng.integers(0, max(1, N // 2)))
        out.append((k0, N - 1 if rng.random() < 0.7 else N, cone, G, np.abs(rng.normal(size=p)) + (0.5 if cone == altro_amd.CONE_INEQUALITY else 0.0) * 1.0))
    if kind == 4 and m >= 3:  # a second-order cone on the first inputs
        Gs = np.zeros((4, w)); Gs[0, n] = 1.0; Gs[1, n + 1] = 1.0; Gs[2, n + 2] = 1.0
        out[0:0] = [(0, N - 1, altro_amd.CONE_SOC, Gs, np.array([0.0, 0.0, 0.0, -0.3 - 0.1 * rng.random()]))]
    return out[:2] if len(out) > 2 else out


def solve(p, N, n, m, blocks, dtype, dpp, kw):
    for s in SWITCHES:
        os.environ[s] = ("2" if s == "ALTRO_HIP_MERIT_DPP" else "1") if dpp else "0"
    bt = altro_amd.Batch(N, n, m, p["x0"].shape[0], dtype=dtype)
    bt.set_dynamics(p["A"], p["B"], p["f"])
    bt.set_tracking_cost(p["Qd"], p["Rd"], p["xref"], p["uref"])
    bt.set_initial_state(p["x0"])
    bt.set_input_guess(p["u0"])
    for (k0, k1, cone, G, g) in blocks:
        bt.add_linear_constraint(k0, k1, cone, G, g)
    res = dict(bt.ilqr_solve(**kw))
    res["x"], res["u"] = bt.get_nominal()
    res["K"], res["d"] = bt.get("K"), bt.get("d")
    bt.close()
    return res


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    bad = 0
    for c in range(cases):
        n, m = (12, 4) if rng.random() < 0.5 else (int(rng.integers(7, 13)), int(rng.integers(1, 5)))
        N = int(rng.integers(1, 40))
        batch = int(rng.integers(1, 70))
        dtype = altro_amd.F32 if rng.random() < 0.2 else altro_amd.F64
        p = problems.ilqr12x4_problem(batch, N, bool(rng.integers(0, 2)), n=n, m=m)
        if rng.random() < 0.3:
            p["x0"][:: int(rng.integers(2, 5))] *= 0.0
        blocks = blocks_for(rng, N, n, m)
        kw = dict(iterations_max=int(rng.integers(3, 30)), use_backtracking=bool(rng.integers(0, 2)), penalty_initial=1.0, penalty_scaling=10.0)
        a = solve(p, N, n, m, blocks, dtype, True, kw)
        b = solve(p, N, n, m, blocks, dtype, False, kw)
        diff = [k for k in KEYS + ("x", "u", "K", "d") if not np.array_equal(np.asarray(a[k]), np.asarray(b[k]), equal_nan=True)]
        tag = "ok " if not diff else "DIFF"
        bad += bool(diff)
        print("%s case %3d: (n, m) = (%2d, %d) N = %2d batch = %2d %s blocks %s cones %s backtracking %d sweeps %d merit launches %d%s"
              % (tag, c, n, m, N, batch, "f32" if dtype == altro_amd.F32 else "f64", len(blocks), [bk[2] for bk in blocks],
                 kw["use_backtracking"], a["sweeps"], a["merit_launches"], (" -> " + ",".join(diff)) if diff else ""))
    print("%d of %d cases differ" % (bad, cases))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
