#!/usr/bin/env python3
"""Random batches through plan MFMA16's AL-iLQR solve with the affine line-search rounds and again with every trial a
rollout (ALTRO_HIP_AFFINE=0): the two agree to rounding, not bit for bit, so what is counted is how many PROBLEMS end with another
status or iteration count, and how far the trajectories of those that agree are apart.  Cases as in tools/fuzz_dpp.py (fp64 only).

    python tools/fuzz_affine.py [cases] [seed]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import altro_amd  # noqa: E402
from tests import problems  # noqa: E402
from tools.fuzz_dpp import blocks_for  # noqa: E402


def solve(p, N, n, m, blocks, affine, kw):
    os.environ["ALTRO_HIP_AFFINE"] = "1" if affine else "0"
    bt = altro_amd.Batch(N, n, m, p["x0"].shape[0])
    bt.set_dynamics(p["A"], p["B"], p["f"])
    bt.set_tracking_cost(p["Qd"], p["Rd"], p["xref"], p["uref"])
    bt.set_initial_state(p["x0"]); bt.set_input_guess(p["u0"])
    for (k0, k1, cone, G, g) in blocks:
        bt.add_linear_constraint(k0, k1, cone, G, g)
    res = dict(bt.ilqr_solve(**kw))
    res["x"], res["u"] = bt.get_nominal()
    bt.close()
    return res


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    problems_total = problems_off = 0
    worst = 0.0
    for c in range(cases):
        n, m = (12, 4) if rng.random() < 0.5 else (int(rng.integers(7, 13)), int(rng.integers(1, 5)))
        N = int(rng.integers(1, 80)); batch = int(rng.integers(1, 70))
        p = problems.ilqr12x4_problem(batch, N, bool(rng.integers(0, 2)), n=n, m=m)
        blocks = blocks_for(rng, N, n, m)
        kw = dict(iterations_max=int(rng.integers(3, 40)), use_backtracking=bool(rng.integers(0, 2)), penalty_initial=1.0, penalty_scaling=10.0)
        if "FUZZ_MARGIN" in os.environ:
            kw["decision_margin"] = float(os.environ["FUZZ_MARGIN"])
        a = solve(p, N, n, m, blocks, True, kw)
        b = solve(p, N, n, m, blocks, False, kw)
        same = (a["status"] == b["status"]) & (a["iterations"] == b["iterations"])
        conv = same & (a["status"] == 0)
        d = float(np.abs(a["x"][conv] - b["x"][conv]).max()) if conv.any() else 0.0
        worst = max(worst, d)
        problems_total += batch; problems_off += int((~same).sum())
        for i in np.nonzero(~same)[0]:
            print("      problem %d: affine status %d after %d iterations, rollout status %d after %d" % (i, a["status"][i], a["iterations"][i], b["status"][i], b["iterations"][i]), flush=True)
        print("%s case %3d: (n, m) = (%2d, %d) N = %2d batch = %2d blocks %d backtracking %d sweeps %d merit launches %d / %d: %d of %d problems differ; |dx| %.1e"
              % ("ok " if same.all() else "off", c, n, m, N, batch, len(blocks), kw["use_backtracking"], a["sweeps"], a["merit_launches"],
                 b["merit_launches"], int((~same).sum()), batch, d), flush=True)
    print("%d of %d problems end with another status / iteration count; largest |x_affine - x_rollout| among the converged rest %.2e"
          % (problems_off, problems_total, worst))
    return 0


if __name__ == "__main__":
    sys.exit(main())
