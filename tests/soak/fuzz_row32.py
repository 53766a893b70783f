#!/usr/bin/env python3
"""Soak: the row-layout loop kernels of plan MFMA32's shapes (kernels/ilqr_row32.hip: merit, two-trial merit, stationarity / feasibility,
the head of a solve, the constrained expansion) against plan GENERIC's wave-per-problem kernels (ALTRO_HIP_FORM_GENERIC_MERIT_LDS) on
random shapes of the 156, horizons 2..90 (one to several chunks), odd and even batches, diagonal and dense costs, random constraint
tables (bound-type and dense blocks of 1..32 rows in the zero / orthant cones, ragged ranges, per-problem right-hand sides):
  * phi, phi', the candidate trajectory, the stored gradient, stationarity and feasibility of three merit evaluations: 1e-13 relative,
    and how many cases are bit-identical;
  * whole solves: the same status and iteration count for every problem, trajectories 1e-9 (a solve takes the same decisions whichever
    kernels evaluate it, because they return the same values).

    python tests/soak/fuzz_row32.py [cases] [seed]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import altro_amd  # noqa: E402
from tests import problems  # noqa: E402
from tests.test_gpu_row32 import build, evaluate  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
SHAPES = [(n, m) for n in range(5, 32) for m in range(1, 9) if n + m <= 32 and not (n <= 12 and m <= 4)]
assert len(SHAPES) == 156


def random_blocks(N, n, m, batch):
    w = n + m
    blocks = []
    used = np.zeros(N + 1, dtype=int)
    for _ in range(int(rng.integers(0, 5))):
        k0 = int(rng.integers(0, N)); k1 = int(rng.integers(k0, N + 1))
        if rng.random() < 0.5:
            k0, k1 = 0, N - 1
        if rng.random() < 0.2:
            k0 = k1 = N
        if used[k0:k1 + 1].max() >= 8:
            continue
        used[k0:k1 + 1] += 1
        term = k1 >= N
        cols = n if term else w
        if rng.random() < 0.5:                                   # bound-type rows
            p = int(rng.integers(1, min(32, 2 * cols) + 1))
            G = np.zeros((p, w))
            for r in range(p):
                G[r, int(rng.integers(0, cols))] = 1.0 if rng.random() < 0.5 else -1.0
        else:
            p = int(rng.integers(1, 33))
            G = rng.normal(size=(p, w)) * (rng.random((p, w)) < 0.4)
            if term:
                G[:, n:] = 0.0
            G[np.abs(G).sum(axis=1) == 0, 0] = 1.0
        cone = altro_amd.CONE_INEQUALITY if rng.random() < 0.75 else altro_amd.CONE_EQUALITY
        g = rng.uniform(0.3, 2.0, size=p) * (1.0 if cone == altro_amd.CONE_INEQUALITY else 0.05)
        if rng.random() < 0.25:
            g = g[None, :] * (1.0 + 0.1 * rng.random((batch, p)))
        blocks.append((k0, k1, cone, G, g))
    return blocks


exact = worst_ok = 0
solves_equal = problems_total = 0
for it in range(cases):
    n, m = SHAPES[int(rng.integers(0, len(SHAPES)))]
    N = int(rng.integers(2, 91)); batch = int(rng.integers(1, 10)); dense = bool(rng.random() < 0.5)
    p = problems.ilqr12x4_problem(batch, N, True, n=n, m=m)
    if dense:
        p.update(problems.quadratic_cost(batch, N, n, m))
    blocks = random_blocks(N, n, m, batch)
    out = {}
    for name, forms in (("row", 0), ("lds", altro_amd.FORM_GENERIC_MERIT_LDS)):
        bt = build(p, N, n, m, batch, dense, forms, blocks)
        assert bt.plan == altro_amd.PLAN_MFMA32
        res = evaluate(bt, batch)
        r = bt.ilqr_solve(iterations_max=25, penalty_initial=1.0, penalty_scaling=10.0)
        res["solve_status"] = r["status"].copy(); res["solve_iterations"] = r["iterations"].copy(); res["solve_x"] = bt.get_nominal()[0].copy()
        out[name] = res
        bt.close()
    same = True
    for key in out["row"]:
        a, b = out["row"][key], out["lds"][key]
        if key in ("solve_status", "solve_iterations"):
            assert np.array_equal(a, b), (it, n, m, N, key, a, b)
        elif key == "solve_x":
            np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-9, err_msg="case %d (%d, %d) N %d %s" % (it, n, m, N, key))
        else:
            np.testing.assert_allclose(a, b, rtol=1e-13, atol=1e-13 * max(1.0, float(np.abs(b).max())), err_msg="case %d (%d, %d) N %d %s" % (it, n, m, N, key))
            same = same and np.array_equal(a, b)
    exact += int(same); problems_total += batch
    print("case %2d: (%2d, %d) N %2d batch %d %s, %d blocks: %s; solves: %d converged, iterations %s"
          % (it, n, m, N, batch, "dense" if dense else "diag ", len(blocks), "bit-identical" if same else "within 1e-13",
             int((out["row"]["solve_status"] == 0).sum()), out["row"]["solve_iterations"].tolist()), flush=True)
print("%d of %d cases bit-identical in every evaluated quantity, the rest within 1e-13; all %d problems solved with the same status and iteration count in both forms"
      % (exact, cases, problems_total))
