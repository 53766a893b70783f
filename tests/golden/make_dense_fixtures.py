#!/usr/bin/env python3
"""Generates tests/golden/dense_fixtures.npz: dense outputs of the CPU oracle (oracle/, itself pinned to the
reference's constants by tests/test_oracle_kat.py) for every BASELINE.json config at reduced batch (SURVEY.md section 7
step 1).  The inputs are NOT stored: tests/problems.py regenerates them from its counter-based PRNG; a checksum of the
inputs is, so that a drifting generator is caught.  With these the GPU parity evidence can be replayed on a box that
has neither gcc nor the oracle's .so (tests/test_gpu_golden.py), and the oracle itself is regression-pinned
(tests/test_oracle_golden.py).

    python tests/golden/make_dense_fixtures.py          # rewrites the .npz (run in the build container)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle          # noqa: E402
from tests import problems         # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "dense_fixtures.npz")
from tests.golden_cases import P_KNOTS, SOLVES, TVLQR, checksum, solve_case, tvlqr_problem   # noqa: E402,F401

OKIND = {"double_integrator": oracle.MODEL_DI, "pendulum": oracle.MODEL_PENDULUM, "bicycle": oracle.MODEL_BICYCLE}


def oracle_tvlqr(name):
    pr = tvlqr_problem(name)
    o = oracle.backward_batch(pr["A"], pr["B"], pr["f"], pr["Q"], pr["R"], pr["H"], pr["q"], pr["r"])
    o.update(oracle.forward_batch(pr["A"], pr["B"], pr["f"], o["K"], o["d"], o["P"], o["p"], pr["x0"]))
    out = {k: o[k] for k in ("K", "d", "p", "x", "u", "y", "dV", "status")}
    out["P"] = o["P"][:, P_KNOTS[name]] if name in P_KNOTS else o["P"]
    out["input_checksum"] = checksum(pr)
    return out


def oracle_solve(name):
    c = solve_case(name)
    xs, us, its, sts = [], [], [], []
    for x0 in c["x0s"]:
        s = oracle.ILQR(c["N"], c["n"], c["m"], c["h"], oracle.DYN_MODEL, OKIND[c["model_name"]], model_dim=c["dim"],
                        cost_kind=oracle.COST_DIAGONAL)
        for k in range(c["N"] + 1):
            s.L.oracle_ilqr_set_lqr_cost(s.h, k, np.ascontiguousarray(c["Qfd"] if k == c["N"] else c["Qd"], dtype=float),
                                         np.ascontiguousarray(c["Rd"], dtype=float), np.ascontiguousarray(c["xf"], dtype=float),
                                         np.zeros(c["m"]))
        s.L.oracle_ilqr_set_initial_state(s.h, np.ascontiguousarray(x0, dtype=float))
        s.L.oracle_ilqr_initialize(s.h)
        for k in range(c["N"]):
            s.L.oracle_ilqr_set_input(s.h, k, np.ascontiguousarray(c["u0"], dtype=float))
        s.L.oracle_ilqr_set_options(s.h, c["itmax"], 1e-4, 1e-4, 1e-8, c["bt"])
        status, iters, _ = s.solve()
        xs.append(s.get("x")); us.append(s.get("u")); its.append(iters); sts.append(status)
    return dict(x=np.stack(xs), u=np.stack(us), iterations=np.array(its), status=np.array(sts))


def generate():
    data = {}
    for name in TVLQR:
        for k, v in oracle_tvlqr(name).items():
            data["tvlqr_%s_%s" % (name, k)] = v
    for name in SOLVES:
        for k, v in oracle_solve(name).items():
            data["solve_%s_%s" % (name, k)] = v
    return data


if __name__ == "__main__":
    oracle.build()
    d = generate()
    np.savez_compressed(OUT, **d)
    print("wrote %s: %d arrays, %.1f KB" % (OUT, len(d), os.path.getsize(OUT) / 1e3))
