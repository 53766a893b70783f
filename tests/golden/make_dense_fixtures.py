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
from tests.golden_cases import (AL_KINDS, MERIT_ALPHAS, P_KNOTS, REG_CASE, SOLVES, TVLQR, al_case, checksum, load_kats,   # noqa: E402,F401
                                lq12_case, merit_case, mpc_case, mpc_linear_costs, quad12_case, quad4_case, solve_case, tvlqr_problem)

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


# ---- round 3: constrained solves (f2), MPC (f3), the (12, 4) iLQR loop, merit / expansion / stationarity (a5, a7-a11), reg (f4)
def oracle_al(kind):
    c = al_case(load_kats(), kind)
    out = dict(x=[], u=[], status=[], iterations=[], feasibility=[])
    for x0 in c["x0s"]:
        s = oracle.ILQR(c["N"], c["n"], c["m"], c["h"], oracle.DYN_MODEL, oracle.MODEL_DI, model_dim=c["dim"], cost_kind=oracle.COST_DIAGONAL)
        for k in range(c["N"] + 1):
            s.L.oracle_ilqr_set_lqr_cost(s.h, k, np.full(c["n"], c["Q"]), np.full(c["m"], c["R"]), c["xf"].copy(), np.zeros(c["m"]))
        s.L.oracle_ilqr_set_initial_state(s.h, np.ascontiguousarray(x0))
        for (k0, k1, cone, G, g) in c["blocks"]:
            for k in range(k0, k1 + 1):
                s.add_linear_constraint(k, cone, G, g)
        s.L.oracle_ilqr_initialize(s.h)
        s.set_penalty(c["penalty_initial"], c["penalty_scaling"])
        s.L.oracle_ilqr_set_options(s.h, c["itmax"], 1e-4, 1e-4, 1e-8, 0)
        status, iters, log = s.solve()
        out["x"].append(s.get("x")); out["u"].append(s.get("u")); out["status"].append(status); out["iterations"].append(iters)
        out["feasibility"].append(log[iters - 1, 6] if iters > 0 else 0.0)
    return {k: np.array(v) for k, v in out.items()}


def oracle_mpc():
    import ctypes as C
    c = mpc_case()
    N, n, m = c["N"], c["n"], c["m"]
    mdl = oracle.make_model(oracle.MODEL_BICYCLE)
    L = oracle.lib()
    its, u0s, xs, sts = [], [], [], []
    for x0 in c["x0s"]:
        s = oracle.ILQR(N, n, m, c["h"], oracle.DYN_MODEL, oracle.MODEL_BICYCLE, cost_kind=oracle.COST_DIAGONAL)
        for k in range(N + 1):
            s.L.oracle_ilqr_set_lqr_cost(s.h, k, np.full(n, c["QD"]), np.full(m, c["RD"]), np.ascontiguousarray(c["x_ref"][k]),
                                         np.ascontiguousarray(c["u_ref"][min(k, len(c["u_ref"]) - 1)]))
        for k in range(N + 1):
            s.add_linear_constraint(k, oracle.CONE_INEQUALITY, c["G"], c["g"])
        s.L.oracle_ilqr_set_initial_state(s.h, np.ascontiguousarray(x0))
        s.L.oracle_ilqr_initialize(s.h)
        for k in range(N):
            s.L.oracle_ilqr_set_input(s.h, k, c["u0"])
        s.L.oracle_ilqr_set_options(s.h, 80, 1e-4, 1e-4, 1e-8, 1)
        x = np.array(x0, dtype=float)
        row = ([], [], [], [])
        for it in range(c["nsim"]):
            status, iters, _ = s.solve()
            u = s.get("u")[0].copy()
            xn = np.zeros(n)
            L.oracle_discrete_dynamics(C.byref(mdl), xn, np.ascontiguousarray(x), np.ascontiguousarray(u), c["h"])
            x = xn
            row[0].append(iters); row[1].append(u); row[2].append(x.copy()); row[3].append(status)
            q, cc = mpc_linear_costs(c, it + 1)
            for k in range(N + 1):
                s.L.oracle_ilqr_update_linear_costs(s.h, k, q[k].ctypes.data, None, float(cc[k]))
            s.L.oracle_ilqr_set_initial_state(s.h, x)
            s.L.oracle_ilqr_shift_trajectory(s.h)
        its.append(row[0]); u0s.append(row[1]); xs.append(row[2]); sts.append(row[3])
    return dict(iterations=np.array(its), u0=np.array(u0s), x_next=np.array(xs), status=np.array(sts))


def _lq12_oracle(c, b):
    p, N = c["p"], c["N"]
    s = oracle.ILQR(N, 12, 4, 0.01, oracle.DYN_LINEAR, cost_kind=oracle.COST_DIAGONAL)
    s.L.oracle_ilqr_set_linear_dynamics(s.h, np.ascontiguousarray(p["A"][b]), np.ascontiguousarray(p["B"][b]),
                                        np.ascontiguousarray(p["f"][b]).ctypes.data)
    for k in range(N + 1):
        s.L.oracle_ilqr_set_lqr_cost(s.h, k, np.ascontiguousarray(p["Qd"][b, k]), np.ascontiguousarray(p["Rd"][b, min(k, N - 1)]),
                                     np.ascontiguousarray(p["xref"][b, k]), np.ascontiguousarray(p["uref"][b, min(k, N - 1)]))
    s.L.oracle_ilqr_set_initial_state(s.h, np.ascontiguousarray(p["x0"][b]))
    for (k0, k1, cone, G, g) in c["blocks"]:
        for k in range(k0, k1 + 1):
            s.add_linear_constraint(k, cone, G, g)
    s.L.oracle_ilqr_initialize(s.h)
    for k in range(N):
        s.L.oracle_ilqr_set_input(s.h, k, np.ascontiguousarray(p["u0"][b, k]))
    return s


def oracle_lq12(constrained):
    c = lq12_case(constrained)
    out = dict(x=[], u=[], status=[], iterations=[], feasibility=[])
    for b in range(c["p"]["x0"].shape[0]):
        s = _lq12_oracle(c, b)
        if constrained:
            s.set_penalty(1.0, 10.0)
        s.L.oracle_ilqr_set_options(s.h, c["itmax"], 1e-4, 1e-4, 1e-8, 0)
        status, iters, log = s.solve()
        out["x"].append(s.get("x")); out["u"].append(s.get("u")); out["status"].append(status); out["iterations"].append(iters)
        out["feasibility"].append(log[iters - 1, 6] if iters > 0 else 0.0)
    return {k: np.array(v) for k, v in out.items()}


# ---- round 4: the dense quadratic cost of ALTROSolver::SetQuadraticCost in the iLQR loop (row a9) -----------------------------
def _set_quadratic(s, cost, b, N):
    for k in range(N + 1):
        kk = min(k, N - 1)
        s.L.oracle_ilqr_set_quadratic_cost(s.h, k, np.ascontiguousarray(cost["Q"][b, k]), np.ascontiguousarray(cost["R"][b, kk]).ctypes.data,
                                           np.ascontiguousarray(cost["H"][b, kk]).ctypes.data, np.ascontiguousarray(cost["q"][b, k]),
                                           np.ascontiguousarray(cost["r"][b, kk]).ctypes.data, float(cost["c"][b, k]))


def _quad12_oracle(c, b):
    p, N = c["p"], c["N"]
    s = oracle.ILQR(N, 12, 4, 0.01, oracle.DYN_LINEAR, cost_kind=oracle.COST_QUADRATIC)
    s.L.oracle_ilqr_set_linear_dynamics(s.h, np.ascontiguousarray(p["A"][b]), np.ascontiguousarray(p["B"][b]),
                                        np.ascontiguousarray(p["f"][b]).ctypes.data)
    _set_quadratic(s, p, b, N)
    s.L.oracle_ilqr_set_initial_state(s.h, np.ascontiguousarray(p["x0"][b]))
    for (k0, k1, cone, G, g) in c["blocks"]:
        for k in range(k0, k1 + 1):
            s.add_linear_constraint(k, cone, G, g)
    s.L.oracle_ilqr_initialize(s.h)
    for k in range(N):
        s.L.oracle_ilqr_set_input(s.h, k, np.ascontiguousarray(p["u0"][b, k]))
    return s


def _solve_rows(make, nprob, constrained, itmax):
    out = dict(x=[], u=[], status=[], iterations=[], feasibility=[])
    for b in range(nprob):
        s = make(b)
        if constrained:
            s.set_penalty(1.0, 10.0)
        s.L.oracle_ilqr_set_options(s.h, itmax, 1e-4, 1e-4, 1e-8, 0)
        status, iters, log = s.solve()
        out["x"].append(s.get("x")); out["u"].append(s.get("u")); out["status"].append(status); out["iterations"].append(iters)
        out["feasibility"].append(log[iters - 1, 6] if iters > 0 else 0.0)
    return {k: np.array(v) for k, v in out.items()}


def oracle_quad12(constrained):
    c = quad12_case(constrained)
    return _solve_rows(lambda b: _quad12_oracle(c, b), c["p"]["x0"].shape[0], constrained, c["itmax"])


def _quad4_oracle(c, b):
    s = oracle.ILQR(c["N"], c["n"], c["m"], c["h"], oracle.DYN_MODEL, OKIND[c["model_name"]], model_dim=c["dim"], cost_kind=oracle.COST_QUADRATIC)
    _set_quadratic(s, c["cost"], b, c["N"])
    s.L.oracle_ilqr_set_initial_state(s.h, np.ascontiguousarray(c["x0s"][b], dtype=float))
    for (k0, k1, cone, G, g) in c["blocks"]:
        for k in range(k0, k1 + 1):
            s.add_linear_constraint(k, cone, G, g)
    s.L.oracle_ilqr_initialize(s.h)
    for k in range(c["N"]):
        s.L.oracle_ilqr_set_input(s.h, k, np.ascontiguousarray(c["u0"], dtype=float))
    return s


def oracle_quad4(constrained):
    c = quad4_case(constrained)
    return _solve_rows(lambda b: _quad4_oracle(c, b), len(c["x0s"]), constrained, c["itmax"])


def _merit_rows(make, nprob):
    """For every problem and every alpha of MERIT_ALPHAS (problem-major): phi, phi', candidate, expansion at it, stationarity."""
    keys = ("phi", "dphi", "x", "u", "y", "lx", "lu", "A", "B", "stationarity", "K")
    out = {k: [] for k in keys}
    for b in range(nprob):
        for al in MERIT_ALPHAS:
            s = make(b)
            s.L.oracle_ilqr_open_loop_rollout(s.h); s.L.oracle_ilqr_copy_trajectory(s.h)
            s.L.oracle_ilqr_calc_dynamics_expansions(s.h); s.L.oracle_ilqr_calc_cost_gradient(s.h)
            s.L.oracle_ilqr_calc_expansions(s.h)
            assert s.L.oracle_ilqr_backward_pass(s.h) == -1
            out["K"].append(s.get("K"))
            phi, dphi = s.merit(al)
            out["phi"].append(phi); out["dphi"].append(dphi)
            out["x"].append(s.get("x_cand")); out["u"].append(s.get("u_cand")); out["y"].append(s.get("y_cand"))
            out["lx"].append(s.get("lx")); out["lu"].append(s.get("lu")); out["A"].append(s.get("A")); out["B"].append(s.get("B"))
            out["stationarity"].append(s.L.oracle_ilqr_stationarity(s.h))
    return {k: np.array(v) for k, v in out.items()}


def oracle_merit(name):
    if name == "quad12":
        c = quad12_case(False)
        return _merit_rows(lambda b: _quad12_oracle(c, b), c["p"]["x0"].shape[0])
    if name == "quad4":
        c = quad4_case(False)
        return _merit_rows(lambda b: _quad4_oracle(c, b), len(c["x0s"]))
    if name == "lq12":
        c = lq12_case(False)
        return _merit_rows(lambda b: _lq12_oracle(c, b), c["p"]["x0"].shape[0])
    c = merit_case(name)

    def make(b):
        s = oracle.ILQR(c["N"], c["n"], c["m"], c["h"], oracle.DYN_MODEL, OKIND[c["model_name"]], model_dim=c["dim"],
                        cost_kind=oracle.COST_DIAGONAL)
        for k in range(c["N"] + 1):
            s.L.oracle_ilqr_set_lqr_cost(s.h, k, np.ascontiguousarray(c["Qfd"] if k == c["N"] else c["Qd"], dtype=float),
                                         np.ascontiguousarray(c["Rd"], dtype=float), np.ascontiguousarray(c["xf"], dtype=float),
                                         np.zeros(c["m"]))
        s.L.oracle_ilqr_set_initial_state(s.h, np.ascontiguousarray(c["x0s"][b], dtype=float))
        s.L.oracle_ilqr_initialize(s.h)
        for k in range(c["N"]):
            s.L.oracle_ilqr_set_input(s.h, k, np.ascontiguousarray(c["u0"], dtype=float))
        return s
    return _merit_rows(make, len(c["x0s"]))


def oracle_reg():
    """tvlqr_BackwardPass's reg argument (tvlqr.cpp:159-164): a positive reg on well-posed problems, and indefinite R blocks
    that fail without it (the failing knot point is the status) and pass with it."""
    pr = tvlqr_problem(REG_CASE["name"])
    o = oracle.backward_batch(pr["A"], pr["B"], pr["f"], pr["Q"], pr["R"], pr["H"], pr["q"], pr["r"], reg=REG_CASE["reg"])
    out = {"reg_" + k: o[k] for k in ("K", "d", "P", "p", "dV", "status")}
    R = pr["R"].copy(); R[1] *= -1.0                       # problem 1: R negative definite
    bad = oracle.backward_batch(pr["A"], pr["B"], pr["f"], pr["Q"], R, pr["H"], pr["q"], pr["r"], reg=0.0)
    out["bad_status"] = bad["status"]
    out["bad_K0"] = bad["K"][0]
    return out


def generate():
    data = {}
    for name in TVLQR:
        for k, v in oracle_tvlqr(name).items():
            data["tvlqr_%s_%s" % (name, k)] = v
    for name in SOLVES:
        for k, v in oracle_solve(name).items():
            data["solve_%s_%s" % (name, k)] = v
    for kind in AL_KINDS:
        for k, v in oracle_al(kind).items():
            data["al_%s_%s" % (kind, k)] = v
    for k, v in oracle_mpc().items():
        data["mpc_%s" % k] = v
    for constrained in (False, True):
        for k, v in oracle_lq12(constrained).items():
            data["lq12_%s_%s" % ("al" if constrained else "lq", k)] = v
    for name in ("pendulum", "bicycle", "lq12"):
        for k, v in oracle_merit(name).items():
            data["merit_%s_%s" % (name, k)] = v
    for k, v in oracle_reg().items():
        data["tvlqr_%s" % k] = v
    for constrained in (False, True):
        for k, v in oracle_quad12(constrained).items():
            data["quad12_%s_%s" % ("al" if constrained else "lq", k)] = v
        for k, v in oracle_quad4(constrained).items():
            data["quad4_%s_%s" % ("al" if constrained else "lq", k)] = v
    for name in ("quad12", "quad4"):
        for k, v in oracle_merit(name).items():
            data["merit_%s_%s" % (name, k)] = v
    return data


if __name__ == "__main__":
    oracle.build()
    d = generate()
    np.savez_compressed(OUT, **d)
    print("wrote %s: %d arrays, %.1f KB" % (OUT, len(d), os.path.getsize(OUT) / 1e3))
