"""The HIP path (through the C ABI) against the committed dense golden fixtures -- no oracle, no gcc on the box:
tests/golden/dense_fixtures.npz holds the CPU oracle's outputs for every BASELINE.json config at reduced batch
(tests/golden/make_dense_fixtures.py; the inputs are regenerated from tests/problems.py's seeded streams).
Tolerances are the ones DESIGN.md section 2 states per plan.  Needs an MI355X."""
import os

import numpy as np
import pytest

import altro_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mk():
    from tests import golden_cases   # problem definitions only: nothing there imports the oracle
    return golden_cases


@pytest.fixture(scope="module")
def dense():
    return np.load(os.path.join(ROOT, "tests", "golden", "dense_fixtures.npz"))


def relerr(a, b):
    """Largest error relative to the scale of ITS OWN block: for arrays [problem][knot point][block...] the maximum over
    (problem, knot point) of max|a - b| / max(1, max|b|) taken per block -- a large P_k of one knot point does not lend its
    scale to the small entries of another (a whole-array scale made 1e-9 mean 1e-6 absolute there)."""
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    if a.ndim >= 3:
        ax = tuple(range(2, a.ndim))
        return float((np.abs(a - b).max(axis=ax) / np.maximum(1.0, np.abs(b).max(axis=ax))).max())
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


@pytest.mark.parametrize("name,dtype,flags,tol", [
    ("c1", altro_amd.F64, 0, 1e-9),                       # plan MFMA16 fp64: K, d <= 1e-8 absolute (north star), all 1e-9 relative
    ("c4", altro_amd.F32, 0, 5e-7),                       # fp32 storage, fp64 tiles
    ("c4", altro_amd.F32, altro_amd.F32_PURE, 2e-5),      # pure fp32: what tests/test_gpu_parity.py holds at N = 512
    ("c2shape", altro_amd.F64, 0, 0.0),                   # plan LANE: bit-identical
    ("c3shape", altro_amd.F64, 0, 0.0),
])
def test_tvlqr_against_dense_fixture(mk, dense, name, dtype, flags, tol):
    pr = mk.tvlqr_problem(name)
    assert mk.checksum(pr) == dense["tvlqr_%s_input_checksum" % name]
    batch = pr["A"].shape[0]
    bt = altro_amd.Batch(pr["N"], pr["n"], pr["m"], batch, dtype=dtype, flags=flags)
    bt.set_dynamics(pr["A"], pr["B"], pr["f"]); bt.set_cost(pr["Q"], pr["R"], pr["H"], pr["q"], pr["r"])
    bt.set_initial_state(pr["x0"]); bt.sweep()
    assert (bt.get("status") == -1).all()
    knots = mk.P_KNOTS.get(name)
    for k in ("K", "d", "P", "p", "x", "u", "y"):
        got = bt.get(k)
        if k == "P" and knots is not None:
            got = got[:, knots]
        ref = dense["tvlqr_%s_%s" % (name, k)]
        if tol == 0.0:
            assert np.array_equal(got, ref), (name, k)
        else:
            assert relerr(got, ref) < tol, (name, k, relerr(got, ref))
    if name == "c1":   # the north star's own statement: gains within 1e-8 of the CPU path, absolute
        assert np.abs(bt.get("K") - dense["tvlqr_c1_K"]).max() < 1e-8 and np.abs(bt.get("d") - dense["tvlqr_c1_d"]).max() < 1e-8
    dv_tol = {0.0: 1e-15, 1e-9: 1e-9, 5e-7: 1e-4, 2e-5: 2e-3}[tol]
    assert relerr(bt.get("delta_V"), dense["tvlqr_%s_dV" % name]) <= dv_tol
    bt.close()


@pytest.mark.parametrize("name,tol", [("di_n10", 1e-11), ("di_n50", 1e-11), ("pendulum", 2e-7), ("bicycle", 5e-5)])
def test_solves_against_dense_fixture(mk, dense, name, tol):
    """Whole device solves (plan LANE, device models) land on the oracle's trajectories with the oracle's iteration
    counts: configs[0] at N = 10 and N = 50 (double_integrator_test.cpp:69-85), the pendulum and the bicycle."""
    c = mk.solve_case(name)
    x0s = np.asarray(c["x0s"], dtype=float)
    bt = altro_amd.Batch(c["N"], c["n"], c["m"], x0s.shape[0])
    assert bt.plan == altro_amd.PLAN_LANE
    bt.set_model(c["model"], c["h"])
    bt.set_tracking_cost(np.stack([c["Qd"], c["Qfd"]]).astype(float), np.asarray(c["Rd"], dtype=float)[None],
                         np.stack([c["xf"], c["xf"]]).astype(float), np.zeros((1, c["m"])), k_stride_zero=True,
                         batch_stride_zero=True)
    bt.set_initial_state(x0s)
    bt.set_input_guess(np.asarray(c["u0"], dtype=float)[None, None], k_stride_zero=True, batch_stride_zero=True)
    res = bt.ilqr_solve(iterations_max=c["itmax"], use_backtracking=bool(c["bt"]))
    assert res["status"].tolist() == dense["solve_%s_status" % name].tolist()
    assert res["iterations"].tolist() == dense["solve_%s_iterations" % name].tolist()
    x, u = bt.get_nominal()
    assert relerr(x, dense["solve_%s_x" % name]) < tol
    assert relerr(u, dense["solve_%s_u" % name]) < tol * 10
    bt.close()


# ---- round 3: every remaining row of SURVEY.md section 8 (a) / (f) replays from the fixtures, no oracle on the box -------------
def _di_constrained(mk, kind):
    c = mk.al_case(mk.load_kats(), kind)
    bt = altro_amd.Batch(c["N"], c["n"], c["m"], c["x0s"].shape[0])
    assert bt.plan == altro_amd.PLAN_LANE
    bt.set_model(altro_amd.MODEL_DOUBLE_INTEGRATOR, c["h"])
    Qd = np.full(c["n"], c["Q"]); Rd = np.full(c["m"], c["R"])
    bt.set_tracking_cost(np.stack([Qd, Qd]), Rd[None], np.stack([c["xf"], c["xf"]]), np.zeros((1, c["m"])), k_stride_zero=True,
                         batch_stride_zero=True)
    bt.set_initial_state(c["x0s"])
    bt.set_input_guess(np.zeros((1, 1, c["m"])), k_stride_zero=True, batch_stride_zero=True)
    for (k0, k1, cone, G, g) in c["blocks"]:
        bt.add_linear_constraint(k0, k1, cone, G, g)
    return bt, c


@pytest.mark.parametrize("kind", ["goal", "bounds", "soc"])
def test_constrained_solves_against_dense_fixture(mk, dense, kind):
    """Row f2: the reference's constrained double integrator (its own start: 3 / 5 / 9 iterations,
    double_integrator_test.cpp:255, 374, 491) and four shifted starts, per problem: status, iterations, feasibility,
    trajectory."""
    bt, c = _di_constrained(mk, kind)
    res = bt.ilqr_solve(penalty_initial=c["penalty_initial"], penalty_scaling=c["penalty_scaling"], iterations_max=c["itmax"])
    assert res["status"].tolist() == dense["al_%s_status" % kind].tolist()
    assert res["iterations"].tolist() == dense["al_%s_iterations" % kind].tolist()
    assert res["iterations"][0] == c["iterations"] and res["status"][0] == 0
    tol = 1e-8 if kind != "soc" else 1e-6
    ok = res["status"] == 0
    feas = dense["al_%s_feasibility" % kind]
    assert (np.abs(res["feasibility"][ok] - feas[ok]) <= 1e-9 + 1e-3 * feas[ok]).all()
    x, u = bt.get("x"), bt.get("u")
    assert np.abs(x[ok] - dense["al_%s_x" % kind][ok]).max() < tol
    assert np.abs(u[ok] - dense["al_%s_u" % kind][ok]).max() < tol * 10
    bt.close()


def test_mpc_steps_against_dense_fixture(mk, dense):
    """Row f3: the receding-horizon loop of bicycle_test.cpp:302-337 for three vehicles, four steps; the plant's states come
    from the fixture (the oracle's integrator), the device solves, shifts and updates: u_0 and the iteration counts."""
    c = mk.mpc_case()
    N, n, m = c["N"], c["n"], c["m"]
    bt = altro_amd.Batch(N, n, m, 3)
    bt.set_model(altro_amd.MODEL_BICYCLE, c["h"])
    bt.set_tracking_cost(np.full((1, N + 1, n), c["QD"]), np.full((1, N, m), c["RD"]), c["x_ref"][None, :N + 1], c["u_ref"][None, :N],
                         batch_stride_zero=True)
    bt.add_linear_constraint(0, N, altro_amd.CONE_INEQUALITY, c["G"], c["g"])
    bt.set_initial_state(c["x0s"])
    bt.set_input_guess(c["u0"][None, None], k_stride_zero=True, batch_stride_zero=True)
    same = 0
    for it in range(c["nsim"]):
        res = bt.ilqr_solve(iterations_max=80, use_backtracking=True)
        assert (res["status"] == 0).all() and (dense["mpc_status"][:, it] == 0).all()
        _, u = bt.get_knot(0)
        assert np.abs(u - dense["mpc_u0"][:, it]).max() < 5e-5
        same += int((res["iterations"] == dense["mpc_iterations"][:, it]).sum())
        q, cc = mk.mpc_linear_costs(c, it + 1)
        bt.update_linear_costs(q[None], None, cc[None], 0, N, batch_stride_zero=True)
        bt.set_initial_state(dense["mpc_x_next"][:, it])
        bt.shift_trajectory()
    assert same >= 3 * c["nsim"] - 1     # sin / cos last-ulp differences may move a convergence test by one sweep
    bt.close()


def _lq12(mk, constrained):
    c = mk.lq12_case(constrained)
    p = c["p"]
    bt = altro_amd.Batch(c["N"], 12, 4, p["x0"].shape[0])
    assert bt.plan == altro_amd.PLAN_MFMA16
    bt.set_dynamics(p["A"], p["B"], p["f"])
    bt.set_tracking_cost(p["Qd"], p["Rd"], p["xref"], p["uref"])
    bt.set_initial_state(p["x0"])
    bt.set_input_guess(p["u0"])
    for (k0, k1, cone, G, g) in c["blocks"]:
        bt.add_linear_constraint(k0, k1, cone, G, g)
    return bt, c


@pytest.mark.parametrize("constrained", [False, True])
def test_12x4_ilqr_solves_against_dense_fixture(mk, dense, constrained):
    """Rows a5-a12 (+ f2) on plan MFMA16: whole (12, 4) solves, per problem."""
    bt, c = _lq12(mk, constrained)
    res = bt.ilqr_solve(iterations_max=c["itmax"], penalty_initial=1.0, penalty_scaling=10.0)
    tag = "al" if constrained else "lq"
    assert res["status"].tolist() == dense["lq12_%s_status" % tag].tolist()
    assert res["iterations"].tolist() == dense["lq12_%s_iterations" % tag].tolist()
    ok = res["status"] == 0
    assert ok.sum() >= 3
    x, u = bt.get_nominal()
    tol = 1e-7 if constrained else 1e-9
    assert np.abs(x[ok] - dense["lq12_%s_x" % tag][ok]).max() < tol
    assert np.abs(u[ok] - dense["lq12_%s_u" % tag][ok]).max() < tol * 10
    bt.close()


@pytest.mark.parametrize("name", ["pendulum", "bicycle", "lq12"])
def test_merit_expansion_stationarity_against_dense_fixture(mk, dense, name):
    """Rows a5, a7-a9, a11: MeritFunction with derivative at alpha = 0, 0.35, 1 after rollout / accept / expand / backward:
    phi, phi', candidate x_, u_, y_, the refreshed lx, lu (A, B for the device models), stationarity."""
    na = len(mk.MERIT_ALPHAS)
    if name == "lq12":
        c = mk.lq12_case(False)
        p = {k: (np.repeat(v, na, axis=0) if v is not None else None) for k, v in c["p"].items()}
        bt = altro_amd.Batch(c["N"], 12, 4, p["x0"].shape[0])
        bt.set_dynamics(p["A"], p["B"], p["f"]); bt.set_tracking_cost(p["Qd"], p["Rd"], p["xref"], p["uref"])
        bt.set_initial_state(p["x0"]); bt.set_input_guess(p["u0"])
        nprob, tol = c["p"]["x0"].shape[0], 1e-9
    else:
        c = mk.merit_case(name)
        x0s = np.repeat(np.asarray(c["x0s"], dtype=float), na, axis=0)
        bt = altro_amd.Batch(c["N"], c["n"], c["m"], x0s.shape[0])
        bt.set_model(c["model"], c["h"])
        bt.set_tracking_cost(np.stack([c["Qd"], c["Qfd"]]).astype(float), np.asarray(c["Rd"], dtype=float)[None],
                             np.stack([c["xf"], c["xf"]]).astype(float), np.zeros((1, c["m"])), k_stride_zero=True, batch_stride_zero=True)
        bt.set_initial_state(x0s)
        bt.set_input_guess(np.asarray(c["u0"], dtype=float)[None, None], k_stride_zero=True, batch_stride_zero=True)
        nprob, tol = len(c["x0s"]), 1e-10
    bt.open_loop_rollout(); bt.accept(); bt.expand(); bt.backward()
    assert (bt.get("status") == -1).all()
    assert relerr(bt.get("K"), dense["merit_%s_K" % name]) < tol
    alphas = np.tile(np.array(mk.MERIT_ALPHAS), nprob)
    phi, dphi = bt.merit(alphas)
    assert relerr(phi, dense["merit_%s_phi" % name]) < tol * 10
    assert relerr(dphi, dense["merit_%s_dphi" % name]) < tol * 100
    for k in ("x", "u", "y"):
        assert relerr(bt.get(k), dense["merit_%s_%s" % (name, k)]) < tol * 10, k
    if name != "lq12":      # (plan MFMA16's dynamics are data: its expansion is the cost gradient, checked through stationarity)
        A, B, lx, lu = bt.get_expansion()
        for k, v in (("A", A), ("B", B), ("lx", lx), ("lu", lu)):
            assert relerr(v, dense["merit_%s_%s" % (name, k)]) < tol * 10, k
    st = bt.stationarity()
    ref = dense["merit_%s_stationarity" % name]
    assert (np.abs(st - ref) <= 1e-8 * np.maximum(1.0, np.abs(ref))).all()
    bt.close()


def _quad_batch(mk, name, constrained, repeat=1):
    if name == "quad12":
        c = mk.quad12_case(constrained)
        p = {k: (np.repeat(v, repeat, axis=0) if v is not None else None) for k, v in c["p"].items()}
        bt = altro_amd.Batch(c["N"], 12, 4, p["x0"].shape[0])
        assert bt.plan == altro_amd.PLAN_MFMA16
        bt.set_dynamics(p["A"], p["B"], p["f"])
        bt.set_quadratic_cost(p["Q"], p["R"], p["H"], p["q"], p["r"], p["c"])
        bt.set_initial_state(p["x0"]); bt.set_input_guess(p["u0"])
        nprob = c["p"]["x0"].shape[0]
    else:
        c = mk.quad4_case(constrained)
        cost = {k: np.repeat(v, repeat, axis=0) for k, v in c["cost"].items()}
        x0s = np.repeat(np.asarray(c["x0s"], dtype=float), repeat, axis=0)
        bt = altro_amd.Batch(c["N"], c["n"], c["m"], x0s.shape[0])
        assert bt.plan == altro_amd.PLAN_LANE
        bt.set_model(c["model"], c["h"])
        bt.set_quadratic_cost(cost["Q"], cost["R"], cost["H"], cost["q"], cost["r"], cost["c"])
        bt.set_initial_state(x0s)
        bt.set_input_guess(np.asarray(c["u0"], dtype=float)[None, None], k_stride_zero=True, batch_stride_zero=True)
        nprob = len(c["x0s"])
    for (k0, k1, cone, G, g) in c["blocks"]:
        bt.add_linear_constraint(k0, k1, cone, G, g)
    return bt, c, nprob


@pytest.mark.parametrize("name", ["quad12", "quad4"])
@pytest.mark.parametrize("constrained", [False, True])
def test_dense_quadratic_cost_solves_against_dense_fixture(mk, dense, name, constrained):
    """Row a9 (knotpoint_data.cpp:616-708 in the device loop): whole solves with the dense cost Q, R, H != 0 of
    ALTROSolver::SetQuadraticCost, plan MFMA16 at (12, 4) and plan LANE at (4, 2), with and without constraint blocks."""
    bt, c, nprob = _quad_batch(mk, name, constrained)
    res = bt.ilqr_solve(iterations_max=c["itmax"], penalty_initial=1.0, penalty_scaling=10.0)
    tag = "%s_%s" % (name, "al" if constrained else "lq")
    assert res["status"].tolist() == dense[tag + "_status"].tolist()
    assert res["iterations"].tolist() == dense[tag + "_iterations"].tolist()
    x, u = bt.get_nominal()
    tol = (1e-7 if constrained else 1e-9) if name == "quad12" else 2e-7
    assert np.abs(x - dense[tag + "_x"]).max() < tol
    assert np.abs(u - dense[tag + "_u"]).max() < tol * 10
    if constrained:
        ref = dense[tag + "_feasibility"]
        assert (np.abs(res["feasibility"] - ref) <= 1e-9 + 1e-3 * ref).all()
    bt.close()


@pytest.mark.parametrize("name", ["quad12", "quad4"])
def test_dense_quadratic_cost_merit_rows_against_dense_fixture(mk, dense, name):
    """Row a9 in one merit evaluation with derivative at alpha = 0, 0.35, 1: phi (with u'Hx), phi', candidates, the refreshed
    lx = Qx + H'u + q and lu = Ru + Hx + r (plan LANE: read back; plan MFMA16: through the stationarity), gains from lux = H."""
    na = len(mk.MERIT_ALPHAS)
    bt, c, nprob = _quad_batch(mk, name, False, repeat=na)
    tol = 1e-9 if name == "quad12" else 1e-10
    bt.open_loop_rollout(); bt.accept(); bt.expand(); bt.backward()
    assert (bt.get("status") == -1).all()
    assert relerr(bt.get("K"), dense["merit_%s_K" % name]) < tol * 10
    phi, dphi = bt.merit(np.tile(np.array(mk.MERIT_ALPHAS), nprob))
    assert relerr(phi, dense["merit_%s_phi" % name]) < tol * 10
    assert relerr(dphi, dense["merit_%s_dphi" % name]) < tol * 100
    for k in ("x", "u", "y"):
        assert relerr(bt.get(k), dense["merit_%s_%s" % (name, k)]) < tol * 10, k
    if name == "quad4":
        A, B, lx, lu = bt.get_expansion()
        for k, v in (("A", A), ("B", B), ("lx", lx), ("lu", lu)):
            assert relerr(v, dense["merit_%s_%s" % (name, k)]) < tol * 10, k
    st = bt.stationarity()
    ref = dense["merit_%s_stationarity" % name]
    assert (np.abs(st - ref) <= 1e-8 * np.maximum(1.0, np.abs(ref))).all()
    bt.close()


def test_regularised_backward_pass_against_dense_fixture(mk, dense):
    """Row f4 / a1's reg argument (tvlqr.cpp:159-164): reg > 0 on plan LANE (bit-identical), and the failing knot point of
    an indefinite R as the status."""
    pr = mk.tvlqr_problem(mk.REG_CASE["name"])
    batch = pr["A"].shape[0]
    bt = altro_amd.Batch(pr["N"], pr["n"], pr["m"], batch)
    bt.set_dynamics(pr["A"], pr["B"], pr["f"]); bt.set_cost(pr["Q"], pr["R"], pr["H"], pr["q"], pr["r"])
    bt.set_initial_state(pr["x0"])
    bt.backward(mk.REG_CASE["reg"])
    assert (bt.get("status") == -1).all()
    for k in ("K", "d", "P", "p"):
        assert np.array_equal(bt.get(k), dense["tvlqr_reg_%s" % k]), k
    assert relerr(bt.get("delta_V"), dense["tvlqr_reg_dV"]) <= 1e-15
    R = pr["R"].copy(); R[1] *= -1.0
    bt.set_cost(pr["Q"], R, pr["H"], pr["q"], pr["r"])
    bt.backward(0.0)
    assert bt.get("status").tolist() == dense["tvlqr_bad_status"].tolist()
    assert np.array_equal(bt.get("K")[0], dense["tvlqr_bad_K0"])
    bt.close()
