"""the DENSE quadratic cost of ALTROSolver::SetQuadraticCost (altro_solver.cpp:118-136 ->
KnotPointData::SetQuadraticCost, knotpoint_data.cpp:64-85) inside the device iLQR loop, on both plans, against the oracle's
restatement of CalcOriginalCost / Gradient / Hessian (knotpoint_data.cpp:616-708: value with u'Hx, gradient Qx + H'u + q and
Ru + Hx + r, Hessian blocks with lux = H) -- `oracle.ILQR(cost_kind=COST_QUADRATIC)`.  H != 0 everywhere.

Tolerances are the plans' existing ones: plan LANE 1e-13 (double integrator) / 1e-10 (trigonometric models) on one merit
evaluation, plan MFMA16 phi 1e-11, phi' 1e-9, candidates 1e-10; whole solves: same status / iterations per problem,
trajectories 1e-9 (LQ), 1e-7 (with constraint blocks)."""
import numpy as np
import pytest

import altro_amd
from oracle import oracle
from tests import problems

pytestmark = pytest.mark.gpu


# ---- plan LANE: device models ---------------------------------------------------------------------------------------------
LANE_CASES = {
    "double_integrator": dict(model=altro_amd.MODEL_DOUBLE_INTEGRATOR, okind=oracle.MODEL_DI, N=12, n=4, m=2, h=np.float32(0.25),
                              u0=[0.05, -0.02], x0=lambda b: np.tile([1.0, 2.0, 0, 0], (b, 1)) + 0.1 * np.arange(b)[:, None], dim=2, tol=1e-13),
    "bicycle": dict(model=altro_amd.MODEL_BICYCLE, okind=oracle.MODEL_BICYCLE, N=20, n=4, m=2, h=np.float32(0.1),
                    u0=[0.5, 0.0], x0=lambda b: 0.02 * (np.arange(b * 4).reshape(b, 4) % 7 - 3) / 3.0, dim=0, tol=1e-10),
    "pendulum": dict(model=altro_amd.MODEL_PENDULUM, okind=oracle.MODEL_PENDULUM, N=25, n=2, m=1, h=np.float32(0.05),
                     u0=[0.1], x0=lambda b: np.stack([np.linspace(-0.5, 0.5, b), np.zeros(b)], 1), dim=0, tol=1e-10),
}


def lane_hip(c, cost, x0s, dtype=altro_amd.F64, source=None):
    bt = altro_amd.Batch(c["N"], c["n"], c["m"], x0s.shape[0], dtype=dtype)
    assert bt.plan == altro_amd.PLAN_LANE
    if source is None:
        bt.set_model(c["model"], c["h"])
    else:
        bt.set_model_source(source, c["h"])
    bt.set_quadratic_cost(cost["Q"], cost["R"], cost["H"], cost["q"], cost["r"], cost["c"])
    bt.set_initial_state(x0s)
    bt.set_input_guess(np.asarray(c["u0"], dtype=float)[None, None], k_stride_zero=True, batch_stride_zero=True)
    return bt


def set_oracle_cost(s, cost, b, N):
    for k in range(N + 1):
        kk = min(k, N - 1)
        s.L.oracle_ilqr_set_quadratic_cost(s.h, k, np.ascontiguousarray(cost["Q"][b, k]), np.ascontiguousarray(cost["R"][b, kk]).ctypes.data,
                                           np.ascontiguousarray(cost["H"][b, kk]).ctypes.data, np.ascontiguousarray(cost["q"][b, k]),
                                           np.ascontiguousarray(cost["r"][b, kk]).ctypes.data, float(cost["c"][b, k]))


def lane_oracle(c, cost, b, x0):
    s = oracle.ILQR(c["N"], c["n"], c["m"], c["h"], oracle.DYN_MODEL, c["okind"], model_dim=c["dim"], cost_kind=oracle.COST_QUADRATIC)
    set_oracle_cost(s, cost, b, c["N"])
    s.L.oracle_ilqr_set_initial_state(s.h, np.ascontiguousarray(x0, dtype=float))
    s.L.oracle_ilqr_initialize(s.h)
    for k in range(c["N"]):
        s.L.oracle_ilqr_set_input(s.h, k, np.ascontiguousarray(c["u0"], dtype=float))
    return s


@pytest.mark.parametrize("name", list(LANE_CASES))
def test_lane_merit_and_expansion_parity(name):
    """One merit evaluation and the expansion with the dense cost: phi, phi', candidates, A, B, lx (with H'u), lu (with Hx), and
    the gains the backward sweep forms from lxx = Q, luu = R, lux = H."""
    c = LANE_CASES[name]
    batch = 70
    x0s = c["x0"](batch)
    cost = problems.quadratic_cost(batch, c["N"], c["n"], c["m"])
    bt = lane_hip(c, cost, x0s)
    bt.open_loop_rollout(); bt.accept(); bt.expand(); bt.backward()
    assert (bt.get("status") == -1).all()
    alphas = np.linspace(0.0, 1.5, batch)
    phi, dphi = bt.merit(alphas)
    xc, uc, yc = bt.get("x"), bt.get("u"), bt.get("y")
    A, B, lx, lu = bt.get_expansion()
    tol = c["tol"]
    for b in [0, 1, 17, 69]:
        s = lane_oracle(c, cost, b, x0s[b])
        s.L.oracle_ilqr_open_loop_rollout(s.h); s.L.oracle_ilqr_copy_trajectory(s.h)
        s.L.oracle_ilqr_calc_dynamics_expansions(s.h); s.L.oracle_ilqr_calc_cost_gradient(s.h)
        s.L.oracle_ilqr_calc_expansions(s.h)
        assert s.L.oracle_ilqr_backward_pass(s.h) == -1
        np.testing.assert_allclose(bt.get("K")[b], s.get("K"), rtol=tol * 100, atol=tol * 100)
        p_ref, dp_ref = s.merit(alphas[b])
        assert abs(phi[b] - p_ref) <= tol * 10 * max(1.0, abs(p_ref)), (b, phi[b], p_ref)
        assert abs(dphi[b] - dp_ref) <= tol * 100 * max(1.0, abs(dp_ref)), (b, dphi[b], dp_ref)
        np.testing.assert_allclose(xc[b], s.get("x_cand"), rtol=tol * 10, atol=tol * 10)
        np.testing.assert_allclose(uc[b], s.get("u_cand"), rtol=tol * 10, atol=tol * 10)
        np.testing.assert_allclose(yc[b], s.get("y_cand"), rtol=tol * 100, atol=tol * 100)
        np.testing.assert_allclose(lx[b], s.get("lx"), rtol=tol * 10, atol=tol * 10)
        np.testing.assert_allclose(lu[b], s.get("lu"), rtol=tol * 10, atol=tol * 10)
        np.testing.assert_allclose(A[b], s.get("A"), rtol=tol, atol=tol)


@pytest.mark.parametrize("name,constrained", [("double_integrator", False), ("double_integrator", True), ("bicycle", False),
                                               ("bicycle", True), ("pendulum", False)])
def test_lane_whole_solves(name, constrained):
    """Whole AL-iLQR solves with the dense cost, with and without constraint blocks: status, iterations, trajectory per problem."""
    c = LANE_CASES[name]
    batch = 33
    x0s = c["x0"](batch)
    N, n, m = c["N"], c["n"], c["m"]
    cost = problems.quadratic_cost(batch, N, n, m)
    bt = lane_hip(c, cost, x0s)
    blocks = []
    if constrained:
        w = n + m
        Gb = np.zeros((2 * m, w)); Gb[:m, n:] = np.eye(m); Gb[m:, n:] = -np.eye(m)
        blocks = [(0, N - 1, altro_amd.CONE_INEQUALITY, Gb, np.full(2 * m, 0.4))]
        for (k0, k1, cone, G, g) in blocks:
            bt.add_linear_constraint(k0, k1, cone, G, g)
    res = bt.ilqr_solve(iterations_max=40)
    x, u = bt.get_nominal()
    nconv = 0
    for b in [0, 16, 32]:
        s = lane_oracle(c, cost, b, x0s[b])
        for (k0, k1, cone, G, g) in blocks:
            for k in range(k0, k1 + 1):
                s.add_linear_constraint(k, cone, G, g)
        if blocks:
            s.L.oracle_ilqr_initialize(s.h)
            for k in range(N):
                s.L.oracle_ilqr_set_input(s.h, k, np.ascontiguousarray(c["u0"], dtype=float))
            s.set_penalty(1.0, 10.0)
        s.L.oracle_ilqr_set_options(s.h, 40, 1e-4, 1e-4, 1e-8, 0)
        status, iters, log = s.solve()
        assert res["status"][b] == status and res["iterations"][b] == iters, (b, res["status"][b], status, res["iterations"][b], iters)
        if status != 0:
            continue
        nconv += 1
        np.testing.assert_allclose(x[b], s.get("x"), rtol=2e-7, atol=2e-7)
        np.testing.assert_allclose(u[b], s.get("u"), rtol=2e-6, atol=2e-6)
        if constrained:
            assert np.abs(u[b]).max() <= 0.4 + 2e-4
    assert nconv >= 2


def test_lane_sequenced_loop_and_cost_switch():
    """A dense-cost handle runs the launch-sequenced loop (the one-launch kernel carries the diagonal cost); going back to a
    tracking cost on the same handle restores the diagonal records and the one-launch path, with the results of a fresh handle."""
    c = LANE_CASES["bicycle"]
    batch = 20
    x0s = c["x0"](batch)
    cost = problems.quadratic_cost(batch, c["N"], c["n"], c["m"])
    bt = lane_hip(c, cost, x0s)
    r1 = bt.ilqr_solve(iterations_max=30)
    x1 = bt.get_nominal()[0].copy()
    Qd, Rd, xf = np.full(4, 1e-2), np.full(2, 1e-3), np.array([1.0, 2.0, np.pi / 2, 0.0])

    def tracking(b_):
        b_.set_tracking_cost(np.stack([Qd, 10.0 * np.ones(4)]), Rd[None], np.stack([xf, xf]), np.zeros((1, 2)), k_stride_zero=True, batch_stride_zero=True)
        b_.set_initial_state(x0s)
        b_.set_input_guess(np.asarray(c["u0"], dtype=float)[None, None], k_stride_zero=True, batch_stride_zero=True)
    tracking(bt)
    r2 = bt.ilqr_solve(iterations_max=30)
    fresh = altro_amd.Batch(c["N"], 4, 2, batch)
    fresh.set_model(c["model"], c["h"])
    tracking(fresh)
    r3 = fresh.ilqr_solve(iterations_max=30)
    assert np.array_equal(r2["iterations"], r3["iterations"]) and np.array_equal(bt.get_nominal()[0], fresh.get_nominal()[0])
    # ... and back to the dense cost: the first solve's trajectory again
    bt.set_quadratic_cost(cost["Q"], cost["R"], cost["H"], cost["q"], cost["r"], cost["c"])
    bt.set_initial_state(x0s)
    bt.set_input_guess(np.asarray(c["u0"], dtype=float)[None, None], k_stride_zero=True, batch_stride_zero=True)
    r4 = bt.ilqr_solve(iterations_max=30)
    assert np.array_equal(r1["iterations"], r4["iterations"]) and np.array_equal(bt.get_nominal()[0], x1)


PENDULUM_SRC = r"""
template <typename T> __device__ void altro_user_dynamics(const T* x, const T* u, T* xdot) {
  const T mass = T(1.0), length = T(0.5), b = T(0.1), lc = T(0.5), I = T(0.25), g = T(9.81);
  xdot[0] = x[1];
  xdot[1] = (u[0] - mass * g * lc * sin(x[0]) - b * x[1]) / I;
}
template <typename T> __device__ void altro_user_jacobian(const T* x, const T* u, T* J) {
  const T mass = T(1.0), b = T(0.1), lc = T(0.5), I = T(0.25), g = T(9.81);
  J[0] = T(0); J[1] = -mass * g * lc * cos(x[0]) / I;
  J[2] = T(1); J[3] = -b / I;
  J[4] = T(0); J[5] = T(1) / I;
}
"""


def test_lane_user_model_with_dense_cost():
    """A run-time compiled model (altro_hip_set_model_source) with the dense cost: the module is built with the cost-reading
    kernels instantiated for it; the solve agrees with the oracle's pendulum (same model constants, test_utils.cpp:43-47)."""
    c = LANE_CASES["pendulum"]
    batch = 9
    x0s = c["x0"](batch)
    cost = problems.quadratic_cost(batch, c["N"], 2, 1)
    bt = lane_hip(c, cost, x0s, source=PENDULUM_SRC)
    res = bt.ilqr_solve(iterations_max=40)
    x, u = bt.get_nominal()
    ref = lane_hip(c, cost, x0s)
    res_ref = ref.ilqr_solve(iterations_max=40)
    assert np.array_equal(res["iterations"], res_ref["iterations"])
    np.testing.assert_allclose(x, ref.get_nominal()[0], rtol=1e-9, atol=1e-9)
    for b in [0, 8]:
        s = lane_oracle(c, cost, b, x0s[b])
        s.L.oracle_ilqr_set_options(s.h, 40, 1e-4, 1e-4, 1e-8, 0)
        status, iters, log = s.solve()
        assert res["status"][b] == status and res["iterations"][b] == iters
        np.testing.assert_allclose(x[b], s.get("x"), rtol=2e-7, atol=2e-7)


# ---- plan MFMA16: dynamics as data ------------------------------------------------------------------------------------------
N16 = 24


def mf_problem(batch, n=12, m=4, N=N16, with_f=True):
    p = problems.ilqr12x4_problem(batch, N, with_f, n=n, m=m)
    p.update(problems.quadratic_cost(batch, N, n, m))
    return p


def mf_hip(p, n=12, m=4, N=N16, dtype=altro_amd.F64):
    bt = altro_amd.Batch(N, n, m, p["x0"].shape[0], dtype=dtype, plan=altro_amd.PLAN_MFMA16)
    bt.set_dynamics(p["A"], p["B"], p["f"])
    bt.set_quadratic_cost(p["Q"], p["R"], p["H"], p["q"], p["r"], p["c"])
    bt.set_initial_state(p["x0"])
    bt.set_input_guess(p["u0"])
    return bt


def mf_oracle(p, b, n=12, m=4, N=N16, blocks=()):
    s = oracle.ILQR(N, n, m, 0.01, oracle.DYN_LINEAR, cost_kind=oracle.COST_QUADRATIC)
    f = p["f"][b] if p["f"] is not None else None
    s.L.oracle_ilqr_set_linear_dynamics(s.h, np.ascontiguousarray(p["A"][b]), np.ascontiguousarray(p["B"][b]),
                                        None if f is None else np.ascontiguousarray(f).ctypes.data)
    set_oracle_cost(s, p, b, N)
    s.L.oracle_ilqr_set_initial_state(s.h, np.ascontiguousarray(p["x0"][b]))
    for (k0, k1, cone, G, g) in blocks:
        for k in range(k0, k1 + 1):
            s.add_linear_constraint(k, cone, G, g)
    s.L.oracle_ilqr_initialize(s.h)
    for k in range(N):
        s.L.oracle_ilqr_set_input(s.h, k, np.ascontiguousarray(p["u0"][b, k]))
    return s


@pytest.mark.parametrize("with_f", [False, True])
def test_mfma16_merit_parity(with_f):
    """solver.cpp:273-355 on the (12, 4) tile with the dense cost: phi, phi', candidate x_ / u_ / y_, the stationarity of the
    candidate (which reads the refreshed lx, lu: knotpoint_data.cpp:659-668), odd batch (a wave with one problem)."""
    batch = 9
    p = mf_problem(batch, with_f=with_f)
    bt = mf_hip(p)
    bt.open_loop_rollout(); bt.accept(); bt.expand(); bt.backward()
    assert (bt.get("status") == -1).all()
    alphas = np.linspace(0.0, 1.2, batch)
    phi, dphi = bt.merit(alphas)
    xc, uc, yc = bt.get("x"), bt.get("u"), bt.get("y")
    st = bt.stationarity()
    for b in [0, 4, 8]:
        s = mf_oracle(p, b)
        s.L.oracle_ilqr_open_loop_rollout(s.h); s.L.oracle_ilqr_copy_trajectory(s.h)
        s.L.oracle_ilqr_calc_dynamics_expansions(s.h); s.L.oracle_ilqr_calc_cost_gradient(s.h)
        s.L.oracle_ilqr_calc_expansions(s.h)
        assert s.L.oracle_ilqr_backward_pass(s.h) == -1
        np.testing.assert_allclose(bt.get("K")[b], s.get("K"), rtol=1e-9, atol=1e-9)
        p_ref, dp_ref = s.merit(alphas[b])
        assert abs(phi[b] - p_ref) <= 1e-11 * max(1.0, abs(p_ref)), (b, phi[b], p_ref)
        assert abs(dphi[b] - dp_ref) <= 1e-9 * max(1.0, abs(dp_ref)), (b, dphi[b], dp_ref)
        np.testing.assert_allclose(xc[b], s.get("x_cand"), rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(uc[b], s.get("u_cand"), rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(yc[b], s.get("y_cand"), rtol=1e-9, atol=1e-9)
        assert abs(st[b] - s.L.oracle_ilqr_stationarity(s.h)) <= 1e-8 * max(1.0, st[b])


@pytest.mark.parametrize("nn,mm", [(12, 4), (9, 3)])
def test_mfma16_lq_solve(nn, mm):
    """Whole solves of an LQ problem with the dense cost (alpha = 1, <= 3 sweeps), the exact tile shape and a padded one."""
    batch = 41
    p = mf_problem(batch, nn, mm)
    bt = mf_hip(p, nn, mm)
    res = bt.ilqr_solve(iterations_max=10)
    assert (res["status"] == 0).all() and (res["iterations"] <= 3).all()
    x, u = bt.get_nominal()
    for b in [0, 20, 40]:
        s = mf_oracle(p, b, nn, mm)
        s.L.oracle_ilqr_set_options(s.h, 10, 1e-4, 1e-4, 1e-8, 0)
        status, iters, log = s.solve()
        assert status == 0 and iters == res["iterations"][b]
        np.testing.assert_allclose(x[b], s.get("x"), rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(u[b], s.get("u"), rtol=1e-8, atol=1e-8)


@pytest.mark.parametrize("use_backtracking", [False, True])
def test_mfma16_constrained_solve(use_backtracking):
    """Input bounds + state half-spaces + an equality block with the dense cost: the AL loop's Hessian is W + rho G'J'JG
    (knotpoint_data.cpp:537-613 on top of :691-698).  Same status / iterations / feasibility per problem, trajectories 1e-7."""
    batch = 40
    p = mf_problem(batch)
    blocks = problems.ilqr12x4_constraint_blocks(N16)
    bt = mf_hip(p)
    for (k0, k1, cone, G, g) in blocks:
        bt.add_linear_constraint(k0, k1, cone, G, g)
    res = bt.ilqr_solve(iterations_max=60, penalty_initial=1.0, penalty_scaling=10.0, use_backtracking=use_backtracking)
    x, u = bt.get_nominal()
    nconv = 0
    for b in [0, 17, 39]:
        s = mf_oracle(p, b, blocks=blocks)
        s.set_penalty(1.0, 10.0)
        s.L.oracle_ilqr_set_options(s.h, 60, 1e-4, 1e-4, 1e-8, 1 if use_backtracking else 0)
        status, iters, log = s.solve()
        assert res["status"][b] == status and res["iterations"][b] == iters, (b, res["status"][b], status, res["iterations"][b], iters)
        if status != 0:
            continue
        nconv += 1
        assert abs(res["feasibility"][b] - log[iters - 1, 6]) <= 1e-9 + 1e-3 * log[iters - 1, 6]
        np.testing.assert_allclose(x[b], s.get("x"), rtol=1e-7, atol=1e-7)
        np.testing.assert_allclose(u[b], s.get("u"), rtol=1e-6, atol=1e-6)
        assert np.abs(u[b]).max() <= 0.3 + 2e-4 and abs(u[b][0, 0] - 0.05) < 2e-4
    assert nconv >= 2
    assert (res["dual_updates"] > 0).all()


def test_mfma16_update_linear_costs_dense():
    """UpdateLinearCosts (altro_solver.cpp:266-281) on a dense-cost handle: q, r, c move, Q, R, H stay (the MPC caller's pattern)."""
    batch = 6
    p = mf_problem(batch)
    bt = mf_hip(p)
    ors = [mf_oracle(p, b) for b in range(batch)]
    for s in ors:
        s.L.oracle_ilqr_set_options(s.h, 10, 1e-4, 1e-4, 1e-8, 0)
    for it in range(2):
        res = bt.ilqr_solve(iterations_max=10)
        x, u = bt.get_nominal()
        qnew = p["q"] + 0.05 * (it + 1)
        rnew = p["r"] - 0.02 * (it + 1)
        cnew = 0.1 * (it + 1) * np.ones((batch, N16 + 1))
        for b in range(batch):
            s = ors[b]
            status, iters, log = s.solve()
            assert res["status"][b] == status and res["iterations"][b] == iters, (it, b)
            np.testing.assert_allclose(x[b], s.get("x"), rtol=1e-9, atol=1e-9)
            for k in range(N16 + 1):
                s.L.oracle_ilqr_update_linear_costs(s.h, k, np.ascontiguousarray(qnew[b, k]).ctypes.data,
                                                    np.ascontiguousarray(rnew[b, k]).ctypes.data if k < N16 else None, float(cnew[b, k]))
        bt.update_linear_costs(qnew, None, cnew, 0, N16)            # a range that ends at the terminal knot point (q_N lives apart)
        bt.update_linear_costs(None, rnew, None, 0, N16 - 1)


def test_mfma16_dense_cost_fp32_storage():
    """fp32 records, fp64 arithmetic: the dense cost through the same kernels (S = float)."""
    batch = 10
    p = mf_problem(batch)
    bt = mf_hip(p, dtype=altro_amd.F32)
    res = bt.ilqr_solve(iterations_max=10, tol_stationarity=1e-3)
    x, u = bt.get_nominal()
    for b in [0, 9]:
        s = mf_oracle(p, b)
        s.L.oracle_ilqr_set_options(s.h, 10, 1e-4, 1e-4, 1e-8, 0)
        status, iters, log = s.solve()
        np.testing.assert_allclose(x[b], s.get("x"), rtol=2e-4, atol=2e-4)


def test_dense_cost_equals_tracking_cost_when_diagonal():
    """A diagonal cost handed over as dense blocks (H = 0) is the tracking cost: same iterations, trajectories to rounding, on
    both plans -- ties the new path to the one every other test pins."""
    batch = 12
    p = problems.ilqr12x4_problem(batch, N16, True)
    a = altro_amd.Batch(N16, 12, 4, batch)
    a.set_dynamics(p["A"], p["B"], p["f"]); a.set_tracking_cost(p["Qd"], p["Rd"], p["xref"], p["uref"])
    a.set_initial_state(p["x0"]); a.set_input_guess(p["u0"])
    ra = a.ilqr_solve(iterations_max=10)
    Q = np.zeros((batch, N16 + 1, 144)); R = np.zeros((batch, N16, 16))
    Q[:, :, ::13] = p["Qd"]; R[:, :, ::5] = p["Rd"]
    q = -p["Qd"] * p["xref"]; r = -p["Rd"] * p["uref"]
    c = 0.5 * (p["xref"] ** 2 * p["Qd"]).sum(-1); c[:, :N16] += 0.5 * (p["uref"] ** 2 * p["Rd"]).sum(-1)
    d = altro_amd.Batch(N16, 12, 4, batch)
    d.set_dynamics(p["A"], p["B"], p["f"]); d.set_quadratic_cost(Q, R, np.zeros((batch, N16, 48)), q, r, c)
    d.set_initial_state(p["x0"]); d.set_input_guess(p["u0"])
    rd = d.ilqr_solve(iterations_max=10)
    assert np.array_equal(ra["iterations"], rd["iterations"])
    np.testing.assert_allclose(d.get_nominal()[0], a.get_nominal()[0], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(d.get_nominal()[1], a.get_nominal()[1], rtol=1e-10, atol=1e-10)


n12, m4 = 12, 4


def test_full_size_c1_dense_cost_properties():
    """BASELINE.json configs[1] at full size (N = 256, n = 12, m = 4, 4096 problems) with the general quadratic cost in the device loop:
    size-independent checks.  (1) an LQ problem with a dense cost converges in one sweep with alpha = 1 for every problem, and its
    KKT residual |lx + A^T y+ - y|, |lu + B^T y+| with lx = Q x + H^T u + q, lu = R u + H x + r (knotpoint_data.cpp:659-668), recomputed
    in numpy from the returned trajectory and duals, vanishes; (2) a seeded sample of problems against the oracle; (3) with diagonal
    Q, R and H = 0 the solve is the tracking cost's, bit for bit, for all 4096 problems."""
    batch, Nf = 4096, 256
    one = problems.c1_double_integrator(1, N=Nf)
    x0 = 2.0 * problems.uniform01((batch, n12), 21) - 1.0
    c = problems.quadratic_cost(1, Nf, n12, m4, stream=311)

    def handle():
        bt = altro_amd.Batch(Nf, n12, m4, batch)
        assert bt.plan == altro_amd.PLAN_MFMA16
        bt.set_dynamics(one["A"][0, :1], one["B"][0, :1], None, k_stride_zero=True, batch_stride_zero=True)
        bt.set_initial_state(x0)
        bt.set_input_guess(np.zeros((1, 1, m4)), k_stride_zero=True, batch_stride_zero=True)
        return bt

    bt = handle()
    bt.set_quadratic_cost(c["Q"], c["R"], c["H"], c["q"], c["r"], c["c"], batch_stride_zero=True)
    res = bt.ilqr_solve(iterations_max=10)
    assert (res["status"] == 0).all() and (res["iterations"] == 1).all() and (res["alpha"] == 1.0).all()
    x, u = bt.get_nominal()
    y = bt.get("y")
    A = one["A"][0, 0].reshape(n12, n12).T; B = one["B"][0, 0].reshape(m4, n12).T
    Q = c["Q"][0].reshape(Nf + 1, n12, n12).transpose(0, 2, 1); R = c["R"][0].reshape(Nf, m4, m4).transpose(0, 2, 1)
    H = c["H"][0].reshape(Nf, n12, m4).transpose(0, 2, 1)                      # [k][m][n]
    lx = np.einsum("kij,bkj->bki", Q, x) + c["q"][0][None]
    lx[:, :Nf] += np.einsum("kij,bki->bkj", H, u)
    lu = np.einsum("kij,bkj->bki", R, u) + np.einsum("kij,bkj->bki", H, x[:, :Nf]) + c["r"][0][None]
    res_x = lx[:, :Nf] + y[:, 1:] @ A - y[:, :Nf]
    res_u = lu + y[:, 1:] @ B
    res_N = lx[:, Nf] - y[:, Nf]
    scale = max(1.0, float(np.abs(y).max()))
    assert max(np.abs(res_x).max(), np.abs(res_u).max(), np.abs(res_N).max()) / scale < 1e-9
    for b in (0, 2222, 4095):
        s = oracle.ILQR(Nf, n12, m4, 0.01, oracle.DYN_LINEAR, cost_kind=oracle.COST_QUADRATIC)
        s.L.oracle_ilqr_set_linear_dynamics(s.h, np.ascontiguousarray(np.tile(one["A"][0, :1], (Nf, 1))), np.ascontiguousarray(np.tile(one["B"][0, :1], (Nf, 1))), None)
        for k in range(Nf + 1):
            kk = min(k, Nf - 1)
            s.L.oracle_ilqr_set_quadratic_cost(s.h, k, np.ascontiguousarray(c["Q"][0, k]), np.ascontiguousarray(c["R"][0, kk]).ctypes.data,
                                               np.ascontiguousarray(c["H"][0, kk]).ctypes.data, np.ascontiguousarray(c["q"][0, k]),
                                               np.ascontiguousarray(c["r"][0, kk]).ctypes.data, float(c["c"][0, k]))
        s.L.oracle_ilqr_set_initial_state(s.h, np.ascontiguousarray(x0[b]))
        s.L.oracle_ilqr_initialize(s.h)
        for k in range(Nf):
            s.L.oracle_ilqr_set_input(s.h, k, np.zeros(m4))
        s.L.oracle_ilqr_set_options(s.h, 10, 1e-4, 1e-4, 1e-8, 0)
        status, iters, log = s.solve()
        assert status == 0 and iters == 1
        np.testing.assert_allclose(x[b], s.get("x"), rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(u[b], s.get("u"), rtol=1e-8, atol=1e-8)
    bt.close()
    # (3) diagonal blocks: the tracking cost's solve
    Qd = np.stack([np.ones(n12), 100.0 * np.ones(n12)]); Rd = np.full((1, m4), 1e-2)
    bt1 = handle()
    bt1.set_tracking_cost(Qd, Rd, np.zeros((2, n12)), np.zeros((1, m4)), k_stride_zero=True, batch_stride_zero=True)
    r1 = bt1.ilqr_solve(iterations_max=10); x1, u1 = bt1.get_nominal(); bt1.close()
    bt2 = handle()
    bt2.set_quadratic_cost(np.stack([np.diag(Qd[0]).reshape(-1), np.diag(Qd[1]).reshape(-1)]), np.diag(Rd[0]).reshape(1, -1), np.zeros((1, m4 * n12)),
                           np.zeros((2, n12)), np.zeros((1, m4)), np.zeros(2), k_stride_zero=True, batch_stride_zero=True)
    r2 = bt2.ilqr_solve(iterations_max=10); x2, u2 = bt2.get_nominal(); bt2.close()
    assert np.array_equal(r1["iterations"], r2["iterations"]) and np.array_equal(x1, x2) and np.array_equal(u1, u2)
