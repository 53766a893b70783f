"""The iLQR loop on plan GENERIC (missing #3): shapes beyond the (12, 4) tile -- any n, m up to 64 -- with dynamics
given as data and a quadratic cost (tracking or dense), through kernels/ilqr_generic.hip, against the oracle's restatement of
SolverImpl with ORACLE_DYN_LINEAR.  Correctness-first plan: one wave per problem, sums reduced over the wave -- results agree to
rounding (merit phi 1e-11, phi' 1e-9, candidates 1e-10; whole LQ solves: same status / iterations, trajectories 1e-9)."""
import numpy as np
import pytest

import altro_amd
from oracle import oracle
from tests import problems

pytestmark = pytest.mark.gpu


def make(batch, N, n, m, dense, with_f=True):
    p = problems.ilqr12x4_problem(batch, N, with_f, n=n, m=m)      # random LTV dynamics + tracking cost of any shape
    if dense:
        p.update(problems.quadratic_cost(batch, N, n, m))
    bt = altro_amd.Batch(N, n, m, batch, plan=altro_amd.PLAN_GENERIC)   # (the bit-for-bit plan by name: AUTO gives these shapes plan MFMA32, tests/test_gpu_tile32.py)
    assert bt.plan == altro_amd.PLAN_GENERIC
    bt.set_dynamics(p["A"], p["B"], p["f"])
    if dense:
        bt.set_quadratic_cost(p["Q"], p["R"], p["H"], p["q"], p["r"], p["c"])
    else:
        bt.set_tracking_cost(p["Qd"], p["Rd"], p["xref"], p["uref"])
    bt.set_initial_state(p["x0"])
    bt.set_input_guess(p["u0"])
    return p, bt


def make_oracle(p, b, N, n, m, dense):
    s = oracle.ILQR(N, n, m, 0.01, oracle.DYN_LINEAR, cost_kind=oracle.COST_QUADRATIC if dense else oracle.COST_DIAGONAL)
    f = p["f"][b] if p["f"] is not None else None
    s.L.oracle_ilqr_set_linear_dynamics(s.h, np.ascontiguousarray(p["A"][b]), np.ascontiguousarray(p["B"][b]),
                                        None if f is None else np.ascontiguousarray(f).ctypes.data)
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
    s.L.oracle_ilqr_initialize(s.h)
    for k in range(N):
        s.L.oracle_ilqr_set_input(s.h, k, np.ascontiguousarray(p["u0"][b, k]))
    return s


@pytest.mark.parametrize("n,m,dense", [(16, 5, False), (16, 5, True), (13, 4, True), (32, 8, False), (12, 6, True), (48, 12, True),
                                       (10, 33, True)])
def test_merit_expansion_stationarity_generic(n, m, dense):
    N, batch = 14, 7
    p, bt = make(batch, N, n, m, dense)
    bt.open_loop_rollout(); bt.accept(); bt.expand(); bt.backward()
    assert (bt.get("status") == -1).all()
    A0, B0, lx0, lu0 = bt.get_expansion()
    alphas = np.linspace(0.0, 1.2, batch)
    phi, dphi = bt.merit(alphas)
    xc, uc, yc = bt.get("x"), bt.get("u"), bt.get("y")
    st = bt.stationarity()
    for b in [0, 3, 6]:
        s = make_oracle(p, b, N, n, m, dense)
        s.L.oracle_ilqr_open_loop_rollout(s.h); s.L.oracle_ilqr_copy_trajectory(s.h)
        s.L.oracle_ilqr_calc_dynamics_expansions(s.h); s.L.oracle_ilqr_calc_cost_gradient(s.h)
        s.L.oracle_ilqr_calc_expansions(s.h)
        np.testing.assert_allclose(lx0[b], s.get("lx"), rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(lu0[b], s.get("lu"), rtol=1e-12, atol=1e-12)
        assert s.L.oracle_ilqr_backward_pass(s.h) == -1
        assert np.array_equal(bt.get("K")[b], s.get("K"))           # plan GENERIC's sweep is the oracle's bit for bit
        p_ref, dp_ref = s.merit(alphas[b])
        assert abs(phi[b] - p_ref) <= 1e-11 * max(1.0, abs(p_ref)), (b, phi[b], p_ref)
        assert abs(dphi[b] - dp_ref) <= 1e-9 * max(1.0, abs(dp_ref)), (b, dphi[b], dp_ref)
        np.testing.assert_allclose(xc[b], s.get("x_cand"), rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(uc[b], s.get("u_cand"), rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(yc[b], s.get("y_cand"), rtol=1e-9, atol=1e-9)
        ref_st = s.L.oracle_ilqr_stationarity(s.h)
        assert abs(st[b] - ref_st) <= 1e-8 * max(1.0, ref_st)


@pytest.mark.parametrize("n,m,dense,dtype", [(16, 5, False, altro_amd.F64), (20, 7, True, altro_amd.F64), (32, 8, True, altro_amd.F64),
                                              (16, 5, True, altro_amd.F32), (40, 10, True, altro_amd.F64), (64, 16, False, altro_amd.F64),
                                              (20, 40, True, altro_amd.F64)])   # (past 32 a lane owns a state row AND an input row)
def test_whole_lq_solves_generic(n, m, dense, dtype):
    """Whole solves of an LQ problem (alpha = 1, <= 3 sweeps) on plan GENERIC; MPC operations on the resident batch afterwards
    (UpdateLinearCosts, SetInitialState, ShiftTrajectory: bicycle_test.cpp:302-337's pattern) and a second solve."""
    N, batch = 18, 11
    p = problems.ilqr12x4_problem(batch, N, True, n=n, m=m)
    if dense:
        p.update(problems.quadratic_cost(batch, N, n, m))
    bt = altro_amd.Batch(N, n, m, batch, dtype=dtype, plan=altro_amd.PLAN_GENERIC)
    assert bt.plan == altro_amd.PLAN_GENERIC
    bt.set_dynamics(p["A"], p["B"], p["f"])
    if dense:
        bt.set_quadratic_cost(p["Q"], p["R"], p["H"], p["q"], p["r"], p["c"])
    else:
        bt.set_tracking_cost(p["Qd"], p["Rd"], p["xref"], p["uref"])
    bt.set_initial_state(p["x0"]); bt.set_input_guess(p["u0"])
    f32 = dtype == altro_amd.F32
    res = bt.ilqr_solve(iterations_max=10, tol_stationarity=1e-2 if f32 else 1e-4)
    assert (res["status"] == 0).all() and (res["iterations"] <= 3).all()
    x, u = bt.get_nominal()
    tol = 2e-4 if f32 else 1e-9
    ors = {}
    for b in [0, 5, 10]:
        s = make_oracle(p, b, N, n, m, dense)
        s.L.oracle_ilqr_set_options(s.h, 10, 1e-4, 1e-4, 1e-8, 0)
        status, iters, log = s.solve()
        assert status == 0 and (f32 or iters == res["iterations"][b])
        np.testing.assert_allclose(x[b], s.get("x"), rtol=tol, atol=tol)
        np.testing.assert_allclose(u[b], s.get("u"), rtol=tol * 10, atol=tol * 10)
        ors[b] = s
    if f32:
        return
    # receding horizon: move the linear cost terms, take x_1 as the new initial state, shift, solve again
    qbase = p["q"] if dense else -(p["Qd"] * p["xref"])
    qnew = qbase + 0.05
    x1, _ = bt.get_knot(1)
    bt.update_linear_costs(qnew, None, np.zeros((batch, N + 1)), 0, N)
    bt.set_initial_state(x1)
    bt.shift_trajectory()
    res2 = bt.ilqr_solve(iterations_max=10)
    x2, u2 = bt.get_nominal()
    for b, s in ors.items():
        for k in range(N + 1):
            s.L.oracle_ilqr_update_linear_costs(s.h, k, np.ascontiguousarray(qnew[b, k]).ctypes.data, None, 0.0)
        s.L.oracle_ilqr_set_initial_state(s.h, np.ascontiguousarray(x1[b]))
        s.L.oracle_ilqr_shift_trajectory(s.h)
        status, iters, log = s.solve()
        assert res2["status"][b] == status and res2["iterations"][b] == iters
        np.testing.assert_allclose(x2[b], s.get("x"), rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(u2[b], s.get("u"), rtol=1e-8, atol=1e-8)


def _blocks(N, n, m, soc):
    """|u_i| <= 0.3 for the first four inputs (two INEQUALITY blocks share a knot point with nothing else; a third kind rides on other
    knot points): x1 <= 1.2 and -x2 <= 1.2 at 1 <= k < N ... as far as two blocks per knot point allow: bounds everywhere, the state
    half-spaces where no equality block sits, u_0[0] == 0.05 at k = 0, a second-order cone on (u_0, u_1, u_2 ; 0.35) instead of the
    half-spaces when `soc`, and a terminal half-space block."""
    w = n + m
    Gb = np.zeros((8, w)); Gb[:4, n:n + 4] = np.eye(4); Gb[4:, n:n + 4] = -np.eye(4)
    Gs = np.zeros((2, w)); Gs[0, 1] = 1.0; Gs[1, n - 1] = -1.0
    Ge = np.zeros((1, w)); Ge[0, n] = 1.0
    Gt = np.zeros((3, w)); Gt[0, 0] = 1.0; Gt[1, 1] = -1.0; Gt[2, n - 2] = 1.0
    out = [(0, N - 1, altro_amd.CONE_INEQUALITY, Gb, np.full(8, 0.3)), (0, 0, altro_amd.CONE_EQUALITY, Ge, np.array([0.05])),
           (N, N, altro_amd.CONE_INEQUALITY, Gt, np.array([0.4, 0.4, 0.2]))]
    if soc:
        Gc = np.zeros((4, w)); Gc[0, n + 1] = 1.0; Gc[1, n + 2] = 1.0; Gc[2, n + 3] = 1.0
        out.append((1, N - 1, altro_amd.CONE_SOC, Gc, np.array([0.0, 0.0, 0.0, -0.35])))
    else:
        out.append((1, N - 1, altro_amd.CONE_INEQUALITY, Gs, np.array([1.2, 1.2])))
    return out


@pytest.mark.parametrize("n,m,dense,soc,backtracking", [(14, 5, False, False, False), (16, 6, True, False, True), (13, 5, False, True, False),
                                                         (20, 8, True, True, False)])
def test_constrained_solves_generic(n, m, dense, soc, backtracking):
    """Constraint blocks on plan GENERIC (knotpoint_data.cpp:489-613, solver.cpp:383-409, 470-489 for shapes beyond the tile):
    inequality / equality blocks, a second-order cone, a terminal block, two blocks per knot point -- whole AL-iLQR solves against
    the oracle per problem: same status, iterations and dual updates, feasibility, trajectories 1e-7; the duals themselves 1e-6."""
    N, batch = 16, 9
    p, bt = make(batch, N, n, m, dense)
    blocks = _blocks(N, n, m, soc)
    for (k0, k1, cone, G, g) in blocks:
        bt.add_linear_constraint(k0, k1, cone, G, g)
    res = bt.ilqr_solve(iterations_max=60, penalty_initial=1.0, penalty_scaling=10.0, use_backtracking=backtracking)
    x, u = bt.get_nominal()
    assert (res["dual_updates"] > 0).all()
    nconv = 0
    for b in [0, 4, 8]:
        s = make_oracle(p, b, N, n, m, dense)
        for (k0, k1, cone, G, g) in blocks:
            for k in range(k0, k1 + 1):
                s.add_linear_constraint(k, cone, G, g)
        s.L.oracle_ilqr_initialize(s.h)
        for k in range(N):
            s.L.oracle_ilqr_set_input(s.h, k, np.ascontiguousarray(p["u0"][b, k]))
        s.set_penalty(1.0, 10.0)
        s.L.oracle_ilqr_set_options(s.h, 60, 1e-4, 1e-4, 1e-8, 1 if backtracking else 0)
        status, iters, log = s.solve()
        assert res["status"][b] == status and res["iterations"][b] == iters, (b, res["status"][b], status, res["iterations"][b], iters)
        if status != 0:
            continue
        nconv += 1
        assert abs(res["feasibility"][b] - log[iters - 1, 6]) <= 1e-9 + 1e-3 * log[iters - 1, 6]
        np.testing.assert_allclose(x[b], s.get("x"), rtol=1e-7, atol=1e-7)
        np.testing.assert_allclose(u[b], s.get("u"), rtol=1e-6, atol=1e-6)
        assert np.abs(u[b][:, :4]).max() <= 0.3 + 2e-4 and abs(u[b][0, 0] - 0.05) < 2e-4
    assert nconv >= 1
    assert (bt.feasibility() >= 0.0).all()
    z = bt.get_duals(0, 0, 8)
    assert z.shape == (batch, 8) and (z <= 1e-12).all()      # duals of an INEQUALITY block live in the non-positive orthant
    bt.close()


def _input_cone(N, n, m):
    """||u|| <= 0.4 over ALL inputs: (u; 0.4) in the cone, p = m + 1 rows."""
    Gu = np.zeros((m + 1, n + m)); Gu[np.arange(m), n + np.arange(m)] = 1.0
    return (0, N - 1, altro_amd.CONE_SOC, Gu, np.concatenate([np.zeros(m), [-0.4]]))


def _fresh(p, N, n, m, batch, dense, plan, blocks):
    bt = altro_amd.Batch(N, n, m, batch, plan=plan)
    bt.set_dynamics(p["A"], p["B"], p["f"])
    if dense:
        bt.set_quadratic_cost(p["Q"], p["R"], p["H"], p["q"], p["r"], p["c"])
    else:
        bt.set_tracking_cost(p["Qd"], p["Rd"], p["xref"], p["uref"])
    bt.set_initial_state(p["x0"]); bt.set_input_guess(p["u0"])
    for (k0, k1, cone, G, g) in blocks:
        bt.add_linear_constraint(k0, k1, cone, G, g)
    return bt


@pytest.mark.parametrize("n,m,dense,which,plan", [(14, 5, False, "inputs", altro_amd.PLAN_GENERIC), (16, 6, True, "terminal", altro_amd.PLAN_GENERIC),
                                                   (13, 4, False, "terminal", altro_amd.PLAN_AUTO), (20, 8, False, "dense", altro_amd.PLAN_GENERIC)])
def test_second_order_cones_of_many_rows(n, m, dense, which, plan):
    """VERDICT r5 item 4's second half: cones of 5 .. 12 rows on plans GENERIC / MFMA32 (one lane per row, the projection's Jacobian and
    curvature applied from their closed forms: gen_al_rows, generic_expand_al_kernel; cones.cpp:13-123 takes any dimension) -- whole
    AL-iLQR solves against the oracle, EVERY problem of the batch: status, iterations, trajectories; the input bound holds and binds;
    the cone's duals lie inside the (self-dual) cone.  `terminal`: + a cone over the first eight states at k = N (p = 9); `dense`: + a
    cone of twelve rows with a dense G and offsets at 1 <= k < N (two cones at a knot point) -- both bounds placed to bind on the
    solution that has the input cone alone."""
    N, batch = 14, 7
    w = n + m
    p, bt0 = make(batch, N, n, m, dense)
    bt0.close()
    blocks = [_input_cone(N, n, m)]
    bt0 = _fresh(p, N, n, m, batch, dense, plan, blocks)
    r0 = bt0.ilqr_solve(iterations_max=60, penalty_initial=1.0, penalty_scaling=10.0)
    x0s, u0s = bt0.get_nominal()
    bt0.close()
    assert (r0["status"] == 0).sum() >= batch - 1
    if which == "terminal":
        Gx = np.zeros((9, w)); Gx[np.arange(8), np.arange(8)] = 1.0
        blocks.append((N, N, altro_amd.CONE_SOC, Gx, np.concatenate([np.zeros(8), [-0.85 * float(np.linalg.norm(x0s[:, N, :8], axis=1).max())]])))
    if which == "dense":
        rng = np.random.default_rng(7 * n + m)
        Gd = np.zeros((12, w)); Gd[:11] = rng.normal(size=(11, w)) * (rng.random((11, w)) < 0.4)
        gd = np.concatenate([0.05 * rng.normal(size=11), [0.0]])
        v = np.einsum("rw,bkw->bkr", Gd[:11], np.concatenate([x0s[:, 1:N], u0s[:, 1:N]], axis=2)) - gd[:11]
        gd[11] = -0.85 * float(np.linalg.norm(v, axis=2).max())
        blocks.append((1, N - 1, altro_amd.CONE_SOC, Gd, gd))
    assert max(G.shape[0] for (_, _, _, G, _) in blocks) > 4
    bt = _fresh(p, N, n, m, batch, dense, plan, blocks)
    assert bt.plan == (altro_amd.PLAN_MFMA32 if plan == altro_amd.PLAN_AUTO else plan)
    res = bt.ilqr_solve(iterations_max=60, penalty_initial=1.0, penalty_scaling=10.0)
    x, u = bt.get_nominal()
    nconv = nbind = 0
    for b in range(batch):
        s = make_oracle(p, b, N, n, m, dense)
        for (k0, k1, cone, G, g) in blocks:
            for k in range(k0, k1 + 1):
                s.add_linear_constraint(k, cone, G, g)
        s.L.oracle_ilqr_initialize(s.h)
        for k in range(N):
            s.L.oracle_ilqr_set_input(s.h, k, np.ascontiguousarray(p["u0"][b, k]))
        s.set_penalty(1.0, 10.0)
        s.L.oracle_ilqr_set_options(s.h, 60, 1e-4, 1e-4, 1e-8, 0)
        status, iters, log = s.solve()
        assert res["status"][b] == status and res["iterations"][b] == iters, (b, res["status"][b], status, res["iterations"][b], iters)
        if status != 0:
            continue
        nconv += 1
        np.testing.assert_allclose(x[b], s.get("x"), rtol=1e-7, atol=1e-7)
        np.testing.assert_allclose(u[b], s.get("u"), rtol=1e-6, atol=1e-6)
        nrm = np.linalg.norm(u[b], axis=1)
        assert nrm.max() <= 0.4 + 5e-4
        nbind += int(nrm.max() >= 0.4 - 5e-4)
    assert nconv >= batch - 2
    assert nbind >= 1                                  # the cone binds somewhere: its duals moved
    pz = blocks[0][3].shape[0]
    z = bt.get_duals(0, 0, pz)
    assert z.shape == (batch, pz) and (np.linalg.norm(z[:, :-1], axis=1) <= z[:, -1] * (1 + 1e-12) + 1e-9).all()   # z in the cone
    assert np.abs(z).max() > 0.0
    if which != "inputs":                              # the second cone moved the solution: it is not the input cone's alone
        assert np.abs(x - x0s).max() > 1e-3
    assert (res["dual_updates"] > 0).all()
    bt.close()


def test_generic_plan_says_what_it_does_not_do():
    bt = altro_amd.Batch(10, 16, 5, 4, plan=altro_amd.PLAN_GENERIC)
    with pytest.raises(altro_amd.AltroHipError):
        bt.set_model(altro_amd.MODEL_BICYCLE, 0.1)
    # dimensions past 64: the TVLQR sweeps run (tests/test_gpu_parity.py::test_generic_random), the iLQR loop says why it does not
    N, n, m, batch = 5, 80, 10, 2
    p = problems.ilqr12x4_problem(batch, N, True, n=n, m=m)
    big = altro_amd.Batch(N, n, m, batch, plan=altro_amd.PLAN_GENERIC)
    assert big.plan == altro_amd.PLAN_GENERIC
    big.set_dynamics(p["A"], p["B"], p["f"])
    big.set_tracking_cost(p["Qd"], p["Rd"], p["xref"], p["uref"])
    big.set_initial_state(p["x0"]); big.set_input_guess(p["u0"])
    with pytest.raises(altro_amd.AltroHipError, match="n, m <= 64"):
        big.ilqr_solve(iterations_max=3)


@pytest.mark.parametrize("dense", [False, True])
def test_input_guess_before_the_cost_is_kept(dense):
    """the reference allows SetInput at any time (altro_solver.cpp:242-251); on plan GENERIC the first cost
    setter used to zero the candidate inputs, so a guess given BEFORE the cost was lost and the solve started from u = 0."""
    N, n, m, batch = 9, 14, 5, 4
    p = problems.ilqr12x4_problem(batch, N, True, n=n, m=m)
    if dense:
        p.update(problems.quadratic_cost(batch, N, n, m))

    def build(guess_first):
        bt = altro_amd.Batch(N, n, m, batch, plan=altro_amd.PLAN_GENERIC)
        assert bt.plan == altro_amd.PLAN_GENERIC
        bt.set_dynamics(p["A"], p["B"], p["f"])
        bt.set_initial_state(p["x0"])
        if guess_first:
            bt.set_input_guess(p["u0"])
        if dense:
            bt.set_quadratic_cost(p["Q"], p["R"], p["H"], p["q"], p["r"], p["c"])
        else:
            bt.set_tracking_cost(p["Qd"], p["Rd"], p["xref"], p["uref"])
        if not guess_first:
            bt.set_input_guess(p["u0"])
        return bt
    a, b = build(True), build(False)
    assert np.abs(p["u0"]).max() > 0
    a.open_loop_rollout(); b.open_loop_rollout()
    np.testing.assert_array_equal(a.get("u"), p["u0"])
    np.testing.assert_array_equal(a.get("x"), b.get("x"))
    ra, rb = a.ilqr_solve(iterations_max=10), b.ilqr_solve(iterations_max=10)
    assert np.array_equal(ra["iterations"], rb["iterations"])
    xa, ua = a.get_nominal(); xb, ub = b.get_nominal()
    np.testing.assert_array_equal(xa, xb); np.testing.assert_array_equal(ua, ub)


def test_whole_solves_with_the_matrix_core_products():
    """ALTRO_HIP_GENERIC_MATRIX_CORES through the iLQR loop (unconstrained LQ problems: one exact step): statuses and iterations of the
    bit-exact form, trajectories to 1e-10; a constrained problem converges to the same solution (1e-6), whatever path it takes."""
    N, n, m, batch = 18, 20, 7, 9
    p = problems.ilqr12x4_problem(batch, N, True, n=n, m=m)
    p.update(problems.quadratic_cost(batch, N, n, m))
    out = {}
    for name, flags in (("exact", 0), ("mc", altro_amd.GENERIC_MATRIX_CORES)):
        for constrained in (False, True):
            bt = altro_amd.Batch(N, n, m, batch, plan=altro_amd.PLAN_GENERIC, flags=flags)
            bt.set_dynamics(p["A"], p["B"], p["f"])
            bt.set_quadratic_cost(p["Q"], p["R"], p["H"], p["q"], p["r"], p["c"])
            bt.set_initial_state(p["x0"]); bt.set_input_guess(p["u0"])
            if constrained:
                G = np.zeros((2 * m, n + m)); G[:m, n:] = np.eye(m); G[m:, n:] = -np.eye(m)
                bt.add_linear_constraint(0, N - 1, altro_amd.CONE_INEQUALITY, G, np.full(2 * m, 0.3))
            res = bt.ilqr_solve(iterations_max=60, tol_stationarity=1e-5, penalty_initial=1.0, penalty_scaling=10.0)
            out[(name, constrained)] = (res, bt.get_nominal())
            bt.close()
    (re, (xe, ue)), (rm, (xm, um)) = out[("exact", False)], out[("mc", False)]
    assert (re["status"] == 0).all() and np.array_equal(re["status"], rm["status"]) and np.array_equal(re["iterations"], rm["iterations"])
    np.testing.assert_allclose(xm, xe, rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(um, ue, rtol=1e-9, atol=1e-9)
    (re, (xe, ue)), (rm, (xm, um)) = out[("exact", True)], out[("mc", True)]
    both = (re["status"] == 0) & (rm["status"] == 0)             # (one of the nine runs out of iterations in either form)
    assert both.sum() >= batch - 2 and (re["status"] == 0).sum() == (rm["status"] == 0).sum()
    np.testing.assert_allclose(xm[both], xe[both], rtol=0, atol=2e-4)   # (both within the solver's tolerances of one solution)
    np.testing.assert_allclose(um[both], ue[both], rtol=0, atol=2e-3)
    assert np.abs(um[both]).max() <= 0.3 + 2e-4
