"""The iLQR loop on plan MFMA16 ((n, m) = (12, 4), wave per problem) for dynamics given as data
(x+ = A_k x + B_k u + f_k, the reference's SetLinearDynamics path) with the diagonal tracking cost, against the
oracle's restatement of SolverImpl with ORACLE_DYN_LINEAR.  An LQ problem: the reference converges in one or two
sweeps with alpha = 1 (solver_impl_test.cpp:309-315, double_integrator_test.cpp:129-132)."""
import numpy as np
import pytest

import altro_amd
from oracle import oracle
from tests import problems

pytestmark = pytest.mark.gpu

N, n, m = 24, 12, 4


def make_problem(batch, with_f):
    return problems.ilqr12x4_problem(batch, N, with_f)


def make_hip(p):
    batch = p["x0"].shape[0]
    bt = altro_amd.Batch(N, n, m, batch)
    assert bt.plan == altro_amd.PLAN_MFMA16
    bt.set_dynamics(p["A"], p["B"], p["f"])
    bt.set_tracking_cost(p["Qd"], p["Rd"], p["xref"], p["uref"])
    bt.set_initial_state(p["x0"])
    bt.set_input_guess(p["u0"])
    return bt


def make_oracle(p, b):
    s = oracle.ILQR(N, n, m, 0.01, oracle.DYN_LINEAR, cost_kind=oracle.COST_DIAGONAL)
    f = p["f"][b] if p["f"] is not None else None
    s.L.oracle_ilqr_set_linear_dynamics(s.h, np.ascontiguousarray(p["A"][b]), np.ascontiguousarray(p["B"][b]),
                                        None if f is None else np.ascontiguousarray(f).ctypes.data)
    for k in range(N + 1):
        s.L.oracle_ilqr_set_lqr_cost(s.h, k, np.ascontiguousarray(p["Qd"][b, k]), np.ascontiguousarray(p["Rd"][b, min(k, N - 1)]),
                                     np.ascontiguousarray(p["xref"][b, k]), np.ascontiguousarray(p["uref"][b, min(k, N - 1)]))
    s.L.oracle_ilqr_set_initial_state(s.h, np.ascontiguousarray(p["x0"][b]))
    s.L.oracle_ilqr_initialize(s.h)
    for k in range(N):
        s.L.oracle_ilqr_set_input(s.h, k, np.ascontiguousarray(p["u0"][b, k]))
    return s


@pytest.mark.parametrize("with_f", [False, True])
def test_merit_function_parity_mfma16(with_f):
    """solver.cpp:273-355 on the device for (12, 4): phi, dphi, candidate x_/u_/y_, refreshed lx, lu."""
    batch = 9
    p = make_problem(batch, with_f)
    bt = make_hip(p)
    bt.open_loop_rollout(); bt.accept(); bt.expand(); bt.backward()
    assert (bt.get("status") == -1).all()
    alphas = np.linspace(0.0, 1.2, batch)
    phi, dphi = bt.merit(alphas)
    xc, uc, yc = bt.get("x"), bt.get("u"), bt.get("y")
    st = bt.stationarity()
    for b in [0, 4, 8]:
        s = make_oracle(p, b)
        s.L.oracle_ilqr_open_loop_rollout(s.h); s.L.oracle_ilqr_copy_trajectory(s.h)
        s.L.oracle_ilqr_calc_dynamics_expansions(s.h); s.L.oracle_ilqr_calc_cost_gradient(s.h)
        s.L.oracle_ilqr_calc_expansions(s.h)
        assert s.L.oracle_ilqr_backward_pass(s.h) == -1
        np.testing.assert_allclose(bt.get("K")[b], s.get("K"), rtol=1e-9, atol=1e-9)
        p_ref, dp_ref = s.merit(alphas[b])
        assert abs(phi[b] - p_ref) <= 1e-11 * max(1.0, abs(p_ref))
        assert abs(dphi[b] - dp_ref) <= 1e-9 * max(1.0, abs(dp_ref))
        np.testing.assert_allclose(xc[b], s.get("x_cand"), rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(uc[b], s.get("u_cand"), rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(yc[b], s.get("y_cand"), rtol=1e-9, atol=1e-9)
        assert abs(st[b] - s.L.oracle_ilqr_stationarity(s.h)) <= 1e-8 * max(1.0, st[b])


@pytest.mark.parametrize("with_f", [False, True])
def test_batched_lq_solve_mfma16(with_f):
    """Whole solves: same status / iterations as the oracle per problem (an LQ problem: alpha = 1, <= 3 sweeps),
    same trajectories."""
    batch = 70
    p = make_problem(batch, with_f)
    bt = make_hip(p)
    res = bt.ilqr_solve(iterations_max=10)
    assert (res["status"] == 0).all()
    assert (res["iterations"] <= 3).all()
    assert (np.abs(res["alpha"] - 1.0) < 1e-12).all()
    x, u = bt.get_nominal()
    for b in [0, 33, 69]:
        s = make_oracle(p, b)
        s.L.oracle_ilqr_set_options(s.h, 10, 1e-4, 1e-4, 1e-8, 0)
        status, iters, log = s.solve()
        assert status == 0 and iters == res["iterations"][b]
        np.testing.assert_allclose(x[b], s.get("x"), rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(u[b], s.get("u"), rtol=1e-8, atol=1e-8)


def test_c1_double_integrator_ilqr():
    """BASELINE.json configs[1] as an iLQR solve: the C1 double integrator (n = 12, m = 4) converges in <= 3
    sweeps (double_integrator_test.cpp:129-132) and beats the open-loop cost."""
    batch = 130
    one = problems.c1_double_integrator(1, N=64)
    bt = altro_amd.Batch(64, 12, 4, batch)
    bt.set_dynamics(one["A"][0, :1], one["B"][0, :1], None, k_stride_zero=True, batch_stride_zero=True)
    Qd = np.stack([np.ones(12), 100.0 * np.ones(12)])
    bt.set_tracking_cost(Qd, np.full((1, 4), 1e-2), np.zeros((2, 12)), np.zeros((1, 4)), k_stride_zero=True, batch_stride_zero=True)
    x0 = 2.0 * problems.uniform01((batch, 12), 21) - 1.0
    bt.set_initial_state(x0)
    bt.set_input_guess(np.zeros((1, 1, 4)), k_stride_zero=True, batch_stride_zero=True)
    res = bt.ilqr_solve(iterations_max=10)
    assert (res["status"] == 0).all() and (res["iterations"] <= 3).all()
    x, u = bt.get_nominal()
    for b in [0, 129]:
        s = oracle.ILQR(64, 12, 4, 0.01, oracle.DYN_LINEAR, cost_kind=oracle.COST_DIAGONAL)
        s.L.oracle_ilqr_set_linear_dynamics(s.h, np.ascontiguousarray(np.tile(one["A"][0, :1], (64, 1))),
                                            np.ascontiguousarray(np.tile(one["B"][0, :1], (64, 1))), None)
        for k in range(65):
            s.L.oracle_ilqr_set_lqr_cost(s.h, k, np.ascontiguousarray(Qd[1 if k == 64 else 0]), np.full(4, 1e-2), np.zeros(12), np.zeros(4))
        s.L.oracle_ilqr_set_initial_state(s.h, np.ascontiguousarray(x0[b]))
        s.L.oracle_ilqr_initialize(s.h)
        s.L.oracle_ilqr_set_options(s.h, 10, 1e-4, 1e-4, 1e-8, 0)
        status, iters, log = s.solve()
        assert status == 0 and iters == res["iterations"][b]
        np.testing.assert_allclose(x[b], s.get("x"), rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(u[b], s.get("u"), rtol=1e-8, atol=1e-8)


def test_full_size_c1_ilqr_solve():
    """BASELINE.json configs[1] at full size (N = 256, batch = 4096) as an iLQR solve: every problem converges in
    one sweep with alpha = 1 and a stationarity at rounding level.  (Regression: a null-stream memset of the
    freshly allocated cost-parameter records raced the upload kernels on the handle's non-blocking stream and
    zeroed the first problems' parameters.)"""
    Nf, batch = 256, 4096
    one = problems.c1_double_integrator(1, N=Nf)
    bt = altro_amd.Batch(Nf, 12, 4, batch)
    bt.set_dynamics(one["A"][0, :1], one["B"][0, :1], None, k_stride_zero=True, batch_stride_zero=True)
    Qd = np.stack([np.ones(12), 100.0 * np.ones(12)])
    bt.set_tracking_cost(Qd, np.full((1, 4), 1e-2), np.zeros((2, 12)), np.zeros((1, 4)), k_stride_zero=True, batch_stride_zero=True)
    bt.set_initial_state(2.0 * problems.uniform01((batch, 12), 21) - 1.0)
    bt.set_input_guess(np.zeros((1, 1, 4)), k_stride_zero=True, batch_stride_zero=True)
    res = bt.ilqr_solve(iterations_max=10)
    assert (res["status"] == 0).all()
    assert (res["iterations"] == 1).all()
    assert (res["alpha"] == 1.0).all()
    assert res["stationarity"].max() < 1e-10
    assert res["sweeps"] == 1


def _constrained_problem(batch):
    return make_problem(batch, True), problems.ilqr12x4_constraint_blocks(N)


def test_constrained_lq_solve_mfma16():
    """Linear MPC-style problem at (12, 4): input bounds + state half-spaces (INEQUALITY) and an EQUALITY block on
    the first input, through the AL loop on plan MFMA16, against the oracle per problem: same status / iterations / number of
    dual updates / feasibility, same trajectory."""
    batch = 40
    p, blocks = _constrained_problem(batch)
    bt = make_hip(p)
    for (k0, k1, cone, G, g) in blocks:
        bt.add_linear_constraint(k0, k1, cone, G, g)
    res = bt.ilqr_solve(iterations_max=60, penalty_initial=1.0, penalty_scaling=10.0)
    x, u = bt.get_nominal()
    nconv = 0
    for b in [0, 17, 39]:
        s = make_oracle(p, b)
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
        assert abs(res["feasibility"][b] - log[iters - 1, 6]) <= 1e-9 + 1e-3 * log[iters - 1, 6]
        np.testing.assert_allclose(x[b], s.get("x"), rtol=1e-7, atol=1e-7)
        np.testing.assert_allclose(u[b], s.get("u"), rtol=1e-6, atol=1e-6)
        assert np.abs(u[b]).max() <= 0.3 + 2e-4 and abs(u[b][0, 0] - 0.05) < 2e-4
    assert nconv >= 2
    assert (res["dual_updates"] > 0).all()


def test_receding_horizon_ops_mfma16():
    """The MPC caller pattern of test/bicycle_test.cpp:302-337 on the (12, 4) plan: solve, take u_0, move the linear
    cost terms (UpdateLinearCosts), SetInitialState, ShiftTrajectory -- with input bounds -- against the oracle
    running each problem alone."""
    batch, nsim = 6, 3
    p, blocks = _constrained_problem(batch)
    blocks = blocks[:1]                       # input bounds only
    bt = make_hip(p)
    for (k0, k1, cone, G, g) in blocks:
        bt.add_linear_constraint(k0, k1, cone, G, g)
    ors = []
    for b in range(batch):
        s = make_oracle(p, b)
        for (k0, k1, cone, G, g) in blocks:
            for k in range(k0, k1 + 1):
                s.add_linear_constraint(k, cone, G, g)
        s.L.oracle_ilqr_initialize(s.h)
        for k in range(N):
            s.L.oracle_ilqr_set_input(s.h, k, np.ascontiguousarray(p["u0"][b, k]))
        s.set_penalty(1.0, 10.0)
        s.L.oracle_ilqr_set_options(s.h, 60, 1e-4, 1e-4, 1e-8, 0)
        ors.append(s)
    xs = p["x0"].copy()
    for it in range(nsim):
        res = bt.ilqr_solve(iterations_max=60, penalty_initial=1.0, penalty_scaling=10.0)
        xk, u0 = bt.get_knot(0)
        x1, _ = bt.get_knot(1)
        qnew = -(p["Qd"] * (p["xref"] + 0.05 * (it + 1)))          # a moving reference: q = -Qd xref'
        cnew = 0.1 * (it + 1) * np.ones((batch, N + 1))
        for b in range(batch):
            s = ors[b]
            status, iters, log = s.solve()
            assert res["status"][b] == status and res["iterations"][b] == iters, (it, b)
            np.testing.assert_allclose(u0[b], s.get("u")[0], rtol=1e-6, atol=1e-6)
            np.testing.assert_allclose(x1[b], s.get("x")[1], rtol=1e-7, atol=1e-7)
            for k in range(N + 1):
                s.L.oracle_ilqr_update_linear_costs(s.h, k, np.ascontiguousarray(qnew[b, k]).ctypes.data, None, float(cnew[b, k]))
            s.L.oracle_ilqr_set_initial_state(s.h, np.ascontiguousarray(x1[b]))
            s.L.oracle_ilqr_shift_trajectory(s.h)
        bt.update_linear_costs(qnew, None, cnew, 0, N)
        bt.set_initial_state(x1)
        bt.shift_trajectory()


def test_second_order_cone_block_mfma16():
    """A second-order-cone block on the (12, 4) plan: || (u0, u1, u2) || <= 0.35 at every k < N (the form of
    double_integrator_test.cpp:414-448), against the oracle per problem."""
    batch = 12
    p = make_problem(batch, False)
    w = n + m
    Gs = np.zeros((4, w)); Gs[0, 12] = 1.0; Gs[1, 13] = 1.0; Gs[2, 14] = 1.0
    gs = np.array([0.0, 0.0, 0.0, -0.35])
    bt = make_hip(p)
    bt.add_linear_constraint(0, N - 1, altro_amd.CONE_SOC, Gs, gs)
    res = bt.ilqr_solve(iterations_max=80, penalty_initial=1.0, penalty_scaling=10.0)
    x, u = bt.get_nominal()
    nconv = 0
    for b in [0, 5, 11]:
        s = make_oracle(p, b)
        for k in range(N):
            s.add_linear_constraint(k, oracle.CONE_SOC, Gs, gs)
        s.L.oracle_ilqr_initialize(s.h)
        for k in range(N):
            s.L.oracle_ilqr_set_input(s.h, k, np.ascontiguousarray(p["u0"][b, k]))
        s.set_penalty(1.0, 10.0)
        s.L.oracle_ilqr_set_options(s.h, 80, 1e-4, 1e-4, 1e-8, 0)
        status, iters, log = s.solve()
        assert res["status"][b] == status and res["iterations"][b] == iters, (b, res["status"][b], status, res["iterations"][b], iters)
        if status != 0:
            continue
        nconv += 1
        np.testing.assert_allclose(x[b], s.get("x"), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(u[b], s.get("u"), rtol=1e-5, atol=1e-5)
        assert np.linalg.norm(u[b][:, :3], axis=1).max() <= 0.35 + 2e-4
    assert nconv >= 2


def test_regularisation_retry_on_plan_mfma16():
    """SURVEY.md section 8 row f4 on the metric's own shape: the reference keeps reg = 0 and ignores
    a failed Cholesky (tvlqr.cpp:159-164, solver.cpp:363, :373-377) -- the default -- and with reg_retry_max > 0 the
    problems whose Quu is indefinite repeat the wave-per-problem backward sweep with a growing per-problem reg while
    the others keep reg = 0 and their bits."""
    batch = 40
    p = make_problem(batch, True)
    bad = np.arange(batch) % 5 == 2                      # every fifth problem: R = -0.3 I  ->  Quu indefinite
    p["Rd"] = p["Rd"].copy(); p["Rd"][bad] = -0.3
    bt = make_hip(p)
    res = bt.ilqr_solve(iterations_max=2)                # reference behaviour: failures ignored, no retry
    st = bt.get("status")
    assert (res["reg_retries"] == 0).all() and (st[bad] != -1).all() and (st[~bad] == -1).all()
    good_x = bt.get_nominal()[0][~bad].copy()
    bt2 = make_hip(p)
    res2 = bt2.ilqr_solve(iterations_max=2, reg_retry_max=6, reg_min=0.01, reg_scale=10.0)
    st2 = bt2.get("status")
    assert (st2 == -1).all(), st2
    assert (res2["reg_retries"][bad] >= 2).all() and (res2["reg_retries"][~bad] == 0).all(), res2["reg_retries"]
    assert np.isfinite(bt2.get("K")).all() and np.isfinite(bt2.get_nominal()[0]).all()
    assert np.array_equal(bt2.get_nominal()[0][~bad], good_x)     # well-posed problems are untouched by the option
    bt.close(); bt2.close()


@pytest.mark.parametrize("nn,mm,constrained", [(9, 3, False), (9, 3, True), (12, 2, True), (7, 4, False)])
def test_padded_shape_ilqr_solve_mfma16(nn, mm, constrained):
    """The iLQR loop of plan MFMA16 for n <= 12, m <= 4 (zero-padded tile records): whole solves --
    with input bounds and a state half-space when `constrained` -- against the oracle per problem, at the (12, 4)
    tolerances.  The constraint blocks are given over the problem's own [x; u] (n + m columns)."""
    Nh, batch = 17, 20
    p = problems.ilqr12x4_problem(batch, Nh, True, n=nn, m=mm)
    blocks = []
    if constrained:
        w = nn + mm
        Gb = np.zeros((2 * mm, w)); Gb[:mm, nn:] = np.eye(mm); Gb[mm:, nn:] = -np.eye(mm)
        Gs = np.zeros((1, w)); Gs[0, 1] = 1.0
        blocks = [(0, Nh - 1, problems.CONE_INEQUALITY, Gb, np.full(2 * mm, 0.3)), (1, Nh - 1, problems.CONE_INEQUALITY, Gs, np.array([1.2]))]
    bt = altro_amd.Batch(Nh, nn, mm, batch, plan=altro_amd.PLAN_MFMA16)
    bt.set_dynamics(p["A"], p["B"], p["f"])
    bt.set_tracking_cost(p["Qd"], p["Rd"], p["xref"], p["uref"])
    bt.set_initial_state(p["x0"])
    bt.set_input_guess(p["u0"])
    for (k0, k1, cone, G, g) in blocks:
        bt.add_linear_constraint(k0, k1, cone, G, g)
    res = bt.ilqr_solve(iterations_max=60, penalty_initial=1.0, penalty_scaling=10.0)
    x, u = bt.get_nominal()
    assert x.shape == (batch, Nh + 1, nn) and u.shape == (batch, Nh, mm)
    nconv = 0
    for b in [0, 9, 19]:
        s = oracle.ILQR(Nh, nn, mm, 0.01, oracle.DYN_LINEAR, cost_kind=oracle.COST_DIAGONAL)
        s.L.oracle_ilqr_set_linear_dynamics(s.h, np.ascontiguousarray(p["A"][b]), np.ascontiguousarray(p["B"][b]),
                                            np.ascontiguousarray(p["f"][b]).ctypes.data)
        for k in range(Nh + 1):
            s.L.oracle_ilqr_set_lqr_cost(s.h, k, np.ascontiguousarray(p["Qd"][b, k]), np.ascontiguousarray(p["Rd"][b, min(k, Nh - 1)]),
                                         np.ascontiguousarray(p["xref"][b, k]), np.ascontiguousarray(p["uref"][b, min(k, Nh - 1)]))
        s.L.oracle_ilqr_set_initial_state(s.h, np.ascontiguousarray(p["x0"][b]))
        for (k0, k1, cone, G, g) in blocks:
            for k in range(k0, k1 + 1):
                s.add_linear_constraint(k, cone, G, g)
        s.L.oracle_ilqr_initialize(s.h)
        for k in range(Nh):
            s.L.oracle_ilqr_set_input(s.h, k, np.ascontiguousarray(p["u0"][b, k]))
        s.set_penalty(1.0, 10.0)
        s.L.oracle_ilqr_set_options(s.h, 60, 1e-4, 1e-4, 1e-8, 0)
        status, iters, log = s.solve()
        assert res["status"][b] == status and res["iterations"][b] == iters, (b, res["status"][b], status, res["iterations"][b], iters)
        if status != 0:
            continue
        nconv += 1
        tol = 1e-7 if constrained else 1e-9
        np.testing.assert_allclose(x[b], s.get("x"), rtol=tol, atol=tol)
        np.testing.assert_allclose(u[b], s.get("u"), rtol=tol * 10, atol=tol * 10)
        if constrained:
            assert np.abs(u[b]).max() <= 0.3 + 2e-4
    assert nconv >= 2
    bt.close()


def test_expand_after_a_derivative_pass_has_nothing_to_do():
    """CalcExpansions right after MeritFunction with derivative (solver.cpp:448 after :455): the merit pass left lx, lu of its
    candidate in the records, so altro_hip_expand launches nothing -- and what it would have written is what is there, bit for
    bit (forced by touching the input guess, which invalidates the shortcut)."""
    batch = 11
    p = make_problem(batch, True)
    bt = make_hip(p)
    bt.open_loop_rollout(); bt.accept(); bt.expand(); bt.backward()
    bt.merit(np.linspace(0.1, 1.0, batch))
    _, _, lx1, lu1 = bt.get_expansion()
    bt.expand()                                       # no launch
    _, _, lx2, lu2 = bt.get_expansion()
    assert np.array_equal(lx1, lx2) and np.array_equal(lu1, lu2)
    bt.set_input_guess(bt.get("u"))                   # the same inputs again: the shortcut is off, the gradient pass runs
    bt.expand()
    _, _, lx3, lu3 = bt.get_expansion()
    assert np.array_equal(lx1, lx3) and np.array_equal(lu1, lu3)
    bt.merit(0.5, derivative=False)                   # a pass without derivative moves the candidate and leaves no expansion
    bt.expand()
    _, _, lx4, _ = bt.get_expansion()
    assert not np.array_equal(lx1, lx4)
    bt.close()


def _c1_bounded(x0, itmax):
    """bench.py's constrained C1 workload for the initial states x0: device handle solved with iterations_max = itmax."""
    Nf = 256
    one = problems.c1_double_integrator(1, N=Nf)
    bt = altro_amd.Batch(Nf, 12, 4, x0.shape[0])
    bt.set_dynamics(one["A"][0, :1], one["B"][0, :1], None, k_stride_zero=True, batch_stride_zero=True)
    Qd = np.stack([np.ones(12), 100.0 * np.ones(12)])
    bt.set_tracking_cost(Qd, np.full((1, 4), 1e-2), np.zeros((2, 12)), np.zeros((1, 4)), k_stride_zero=True, batch_stride_zero=True)
    bt.set_initial_state(x0)
    bt.set_input_guess(np.zeros((1, 1, 4)), k_stride_zero=True, batch_stride_zero=True)
    Gb = np.zeros((8, 16)); Gb[:4, 12:] = np.eye(4); Gb[4:, 12:] = -np.eye(4)
    bt.add_linear_constraint(0, Nf - 1, altro_amd.CONE_INEQUALITY, Gb, np.full(8, 2.0))
    res = bt.ilqr_solve(iterations_max=itmax)
    x, u = bt.get_nominal()
    bt.close()
    return res, x, u


def _c1_bounded_oracle(x0b, itmax):
    Nf = 256
    one = problems.c1_double_integrator(1, N=Nf)
    A = np.ascontiguousarray(np.tile(one["A"][0, :1], (Nf, 1))); B = np.ascontiguousarray(np.tile(one["B"][0, :1], (Nf, 1)))
    Gb = np.zeros((8, 16)); Gb[:4, 12:] = np.eye(4); Gb[4:, 12:] = -np.eye(4)
    s = oracle.ILQR(Nf, 12, 4, 0.01, oracle.DYN_LINEAR, cost_kind=oracle.COST_DIAGONAL)
    s.L.oracle_ilqr_set_linear_dynamics(s.h, A, B, None)
    for k in range(Nf + 1):
        s.L.oracle_ilqr_set_lqr_cost(s.h, k, np.ascontiguousarray(100.0 * np.ones(12) if k == Nf else np.ones(12)), np.full(4, 1e-2), np.zeros(12), np.zeros(4))
    s.L.oracle_ilqr_set_initial_state(s.h, np.ascontiguousarray(x0b))
    for k in range(Nf):
        s.add_linear_constraint(k, oracle.CONE_INEQUALITY, Gb, np.full(8, 2.0))
    s.L.oracle_ilqr_initialize(s.h)
    s.set_penalty(1.0, 10.0)
    s.L.oracle_ilqr_set_options(s.h, itmax, 1e-4, 1e-4, 1e-8, 0)
    status, iters, log = s.solve()
    return s, status, iters, log


def test_bench_constrained_workload_sample_vs_oracle():
    """bench.py's `config.ilqr_constrained_solve` at its own size: the C1 batch -- 4096 double integrators,
    N = 256, x0 ~ U(-1, 1)^12 -- with input bounds |u| <= 2 as one AL-iLQR solve (cubic line search, iterations_max 40).  A seeded
    sample of 32 problems INCLUDING the batch's stragglers (the problems that did not end in Success, the ones with the most
    iterations and dual updates) goes through oracle.ILQR one by one.

    What the batch's non-converged problems are (tests/soak/straggler_trace.py prints the iterations): after a penalty update one
    sweep's line search works at the resolution of the merit function itself -- e.g. phi(0) = 125.5648687, phi'(0) = -1.3e-3,
    accepted step ~1.5e-3 after 16 evaluations, i.e. a sufficient-decrease margin c1 alpha phi' ~ 2e-10 against values of 1e2 whose
    last digits (1e-13 relative) differ between any two orders of summation.  Device and oracle then resolve the search
    differently (one ends it `Success`, the other with a line-search failure, which by solver.cpp:450-453 stops the solve as
    `Unsolved`).  So: a problem either agrees outright -- status, iterations, dual updates, final step, feasibility, stationarity,
    trajectory 1e-7 -- or (1) it agrees with the oracle, trajectory included, up to the sweep before the first difference and
    (2) that sweep's line search is such a rounding-limited one in the oracle's own log (>= 8 evaluations or a step below 1e-2)."""
    Nf, batch = 256, 4096
    x0 = 2.0 * problems.uniform01((batch, 12), 21) - 1.0
    res, x, u = _c1_bounded(x0, 40)
    stragglers = np.flatnonzero(res["status"] != 0)
    order_it = np.argsort(-res["iterations"], kind="stable")
    order_du = np.argsort(-res["dual_updates"], kind="stable")
    rng = np.random.default_rng(20260929)
    pick = list(stragglers[:8]) + list(order_it[:6]) + list(order_du[:4]) + list(rng.choice(batch, size=64, replace=False))
    sample = []
    for b in pick:
        if int(b) not in sample:
            sample.append(int(b))
        if len(sample) == 32:
            break
    n_strag = sum(1 for b in sample if res["status"][b] != 0)
    assert len(stragglers) == 0 or n_strag >= min(4, len(stragglers))
    same, differ = 0, []
    for b in sample:
        s, status, iters, log = _c1_bounded_oracle(x0[b], 40)
        last = log[iters - 1]      # alpha, phi0, phi, dphi0, stationarity, ls_iters, feasibility, rho
        outright = (res["status"][b] == status and res["iterations"][b] == iters and
                    abs(res["alpha"][b] - last[0]) <= 1e-9 * max(1.0, abs(last[0])))
        if outright:
            assert abs(res["feasibility"][b] - last[6]) <= 1e-9 + 1e-5 * last[6], (b, res["feasibility"][b], last[6])
            assert abs(res["stationarity"][b] - last[4]) <= 1e-8 + 1e-4 * abs(last[4]), (b, res["stationarity"][b], last[4])
            assert abs(res["penalty"][b] - last[7]) <= 1e-12 * last[7]
            np.testing.assert_allclose(x[b], s.get("x"), rtol=1e-7, atol=1e-7)
            np.testing.assert_allclose(u[b], s.get("u"), rtol=1e-6, atol=1e-6)
            same += 1
        else:
            differ.append((b, int(res["status"][b]), int(res["iterations"][b]), status, iters, log))
    print("constrained C1 batch: %d of %d converged, %d stragglers; sample of 32 with %d of them: %d agree outright, %d differ at a "
          "rounding-limited line search" % ((res["status"] == 0).sum(), batch, len(stragglers), n_strag, same, len(differ)))
    assert same >= 16, (same, [(d[0], d[1], d[2], d[3], d[4]) for d in differ])
    for (b, st_d, it_d, st_o, it_o, log) in differ:
        upto = min(it_d, it_o)
        hard = [i for i in range(upto) if log[i][5] >= 8 or log[i][0] < 2e-2]      # sweeps (0-based) with a rounding-limited search
        assert hard, ("problem %d differs (device %d / %d, oracle %d / %d) without a rounding-limited line search" % (b, st_d, it_d, st_o, it_o),
                      log[:upto, [0, 5]])
        i_first = hard[0] + 1                         # 1-based: everything before this sweep must be identical
        if i_first >= 2:
            rd, xd, ud = _c1_bounded(x0[b:b + 1], i_first - 1)
            so, st2, it2, log2 = _c1_bounded_oracle(x0[b], i_first - 1)
            assert rd["iterations"][0] == it2 and rd["status"][0] == st2, (b, i_first, rd["iterations"][0], it2)
            np.testing.assert_allclose(xd[0], so.get("x"), rtol=1e-7, atol=1e-7)
            np.testing.assert_allclose(ud[0], so.get("u"), rtol=1e-6, atol=1e-6)
            ref_phi = log2[i_first - 2][2]          # (the log holds one row per sweep run: i_first - 1 of them)
            assert abs(rd["phi"][0] - ref_phi) <= 1e-10 * max(1.0, abs(ref_phi)), (b, rd["phi"][0], ref_phi)


def test_batch_level_early_return_leaves_finished_problems_untouched():
    """altro_hip_solve_options::stop_when_running_at_most (an extension for batches: the call returns after the first sweep that
    leaves at most k problems running).  512 input-bounded (12, 4) problems, N = 64: the early return takes fewer sweeps; every
    problem that stopped on its own reports bit for bit what the full solve reports (per-problem iterations are independent of the
    batch); the problems cut off report status 1 (Unsolved) with the iterations they took, and there are at most k of them more
    than in the full solve."""
    batch, Nf, k_stop = 512, 64, 12
    x0 = 2.0 * problems.uniform01((batch, 12), 77) - 1.0
    one = problems.c1_double_integrator(1, N=Nf)

    def solve(**kw):
        bt = altro_amd.Batch(Nf, 12, 4, batch)
        bt.set_dynamics(one["A"][0, :1], one["B"][0, :1], None, k_stride_zero=True, batch_stride_zero=True)
        bt.set_tracking_cost(np.stack([np.ones(12), 100.0 * np.ones(12)]), np.full((1, 4), 1e-2), np.zeros((2, 12)), np.zeros((1, 4)),
                             k_stride_zero=True, batch_stride_zero=True)
        bt.set_initial_state(x0)
        bt.set_input_guess(np.zeros((1, 1, 4)), k_stride_zero=True, batch_stride_zero=True)
        Gb = np.zeros((8, 16)); Gb[:4, 12:] = np.eye(4); Gb[4:, 12:] = -np.eye(4)
        bt.add_linear_constraint(0, Nf - 1, altro_amd.CONE_INEQUALITY, Gb, np.full(8, 4.0))
        res = bt.ilqr_solve(iterations_max=40, **kw)
        x, u = bt.get_nominal()
        bt.close()
        return res, x, u

    full, xf, uf = solve()
    early, xe, ue = solve(stop_when_running_at_most=k_stop)
    assert early["sweeps"] < full["sweeps"], (early["sweeps"], full["sweeps"])
    done = early["status"] == 0
    assert done.sum() >= batch - k_stop - int((full["status"] != 0).sum())
    assert np.array_equal(full["status"][done], early["status"][done])
    for key in ("iterations", "stationarity", "feasibility", "phi", "alpha", "dual_updates"):
        assert np.array_equal(full[key][done], early[key][done]), key
    assert np.array_equal(xf[done], xe[done]) and np.array_equal(uf[done], ue[done])
    cut = ~done & (full["status"] == 0)
    assert cut.sum() <= k_stop and (early["status"][cut] == 1).all()
    assert (early["iterations"][cut] <= early["sweeps"]).all() and (early["iterations"][cut] >= 1).all()
