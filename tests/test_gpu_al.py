"""GPU parity of the augmented-Lagrangian / conic part of the batched iLQR loop
against the CPU oracle (oracle/al_oracle.c + ilqr_oracle.c), which is pinned to the reference's constrained
double-integrator iteration counts 3 / 5 / 9 (tests/test_oracle_kat.py).  Constraint blocks are the ones of
test/double_integrator_test.cpp:170-493 and test/pendulum_test.cpp:117-203."""
import numpy as np
import pytest

import altro_amd
from oracle import oracle
from tests import problems

pytestmark = pytest.mark.gpu


def _di_hip(kats, kind, x0s):
    kat = kats["double_integrator_constrained"]
    N = kat["N"]; dim = kat["dim"]; n, m = 2 * dim, dim
    h = np.float32(np.float32(kat["tf"]) / np.float32(N))
    xf = np.array(kat["xf"], dtype=float)
    bt = altro_amd.Batch(N, n, m, x0s.shape[0])
    assert bt.plan == altro_amd.PLAN_LANE
    bt.set_model(altro_amd.MODEL_DOUBLE_INTEGRATOR, h)
    Qd = np.full(n, kat["Q"]); Rd = np.full(m, kat["R"])
    bt.set_tracking_cost(np.stack([Qd, Qd]), Rd[None], np.stack([xf, xf]), np.zeros((1, m)),
                         k_stride_zero=True, batch_stride_zero=True)
    bt.set_initial_state(x0s)
    bt.set_input_guess(np.zeros((1, 1, m)), k_stride_zero=True, batch_stride_zero=True)
    for (k0, k1, cone, G, g) in problems.di_constraint_blocks(kind, N, n, m, xf, kat["u_bnd"]):
        bt.add_linear_constraint(k0, k1, cone, G, g)
    return bt, kat, xf


def _di_oracle(kats, kind, x0):
    kat = kats["double_integrator_constrained"]
    c = kat[kind]
    N = kat["N"]; dim = kat["dim"]; n, m = 2 * dim, dim
    h = np.float32(np.float32(kat["tf"]) / np.float32(N))
    s = oracle.ILQR(N, n, m, h, oracle.DYN_MODEL, oracle.MODEL_DI, model_dim=dim, cost_kind=oracle.COST_DIAGONAL)
    xf = np.array(kat["xf"], dtype=float)
    for k in range(N + 1):
        s.L.oracle_ilqr_set_lqr_cost(s.h, k, np.full(n, kat["Q"]), np.full(m, kat["R"]), xf.copy(), np.zeros(m))
    s.L.oracle_ilqr_set_initial_state(s.h, np.ascontiguousarray(x0, dtype=float))
    for (k0, k1, cone, G, g) in problems.di_constraint_blocks(kind, N, n, m, xf, kat["u_bnd"]):
        for k in range(k0, k1 + 1):
            s.add_linear_constraint(k, cone, G, g)
    s.L.oracle_ilqr_initialize(s.h)
    s.set_penalty(c["penalty_initial"], c["penalty_scaling"])
    return s


@pytest.mark.parametrize("kind", ["goal", "bounds", "soc"])
def test_reference_iteration_counts_on_device(kats, kind):
    """double_integrator_test.cpp:255-256 / :366-375 / :482-492 through the batched device solver: Success,
    3 / 5 / 9 iterations, goal reached, controls saturated -- for every copy in the batch."""
    c = kats["double_integrator_constrained"][kind]
    x0s = np.tile(np.array(c["x0"], dtype=float), (65, 1))
    bt, kat, xf = _di_hip(kats, kind, x0s)
    res = bt.ilqr_solve(penalty_initial=c["penalty_initial"], penalty_scaling=c["penalty_scaling"])
    assert (res["status"] == 0).all(), res["status"]
    assert (res["iterations"] == c["iterations"]).all(), res["iterations"]
    x = bt.get("x"); u = bt.get("u")
    assert (np.linalg.norm(x[:, -1] - xf, axis=1) < c["goal_tol"]).all()
    if kind == "bounds":
        assert np.allclose(u[:, 0], c["u0"], atol=c["u0_tol"])
    if kind == "soc":
        assert (np.abs(np.linalg.norm(u[:, 0], axis=1) - c["u0_norm"]) < c["u0_norm_tol"]).all()
    assert (res["feasibility"] < 1e-4).all()
    assert np.array_equal(x[0], x[64])


@pytest.mark.parametrize("kind", ["goal", "bounds", "soc"])
def test_constrained_batch_matches_per_problem_oracle(kats, kind):
    """Heterogeneous batch: every problem follows the oracle's own iteration / dual-update path."""
    c = kats["double_integrator_constrained"][kind]
    batch = 80
    base = np.array(c["x0"], dtype=float)
    x0s = np.tile(base, (batch, 1))
    x0s[:, 0] += 0.05 * (np.arange(batch) % 9 - 4)
    x0s[:, 1] -= 0.04 * (np.arange(batch) % 7)
    x0s[:, 2] += 0.02 * (np.arange(batch) % 5)
    bt, kat, xf = _di_hip(kats, kind, x0s)
    res = bt.ilqr_solve(penalty_initial=c["penalty_initial"], penalty_scaling=c["penalty_scaling"], iterations_max=60)
    x = bt.get("x"); u = bt.get("u")
    tol = 1e-8 if kind != "soc" else 1e-6
    checked = 0
    for b in [0, 7, 23, 42, 79]:
        s = _di_oracle(kats, kind, x0s[b])
        s.L.oracle_ilqr_set_options(s.h, 60, 1e-4, 1e-4, 1e-8, 0)
        status, iters, log = s.solve()
        assert res["status"][b] == status, (b, res["status"][b], status)
        assert res["iterations"][b] == iters, (b, res["iterations"][b], iters)
        if status != 0:
            continue
        checked += 1
        assert abs(res["feasibility"][b] - log[iters - 1, 6]) <= 1e-9 + 1e-3 * log[iters - 1, 6]
        np.testing.assert_allclose(x[b], s.get("x"), rtol=tol, atol=tol)
        np.testing.assert_allclose(u[b], s.get("u"), rtol=tol * 10, atol=tol * 10)
    assert checked >= 3


def test_al_merit_and_expansion_parity(kats):
    """CalcCost / CalcCostGradient / CalcCostHessian with AL terms (knotpoint_data.cpp:421-448, :572-613) at a
    point with nonzero duals: run two sweeps on both sides, then compare phi, dphi, lx, lu."""
    kind = "soc"
    c = kats["double_integrator_constrained"][kind]
    x0s = np.tile(np.array(c["x0"], dtype=float), (4, 1))
    x0s[1] += [0.1, -0.2, 0.05, 0.0]
    bt, kat, xf = _di_hip(kats, kind, x0s)
    res = bt.ilqr_solve(penalty_initial=1.0, penalty_scaling=100.0, iterations_max=2)
    for b in [0, 1]:
        s = _di_oracle(kats, kind, x0s[b])
        s.L.oracle_ilqr_set_options(s.h, 2, 1e-4, 1e-4, 1e-8, 0)
        s.solve()
        A, B, lx, lu = bt.get_expansion()
        np.testing.assert_allclose(lx[b], s.get("lx"), rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(lu[b], s.get("lu"), rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(bt.get("x")[b], s.get("x"), rtol=1e-10, atol=1e-10)
        assert abs(res["feasibility"][b] - s.feasibility()) < 1e-10
        z_dev = bt.get_duals(0, 0, 3)[b]
        assert np.isfinite(z_dev).all()


def test_pendulum_goal_constraint_on_device(kats):
    """test/pendulum_test.cpp:117-203: c = xf - x (EQUALITY) at k = N; Success, < 1e-4 from the goal, <= 10."""
    kat = kats["pendulum_goal_constrained"]
    N = kat["N"]; n, m = 2, 1
    h = np.float32(np.float32(kat["tf"]) / float(N))
    xf = np.array(kat["xf_pi"]) * np.pi
    bt = altro_amd.Batch(N, n, m, 5)
    bt.set_model(altro_amd.MODEL_PENDULUM, h)
    bt.set_tracking_cost(np.stack([np.full(n, kat["Qd"]), np.full(n, kat["Qfd"])]), np.full((1, m), kat["Rd"]),
                         np.stack([xf, xf]), np.zeros((1, m)), k_stride_zero=True, batch_stride_zero=True)
    bt.set_initial_state(np.zeros((5, n)))
    bt.set_input_guess(np.full((1, 1, m), kat["u_init"]), k_stride_zero=True, batch_stride_zero=True)
    G = np.zeros((n, n + m)); G[:, :n] = -np.eye(n)
    # per-problem right-hand side: the same goal for every problem, sent through the [batch][p] path
    bt.add_linear_constraint(N, N, altro_amd.CONE_EQUALITY, G, np.tile(-xf, (5, 1)))
    res = bt.ilqr_solve(iterations_max=kat["iterations_max"])
    assert (res["status"] == 0).all()
    assert (res["iterations"] <= kat["max_iterations"]).all()
    assert (np.linalg.norm(bt.get("x")[:, -1] - xf, axis=1) < kat["goal_tol"]).all()


def test_constraint_argument_errors(kats):
    bt, kat, xf = _di_hip(kats, "goal", np.zeros((2, 4)))
    G = np.zeros((3, 6))
    with pytest.raises(altro_amd.AltroHipError):
        bt.add_linear_constraint(0, 99, altro_amd.CONE_EQUALITY, G, np.zeros(3))     # BadIndex
    with pytest.raises(altro_amd.AltroHipError):
        bt.add_linear_constraint(0, 1, 7, G, np.zeros(3))                            # unknown cone
    with pytest.raises(altro_amd.AltroHipError):
        bt.add_linear_constraint(0, 1, altro_amd.CONE_SOC, np.zeros((5, 6)), np.zeros(5))   # SOC rows > 4
    bt.add_linear_constraint(10, 10, altro_amd.CONE_INEQUALITY, G, np.zeros(3))
    with pytest.raises(altro_amd.AltroHipError):
        bt.add_linear_constraint(10, 10, altro_amd.CONE_INEQUALITY, G, np.zeros(3))  # third block at k = 10
    bt.clear_constraints()
    res = bt.ilqr_solve(iterations_max=5)
    assert (res["status"] == 0).all()


def test_regularisation_retry_extension(kats):
    """SURVEY.md section 8 row f4 (an extension, labelled as such): the reference keeps reg = 0 and ignores a
    failed Cholesky (tvlqr.cpp:159-164, solver.cpp:363, :449).  Default options reproduce that; with
    reg_retry_max > 0 the failed problems repeat the backward pass with a growing reg until it succeeds."""
    kat = kats["double_integrator_constrained"]
    N = kat["N"]; n, m = 4, 2
    h = np.float32(np.float32(kat["tf"]) / np.float32(N))
    batch = 66
    x0s = np.tile([1.0, 2.0, 0.0, 0.0], (batch, 1)) + 0.01 * np.arange(batch)[:, None]

    def make(Rval):
        bt = altro_amd.Batch(N, n, m, batch)
        bt.set_model(altro_amd.MODEL_DOUBLE_INTEGRATOR, h)
        Qd = np.ones(n)
        Rd = np.full((1, m), Rval)
        bt.set_tracking_cost(np.stack([Qd, Qd]), Rd, np.zeros((2, n)), np.zeros((1, m)), k_stride_zero=True, batch_stride_zero=True)
        bt.set_initial_state(x0s)
        bt.set_input_guess(np.zeros((1, 1, m)), k_stride_zero=True, batch_stride_zero=True)
        return bt
    # Quu = R + B'PB is indefinite with R = -0.5: the reference's behaviour is a failed backward pass that is ignored
    bt = make(-0.5)
    res = bt.ilqr_solve(iterations_max=2)
    assert (res["reg_retries"] == 0).all()
    assert (bt.get("status") != -1).all()
    # extension: retry with reg 0.01 -> 0.1 -> 1.0
    bt = make(-0.5)
    res = bt.ilqr_solve(iterations_max=2, reg_retry_max=5, reg_min=0.01, reg_scale=10.0)
    assert (res["reg_retries"] >= 3).all(), res["reg_retries"]
    assert (bt.get("status") == -1).all()
    assert np.isfinite(bt.get("K")).all()
    # a well-posed problem never retries and gives the same answer as without the option
    a = make(0.01); ra = a.ilqr_solve(iterations_max=5)
    b = make(0.01); rb = b.ilqr_solve(iterations_max=5, reg_retry_max=5, reg_min=0.01)
    assert (rb["reg_retries"] == 0).all()
    assert np.array_equal(a.get("x"), b.get("x")) and np.array_equal(ra["iterations"], rb["iterations"])
    with pytest.raises(altro_amd.AltroHipError):
        make(0.01).ilqr_solve(reg_retry_max=2, reg_scale=0.5)


def test_full_size_c3_constrained_tracking_batch():
    """BASELINE.json configs[3] at full size on one GPU (bicycle, N = 50, 65536 vehicles, steering bound as an
    INEQUALITY block at every knot point): size-independent properties of the batched solve -- nearly every vehicle
    converges, converged ones are feasible and stationary, copies of one problem give identical bits."""
    from tests import mpc_common as M
    N, n, m, batch = 50, 4, 2, 65536
    x_ref, u_ref = problems.bicycle_reference(N + 1)
    bt = altro_amd.Batch(N, n, m, batch)
    bt.set_model(altro_amd.MODEL_BICYCLE, np.float32(0.1))
    bt.set_tracking_cost(np.full((1, N + 1, n), 1e-2), np.full((1, N, m), 1e-3), x_ref[None, :N + 1], u_ref[None, :N],
                         batch_stride_zero=True)
    cone, G, g = M.steering_block()
    bt.add_linear_constraint(0, N, cone, G, g)
    x0 = x_ref[0] + (problems.uniform01((batch, n), 23) - 0.5) * 0.4
    x0[batch // 2] = x0[0]            # two copies of one problem, far apart in the batch
    bt.set_initial_state(x0)
    bt.set_input_guess(np.array([[[u_ref[0][0], 0.0]]]), k_stride_zero=True, batch_stride_zero=True)
    res = bt.ilqr_solve(iterations_max=40, use_backtracking=True)
    ok = res["status"] == 0
    assert ok.mean() > 0.995, ok.mean()
    assert (res["feasibility"][ok] < 1e-4).all() and (res["stationarity"][ok] < 1e-4).all()
    assert set(np.unique(res["status"])) <= {0, 1, 2}
    x, u = bt.get_nominal()
    assert np.abs(x[ok][:, :, 3]).max() <= M.DELTA_MAX + 1e-4
    assert np.array_equal(x[0], x[batch // 2]) and res["iterations"][0] == res["iterations"][batch // 2]
    assert np.isfinite(x).all()
    # eight seeded vehicles of the full batch against the oracle solving each alone: status, iterations
    # and the trajectory (5e-5, the band of this file's backtracked constrained solves; measured value printed)
    from oracle import oracle
    worst = 0.0
    for b in [0, 1, 4099, 16384, 32768, 40001, 65000, 65535]:
        s = oracle.ILQR(N, n, m, np.float32(0.1), oracle.DYN_MODEL, oracle.MODEL_BICYCLE, cost_kind=oracle.COST_DIAGONAL)
        for k in range(N + 1):
            s.L.oracle_ilqr_set_lqr_cost(s.h, k, np.full(n, 1e-2), np.full(m, 1e-3), np.ascontiguousarray(x_ref[k]),
                                         np.ascontiguousarray(u_ref[min(k, N - 1)]))
        for k in range(N + 1):
            s.add_linear_constraint(k, cone, G, g)
        s.L.oracle_ilqr_set_initial_state(s.h, np.ascontiguousarray(x0[b]))
        s.L.oracle_ilqr_initialize(s.h)
        for k in range(N):
            s.L.oracle_ilqr_set_input(s.h, k, np.array([u_ref[0][0], 0.0]))
        s.L.oracle_ilqr_set_options(s.h, 40, 1e-4, 1e-4, 1e-8, 1)
        status, iters, _ = s.solve()
        assert res["status"][b] == status and res["iterations"][b] == iters, (b, res["status"][b], status, res["iterations"][b], iters)
        worst = max(worst, np.abs(x[b] - s.get("x")).max(), np.abs(u[b] - s.get("u")).max())
    print("C3 full size, 8 sampled vehicles vs the oracle: max trajectory difference %.3g" % worst)
    assert worst < 5e-5


def test_per_problem_bounds_match_oracle(kats):
    """Per-problem right-hand sides (g_per_problem): every problem of the batch has its own control bound; each must
    follow the oracle solving it alone with that bound."""
    kat = kats["double_integrator_constrained"]
    c = kat["bounds"]
    N = kat["N"]; n, m = 4, 2
    batch = 48
    x0s = np.tile(np.array(c["x0"], dtype=float), (batch, 1))
    x0s[:, 0] += 0.03 * (np.arange(batch) % 11 - 5)
    ubs = 0.8 + 0.05 * (np.arange(batch) % 9)                     # 0.8 .. 1.2
    bt, _, xf = _di_hip(kats, "goal", x0s)                        # goal block at k = N
    Gb = np.zeros((2 * m, n + m)); Gb[:m, n:] = np.eye(m); Gb[m:, n:] = -np.eye(m)
    bt.add_linear_constraint(0, N - 1, altro_amd.CONE_INEQUALITY, Gb, np.repeat(ubs[:, None], 2 * m, axis=1))
    res = bt.ilqr_solve(penalty_initial=c["penalty_initial"], penalty_scaling=c["penalty_scaling"], iterations_max=60)
    x, u = bt.get("x"), bt.get("u")
    checked = 0
    for b in [0, 8, 13, 30, 47]:
        s = _di_oracle(kats, "goal", x0s[b])
        for k in range(N):
            s.add_linear_constraint(k, oracle.CONE_INEQUALITY, Gb, np.full(2 * m, ubs[b]))
        s.L.oracle_ilqr_initialize(s.h)
        s.set_penalty(c["penalty_initial"], c["penalty_scaling"])
        s.L.oracle_ilqr_set_options(s.h, 60, 1e-4, 1e-4, 1e-8, 0)
        status, iters, log = s.solve()
        assert res["status"][b] == status and res["iterations"][b] == iters, (b, res["status"][b], status, res["iterations"][b], iters)
        if status != 0:
            continue
        checked += 1
        np.testing.assert_allclose(x[b], s.get("x"), rtol=1e-8, atol=1e-8)
        assert np.abs(u[b]).max() <= ubs[b] + 2e-4
    assert checked >= 3


@pytest.mark.parametrize("dim", [1, 3])
def test_constrained_double_integrator_other_dims(dim):
    """The LANE AL path at its smallest and largest shapes, (2, 1) and (6, 3): goal (EQUALITY) + control bounds
    (INEQUALITY) + a second-order-cone bound on (6, 3), against the oracle per problem."""
    n, m, N = 2 * dim, dim, 12
    h = np.float32(0.4)
    batch = 66
    x0s = np.zeros((batch, n)); x0s[:, :dim] = 1.5 + 0.02 * (np.arange(batch) % 13)[:, None]
    bt = altro_amd.Batch(N, n, m, batch)
    bt.set_model(altro_amd.MODEL_DOUBLE_INTEGRATOR, h)
    assert bt.plan == altro_amd.PLAN_LANE      # ((6, 3) under plan AUTO: the padded tile until a LANE-only model arrives)
    Qd = np.ones(n); Rd = np.full(m, 1e-2)
    bt.set_tracking_cost(np.stack([Qd, Qd]), Rd[None], np.zeros((2, n)), np.zeros((1, m)), k_stride_zero=True, batch_stride_zero=True)
    bt.set_initial_state(x0s)
    bt.set_input_guess(np.zeros((1, 1, m)), k_stride_zero=True, batch_stride_zero=True)
    w = n + m
    Gg = np.zeros((n, w)); Gg[:, :n] = np.eye(n)
    blocks = [(N, N, problems.CONE_EQUALITY, Gg, np.zeros(n))]
    if dim == 3:
        Gs = np.zeros((m + 1, w)); Gs[:m, n:] = np.eye(m)
        gs = np.zeros(m + 1); gs[m] = -1.2
        blocks.append((0, N - 1, problems.CONE_SOC, Gs, gs))
    else:
        Gb = np.zeros((2 * m, w)); Gb[:m, n:] = np.eye(m); Gb[m:, n:] = -np.eye(m)
        blocks.append((0, N - 1, problems.CONE_INEQUALITY, Gb, np.full(2 * m, 0.9)))
    for (k0, k1, cone, G, g) in blocks:
        bt.add_linear_constraint(k0, k1, cone, G, g)
    res = bt.ilqr_solve(iterations_max=80, penalty_initial=1.0, penalty_scaling=100.0)
    x, u = bt.get("x"), bt.get("u")
    checked = 0
    for b in [0, 31, 65]:
        s = oracle.ILQR(N, n, m, h, oracle.DYN_MODEL, oracle.MODEL_DI, model_dim=dim, cost_kind=oracle.COST_DIAGONAL)
        for k in range(N + 1):
            s.L.oracle_ilqr_set_lqr_cost(s.h, k, Qd.copy(), Rd.copy(), np.zeros(n), np.zeros(m))
        s.L.oracle_ilqr_set_initial_state(s.h, np.ascontiguousarray(x0s[b]))
        for (k0, k1, cone, G, g) in blocks:
            for k in range(k0, k1 + 1):
                s.add_linear_constraint(k, cone, G, g)
        s.L.oracle_ilqr_initialize(s.h)
        s.set_penalty(1.0, 100.0)
        s.L.oracle_ilqr_set_options(s.h, 80, 1e-4, 1e-4, 1e-8, 0)
        status, iters, log = s.solve()
        assert res["status"][b] == status and res["iterations"][b] == iters, (dim, b, res["status"][b], status, res["iterations"][b], iters)
        if status != 0:
            continue
        checked += 1
        tol = 1e-6 if dim == 3 else 1e-8
        np.testing.assert_allclose(x[b], s.get("x"), rtol=tol, atol=tol)
    assert checked >= 2


def test_constraint_capacity_is_stated_and_enforced():
    """The reference appends constraints without limit (knotpoint_data.cpp:155-178); the device tables hold at most 2 blocks per
    knot point, 8 rows per block (4 for a second-order cone) and 16 block definitions per handle (kernels/al_types.h, stated in
    altro_hip.h next to altro_hip_add_linear_constraint).  Going past a limit is an error that says which one -- never a silent
    truncation; rows of the same cone can be stacked into one block (two 4-row bound blocks == one 8-row block)."""
    N, n, m = 10, 4, 2
    bt = altro_amd.Batch(N, n, m, 3)
    w = n + m
    G1 = np.zeros((1, w)); G1[0, 4] = 1.0
    bt.add_linear_constraint(0, N - 1, altro_amd.CONE_INEQUALITY, G1, np.array([1.0]))
    bt.add_linear_constraint(0, N - 1, altro_amd.CONE_INEQUALITY, -G1, np.array([1.0]))
    with pytest.raises(altro_amd.AltroHipError, match="at most 2 constraint blocks per knot point on this plan"):
        bt.add_linear_constraint(3, 3, altro_amd.CONE_EQUALITY, G1, np.array([0.0]))
    with pytest.raises(altro_amd.AltroHipError, match=r"constraint dimension 9 outside \[1, 8\]"):
        bt.add_linear_constraint(N, N, altro_amd.CONE_INEQUALITY, np.zeros((9, w)), np.zeros(9))
    with pytest.raises(altro_amd.AltroHipError, match=r"constraint dimension 5 outside \[1, 4\]"):
        bt.add_linear_constraint(N, N, altro_amd.CONE_SOC, np.zeros((5, w)), np.zeros(5))
    bt2 = altro_amd.Batch(40, n, m, 3)
    for k in range(16):
        bt2.add_linear_constraint(k, k, altro_amd.CONE_INEQUALITY, G1, np.array([1.0]))
    with pytest.raises(altro_amd.AltroHipError, match="at most 16 constraint blocks"):
        bt2.add_linear_constraint(20, 20, altro_amd.CONE_INEQUALITY, G1, np.array([1.0]))
    bt.close(); bt2.close()
    # plan GENERIC (round 5): 8 blocks per knot point, 64 rows per block (one lane each), 64 blocks per handle
    bt3 = altro_amd.Batch(N, n, m, 3, plan=altro_amd.PLAN_GENERIC)
    for j in range(8):
        bt3.add_linear_constraint(0, N - 1, altro_amd.CONE_INEQUALITY, np.tile(G1, (w, 1)), np.full(w, 1.0 + j))
    with pytest.raises(altro_amd.AltroHipError, match="at most 8 constraint blocks per knot point"):
        bt3.add_linear_constraint(3, 3, altro_amd.CONE_EQUALITY, G1, np.array([0.0]))
    with pytest.raises(altro_amd.AltroHipError, match=r"constraint dimension 65 outside \[1, 64\]"):
        bt3.add_linear_constraint(N, N, altro_amd.CONE_INEQUALITY, np.zeros((65, w)), np.zeros(65))
    bt3.close()


@pytest.mark.parametrize("plan", ["generic", "auto"])
def test_input_box_and_state_box_on_a_12x4_problem(plan):
    """(plan "auto", round 6, VERDICT r5 item 4: the same problem on the handle ALTRO_HIP_PLAN_AUTO gives a (12, 4) problem -- plan MFMA16,
    whose knot-point record now holds six slots of eight rows: the 24-row box takes three of them.)
    VERDICT r4 missing #2: a (12, 4) problem with an input box (8 rows) AND a state box (24 rows) could not be posed -- 2 blocks of 8
    rows per knot point.  Plan GENERIC now holds 8 blocks per knot point and 64 rows per block (one lane per row).  Here: |u| <= 0.8 at
    k < N, |x_i| <= loose at every k (24 rows, evaluated everywhere, binding nowhere), u_0[0] = 0.05 at k = 0 -- three blocks at k = 0 --
    and a TIGHT 24-row state box at k = N that binds (0.7 of the largest terminal state the input-bounded solves reach; these random
    dynamics have little control authority, so some problems cannot meet it: the reference algorithm's non-convergence, status 1, which
    the device must reproduce too).  Every sampled problem: the oracle's status, iteration count and trajectory (1e-7)."""
    from tests.test_gpu_ilqr_generic import make_oracle
    N, n, m, batch = 12, 12, 4, 6
    w = n + m
    p = problems.ilqr12x4_problem(batch, N, True, n=n, m=m)
    Gu = np.zeros((2 * m, w)); Gu[:m, n:] = np.eye(m); Gu[m:, n:] = -np.eye(m)
    Gx = np.zeros((2 * n, w)); Gx[:n, :n] = np.eye(n); Gx[n:, :n] = -np.eye(n)
    Ge = np.zeros((1, w)); Ge[0, n] = 1.0
    ub = 0.8

    def build(blocks):
        bt = altro_amd.Batch(N, n, m, batch, plan=altro_amd.PLAN_GENERIC if plan == "generic" else altro_amd.PLAN_AUTO)
        assert bt.plan == (altro_amd.PLAN_GENERIC if plan == "generic" else altro_amd.PLAN_MFMA16)
        bt.set_dynamics(p["A"], p["B"], p["f"])
        bt.set_tracking_cost(p["Qd"], p["Rd"], p["xref"], p["uref"])
        bt.set_initial_state(p["x0"]); bt.set_input_guess(p["u0"])
        for (k0, k1, cone, G, g) in blocks:
            bt.add_linear_constraint(k0, k1, cone, G, g)
        return bt
    bt0 = build([(0, N - 1, altro_amd.CONE_INEQUALITY, Gu, np.full(2 * m, ub))])
    bt0.ilqr_solve(iterations_max=80, penalty_initial=1.0, penalty_scaling=10.0)
    x0s, _ = bt0.get_nominal()
    bt0.close()
    tight = 0.7 * float(np.abs(x0s[:, N]).max())
    loose = float(np.abs(x0s).max()) * 1.5
    blocks = [(0, N - 1, altro_amd.CONE_INEQUALITY, Gu, np.full(2 * m, ub)), (0, N - 1, altro_amd.CONE_INEQUALITY, Gx, np.full(2 * n, loose)),
              (0, 0, altro_amd.CONE_EQUALITY, Ge, np.array([0.05])), (N, N, altro_amd.CONE_INEQUALITY, Gx, np.full(2 * n, tight))]
    bt = build(blocks)
    res = bt.ilqr_solve(iterations_max=80, penalty_initial=1.0, penalty_scaling=10.0)
    x, u = bt.get_nominal()
    nconv = nbind = 0
    for b in range(batch):
        s = make_oracle(p, b, N, n, m, False)
        for (k0, k1, cone, G, g) in blocks:
            for k in range(k0, k1 + 1):
                s.add_linear_constraint(k, cone, G, g)
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
        np.testing.assert_allclose(x[b], s.get("x"), rtol=1e-7, atol=1e-7)
        np.testing.assert_allclose(u[b], s.get("u"), rtol=1e-6, atol=1e-6)
        assert np.abs(u[b]).max() <= ub + 2e-4 and np.abs(x[b][N]).max() <= tight + 2e-4 and abs(u[b][0, 0] - 0.05) < 2e-4
        nbind += int(np.abs(x[b][N]).max() >= tight - 1e-3)
    assert nconv >= 2 and nbind >= 1, (nconv, nbind)
    z = bt.get_duals(N, 0, 2 * n)
    assert z.shape == (batch, 2 * n) and (z <= 1e-12).all()
    z1 = bt.get_duals(3, 1, 2 * n)
    assert np.abs(z1).max() == 0.0            # the loose box never bound: its duals stayed zero
    bt.close()
