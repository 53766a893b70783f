"""GPU parity of the iLQR loop around the sweep (device models, plan LANE) against the CPU oracle's
restatement of SolverImpl (oracle/ilqr_oracle.c), which is itself pinned to the reference's
merit-function constants, pendulum end state and iteration counts (tests/test_oracle_kat.py).

Tolerances: trigonometric functions differ in the last ulp between the device library and glibc, so
nonlinear models are compared at 1e-10 relative; the double integrator (no transcendental) at 1e-13."""
import numpy as np
import pytest

import altro_amd
from oracle import oracle

pytestmark = pytest.mark.gpu


def make_oracle(model_kind, N, n, m, h, Qd, Rd, Qfd, xf, x0, u0, dim=0):
    s = oracle.ILQR(N, n, m, h, oracle.DYN_MODEL, model_kind, model_dim=dim, cost_kind=oracle.COST_DIAGONAL)
    for k in range(N + 1):
        s.L.oracle_ilqr_set_lqr_cost(s.h, k, np.ascontiguousarray(Qfd if k == N else Qd, dtype=float),
                                     np.ascontiguousarray(Rd, dtype=float), np.ascontiguousarray(xf, dtype=float), np.zeros(m))
    s.L.oracle_ilqr_set_initial_state(s.h, np.ascontiguousarray(x0, dtype=float))
    s.L.oracle_ilqr_initialize(s.h)
    for k in range(N):
        s.L.oracle_ilqr_set_input(s.h, k, np.ascontiguousarray(u0, dtype=float))
    return s


def make_hip(model, N, n, m, h, Qd, Rd, Qfd, xf, x0s, u0, dtype=altro_amd.F64):
    batch = x0s.shape[0]
    bt = altro_amd.Batch(N, n, m, batch, dtype=dtype)
    assert bt.plan == altro_amd.PLAN_LANE
    bt.set_model(model, h)
    bt.set_tracking_cost(np.stack([Qd, Qfd]), np.asarray(Rd)[None], np.stack([xf, xf]), np.zeros((1, m)),
                         k_stride_zero=True, batch_stride_zero=True)
    bt.set_initial_state(x0s)
    bt.set_input_guess(np.asarray(u0, dtype=float)[None, None], k_stride_zero=True, batch_stride_zero=True)
    return bt


CASES = {
    "pendulum": dict(model=altro_amd.MODEL_PENDULUM, okind=oracle.MODEL_PENDULUM, N=50, n=2, m=1,
                     h=np.float32(np.float32(3.0) / 50.0), Qd=[1e-2, 1e-2], Rd=[1e-3], Qfd=[1.0, 1.0],
                     xf=[np.pi, 0.0], u0=[0.1], x0=lambda b: np.stack([np.linspace(-0.5, 0.5, b), np.zeros(b)], 1), dim=0),
    "bicycle": dict(model=altro_amd.MODEL_BICYCLE, okind=oracle.MODEL_BICYCLE, N=30, n=4, m=2,
                    h=np.float32(np.float32(3.0) / 30.0), Qd=[1e-2] * 4, Rd=[1e-3] * 2, Qfd=[10.0] * 4,
                    xf=[1, 2, np.pi / 2, 0.0], u0=[0.5, 0.0],
                    x0=lambda b: 0.02 * (np.arange(b * 4).reshape(b, 4) % 7 - 3) / 3.0, dim=0),
    "double_integrator": dict(model=altro_amd.MODEL_DOUBLE_INTEGRATOR, okind=oracle.MODEL_DI, N=10, n=4, m=2,
                              h=np.float32(0.5), Qd=[1.0] * 4, Rd=[1e-2] * 2, Qfd=[1.0] * 4, xf=[0.0] * 4,
                              u0=[0.0, 0.0], x0=lambda b: np.tile([1.0, 2.0, 0, 0], (b, 1)) + 0.1 * np.arange(b)[:, None], dim=2),
}


@pytest.mark.parametrize("name", list(CASES))
def test_merit_function_parity(name):
    """solver.cpp:273-355 on the device vs the oracle: phi, dphi, candidate x_/u_/y_, refreshed A, B, lx, lu."""
    c = CASES[name]
    batch = 70
    x0s = c["x0"](batch)
    bt = make_hip(c["model"], c["N"], c["n"], c["m"], c["h"], c["Qd"], c["Rd"], c["Qfd"], c["xf"], x0s, c["u0"])
    bt.open_loop_rollout(); bt.accept(); bt.expand(); bt.backward()
    assert (bt.get("status") == -1).all()
    alphas = np.linspace(0.0, 1.5, batch)
    phi, dphi = bt.merit(alphas)
    xc, uc, yc = bt.get("x"), bt.get("u"), bt.get("y")
    A, B, lx, lu = bt.get_expansion()
    tol = 1e-13 if name == "double_integrator" else 1e-10
    for b in [0, 1, 17, 69]:
        s = make_oracle(c["okind"], c["N"], c["n"], c["m"], c["h"], c["Qd"], c["Rd"], c["Qfd"], c["xf"], x0s[b], c["u0"], c["dim"])
        s.L.oracle_ilqr_open_loop_rollout(s.h); s.L.oracle_ilqr_copy_trajectory(s.h)
        s.L.oracle_ilqr_calc_dynamics_expansions(s.h); s.L.oracle_ilqr_calc_cost_gradient(s.h)
        s.L.oracle_ilqr_calc_expansions(s.h)
        assert s.L.oracle_ilqr_backward_pass(s.h) == -1
        np.testing.assert_allclose(bt.get("K")[b], s.get("K"), rtol=tol, atol=tol)
        p_ref, dp_ref = s.merit(alphas[b])
        assert abs(phi[b] - p_ref) <= tol * max(1.0, abs(p_ref))
        assert abs(dphi[b] - dp_ref) <= tol * max(1.0, abs(dp_ref)) * 10
        np.testing.assert_allclose(xc[b], s.get("x_cand"), rtol=tol, atol=tol)
        np.testing.assert_allclose(uc[b], s.get("u_cand"), rtol=tol, atol=tol)
        np.testing.assert_allclose(yc[b], s.get("y_cand"), rtol=tol * 10, atol=tol * 10)
        np.testing.assert_allclose(A[b], s.get("A"), rtol=tol, atol=tol)
        np.testing.assert_allclose(B[b], s.get("B"), rtol=tol, atol=tol)
        np.testing.assert_allclose(lx[b], s.get("lx"), rtol=tol, atol=tol)
        np.testing.assert_allclose(lu[b], s.get("lu"), rtol=tol, atol=tol)
    # the uniform-alpha entry point is the same computation
    phi1, dphi1 = bt.merit(0.75)
    phi2, dphi2 = bt.merit(np.full(batch, 0.75))
    assert np.array_equal(phi1, phi2) and np.array_equal(dphi1, dphi2)


def test_pendulum_solve_reference_pin(kats):
    """test/pendulum_test.cpp:45-115 through the batched device solver: xN to 1e-5, <= 10 iterations."""
    kat = kats["pendulum_solve"]
    c = CASES["pendulum"]
    bt = make_hip(c["model"], c["N"], 2, 1, c["h"], c["Qd"], c["Rd"], c["Qfd"], c["xf"], np.zeros((3, 2)), c["u0"])
    res = bt.ilqr_solve(iterations_max=20)
    assert (res["status"] == 0).all()
    assert (res["iterations"] <= kat["max_iterations"]).all()
    xN = bt.get("x")[:, -1]
    assert np.linalg.norm(xN[0] - np.array(kat["xN"])) < kat["tol"]
    assert np.array_equal(xN[0], xN[2])


def test_bicycle_turn90_reference_pin(kats):
    """test/bicycle_test.cpp:53-138 through the batched device solver: from rest, backtracking line search, at most 30
    iterations, the car ends within 1e-2 of (1, 2, pi/2, 0)."""
    kat = kats["bicycle_turn90"]
    c = CASES["bicycle"]
    bt = make_hip(c["model"], c["N"], 4, 2, c["h"], c["Qd"], c["Rd"], c["Qfd"], c["xf"], np.zeros((3, 4)), c["u0"])
    res = bt.ilqr_solve(iterations_max=kat["iterations_max"], use_backtracking=True)
    xN = bt.get("x")[:, -1]
    assert np.linalg.norm(xN[0] - np.asarray(c["xf"], dtype=float)) < kat["goal_tol"]
    assert np.array_equal(xN[0], xN[2])
    assert (res["iterations"] <= kat["iterations_max"] + 1).all()


@pytest.mark.parametrize("name", list(CASES))
def test_solve_agrees_iteration_by_iteration(name):
    """Where the arithmetic is the same the comparison is tight: stop the device solve and the oracle after 1, 2, 3, 4
    iterations from the same start and compare the nominal trajectory, the step length, phi and the stationarity.
    The only differences are last-ulp ones (device vs glibc sin / cos; one sincos per angle and the addition theorems
    on the device); they are amplified by at most ~10x per back-tracked iteration, which the tolerances spell out.
    EVERY sampled problem is compared at every count (no "k of n" allowance)."""
    c = CASES[name]
    batch = 96
    x0s = c["x0"](batch)
    bk = name == "bicycle"
    bt = make_hip(c["model"], c["N"], c["n"], c["m"], c["h"], c["Qd"], c["Rd"], c["Qfd"], c["xf"], x0s, c["u0"])
    base = {"double_integrator": 1e-13, "pendulum": 1e-10, "bicycle": 1e-10}[name]
    growth = {"double_integrator": 1.0, "pendulum": 10.0, "bicycle": 10.0}[name]
    worst = {}
    for iters in (1, 2, 3, 4):
        bt.set_input_guess(np.asarray(c["u0"], dtype=float)[None, None], k_stride_zero=True, batch_stride_zero=True)
        res = bt.ilqr_solve(iterations_max=iters, use_backtracking=bk)
        x, u = bt.get_nominal()
        tol = base * growth ** (iters - 1)
        for b in [0, 5, 31, 64, 95]:
            s = make_oracle(c["okind"], c["N"], c["n"], c["m"], c["h"], c["Qd"], c["Rd"], c["Qfd"], c["xf"], x0s[b], c["u0"], c["dim"])
            s.L.oracle_ilqr_set_options(s.h, iters, 1e-4, 1e-4, 1e-8, int(bk))
            status, it_o, log = s.solve()
            assert (res["status"][b], res["iterations"][b]) == (status, it_o), (iters, b)
            last = log[min(it_o, iters) - 1]
            # same line-search decisions (a cubic-interpolated step inherits phi's last digits)
            assert abs(res["alpha"][b] - last[0]) <= 1e3 * tol * max(1.0, abs(last[0])), (iters, b, res["alpha"][b], last[0])
            ex = np.abs(x[b] - s.get("x")).max() / max(1.0, np.abs(s.get("x")).max())
            eu = np.abs(u[b] - s.get("u")).max() / max(1.0, np.abs(s.get("u")).max())
            ephi = abs(res["phi"][b] - last[2]) / max(1.0, abs(last[2]))
            est = abs(res["stationarity"][b] - last[4]) / max(abs(last[4]), 1e-3)
            worst[iters] = max(worst.get(iters, 0.0), ex, eu)
            assert ex < tol and eu < 10 * tol, (name, iters, b, ex, eu, tol)
            assert ephi < 10 * tol, (name, iters, b, ephi)
            assert est < 1e4 * tol, (name, iters, b, est)    # a residual: relative to its own (small) size
    print("iteration-by-iteration worst trajectory deviation", name, worst)


@pytest.mark.parametrize("name", list(CASES))
def test_batched_solve_matches_per_problem_oracle(name):
    """Every problem of a heterogeneous batch must take the same iterations / line-search path as the
    oracle solving it alone (masked per-problem line search) and land on the same trajectory.  The per-iteration test
    above is the tight one; here the whole solves (up to 30 back-tracked iterations) are compared: status, iteration
    count and final step length of EVERY sampled problem, and the trajectories of every one that converges -- the
    samples are chosen among problems that do -- at the amplified tolerance stated per model."""
    c = CASES[name]
    batch = 96
    x0s = c["x0"](batch)
    bt = make_hip(c["model"], c["N"], c["n"], c["m"], c["h"], c["Qd"], c["Rd"], c["Qfd"], c["xf"], x0s, c["u0"])
    bk = name == "bicycle"
    res = bt.ilqr_solve(iterations_max=30, use_backtracking=bk)
    x = bt.get("x"); u = bt.get("u")
    tol = {"double_integrator": 1e-11, "pendulum": 2e-7, "bicycle": 5e-5}[name]
    conv = np.flatnonzero(res["status"] == 0)
    capped = np.flatnonzero(res["status"] != 0)
    sample = list(conv[np.linspace(0, len(conv) - 1, 8).astype(int)]) + list(capped[:2])
    assert len(conv) >= batch // 2
    for b in sample:
        s = make_oracle(c["okind"], c["N"], c["n"], c["m"], c["h"], c["Qd"], c["Rd"], c["Qfd"], c["xf"], x0s[b], c["u0"], c["dim"])
        s.L.oracle_ilqr_set_options(s.h, 30, 1e-4, 1e-4, 1e-8, int(bk))
        status, iters, log = s.solve()
        assert res["status"][b] == status, (b, res["status"][b], status)
        assert res["iterations"][b] == iters, (b, res["iterations"][b], iters)
        if status != 0:
            continue   # a run that hits the iteration cap is chaotic in the last digits: outcome only
        assert abs(res["alpha"][b] - log[iters - 1, 0]) <= 1e-6
        np.testing.assert_allclose(x[b], s.get("x"), rtol=tol, atol=tol)
        np.testing.assert_allclose(u[b], s.get("u"), rtol=tol * 10, atol=tol * 10)
        # the final stationarity is a residual at rounding level after many back-tracked iterations: both below the
        # convergence tolerance, and equal to within the trajectory tolerance scaled by the residual's conditioning
        assert res["stationarity"][b] < 1e-4 and log[iters - 1, 4] < 1e-4
        assert abs(res["stationarity"][b] - log[iters - 1, 4]) <= 1e3 * tol
    assert res["sweeps"] <= 30


def test_iteration_cap_reports_like_reference():
    """solver.cpp:503-506: hitting iterations_max gives MaxIterations and iterations = max + 1."""
    c = CASES["pendulum"]
    bt = make_hip(c["model"], c["N"], 2, 1, c["h"], c["Qd"], c["Rd"], c["Qfd"], c["xf"], np.zeros((2, 2)), c["u0"])
    res = bt.ilqr_solve(iterations_max=2)
    s = make_oracle(c["okind"], c["N"], 2, 1, c["h"], c["Qd"], c["Rd"], c["Qfd"], c["xf"], np.zeros(2), c["u0"])
    s.L.oracle_ilqr_set_options(s.h, 2, 1e-4, 1e-4, 1e-8, 0)
    status, iters, _ = s.solve()
    assert res["status"].tolist() == [status, status] == [2, 2]
    assert res["iterations"].tolist() == [iters, iters] == [3, 3]


def test_fp32_lane_solver_tracks_fp64():
    """ALTRO_HIP_F32 on plan LANE (fp32 storage and arithmetic): the pendulum swing-up converges and lands
    within 1e-2 of the fp64 solve (tolerances relaxed to what fp32 stationarity can resolve)."""
    c = CASES["pendulum"]
    x0s = c["x0"](77)
    b64 = make_hip(c["model"], c["N"], 2, 1, c["h"], c["Qd"], c["Rd"], c["Qfd"], c["xf"], x0s, c["u0"])
    b32 = make_hip(c["model"], c["N"], 2, 1, c["h"], c["Qd"], c["Rd"], c["Qfd"], c["xf"], x0s, c["u0"], dtype=altro_amd.F32)
    r64 = b64.ilqr_solve(iterations_max=30)
    r32 = b32.ilqr_solve(iterations_max=30, tol_stationarity=2e-3)
    assert (r64["status"] == 0).all()
    assert (r32["status"] == 0).mean() > 0.9
    ok = r32["status"] == 0
    err = np.abs(b32.get("x")[ok, -1] - b64.get("x")[ok, -1]).max()
    assert err < 1e-2, err


@pytest.mark.parametrize("N", [1, 2, 3])
@pytest.mark.parametrize("name", ["pendulum", "double_integrator"])
def test_shortest_horizons_solve(name, N):
    """Horizons shorter than the kernels' record prefetch depth: same path as the oracle."""
    c = CASES[name]
    x0s = c["x0"](5)
    bt = make_hip(c["model"], N, c["n"], c["m"], c["h"], c["Qd"], c["Rd"], c["Qfd"], c["xf"], x0s, c["u0"])
    res = bt.ilqr_solve(iterations_max=25)
    x = bt.get("x")
    for b in [0, 4]:
        s = make_oracle(c["okind"], N, c["n"], c["m"], c["h"], c["Qd"], c["Rd"], c["Qfd"], c["xf"], x0s[b], c["u0"], c["dim"])
        s.L.oracle_ilqr_set_options(s.h, 25, 1e-4, 1e-4, 1e-8, 0)
        status, iters, log = s.solve()
        assert res["status"][b] == status and res["iterations"][b] == iters
        np.testing.assert_allclose(x[b], s.get("x"), rtol=1e-9, atol=1e-9)


def test_full_size_c2_pendulum_batch():
    """BASELINE.json configs[2] at full size (pendulum, N = 100, 8192 problems): every swing-up converges within the
    reference's iteration budget scale, reaches the upright goal region, and equal problems give equal bits."""
    N, batch = 100, 8192
    xf = np.array([np.pi, 0.0])
    bt = altro_amd.Batch(N, 2, 1, batch)
    bt.set_model(altro_amd.MODEL_PENDULUM, np.float32(0.03))
    bt.set_tracking_cost(np.array([[1e-2, 1e-2], [1.0, 1.0]]), np.array([[1e-3]]), np.stack([xf, xf]), np.zeros((1, 1)),
                         k_stride_zero=True, batch_stride_zero=True)
    from tests import problems
    x0 = np.zeros((batch, 2)); x0[:, 0] = problems.uniform01((batch,), 22) - 0.5
    x0[4097] = x0[3]
    bt.set_initial_state(x0)
    bt.set_input_guess(np.array([[[0.1]]]), k_stride_zero=True, batch_stride_zero=True)
    res = bt.ilqr_solve(iterations_max=30)
    assert (res["status"] == 0).all()
    assert res["iterations"].max() <= 20
    x, u = bt.get_nominal()
    assert np.abs(x[:, -1, 0] - np.pi).max() < 0.2
    assert np.array_equal(x[3], x[4097])
    # eight seeded problems of the full batch against the oracle solving each alone: status, iteration
    # count, the whole trajectory at the solve tolerance of this file (2e-7: device vs glibc sin/cos through up to 20 sweeps)
    for b in [0, 3, 1023, 4096, 4097, 5000, 7777, 8191]:
        s = make_oracle(oracle.MODEL_PENDULUM, N, 2, 1, np.float32(0.03), [1e-2, 1e-2], [1e-3], [1.0, 1.0], xf, x0[b], [0.1])
        s.L.oracle_ilqr_set_options(s.h, 30, 1e-4, 1e-4, 1e-8, 0)
        status, iters, _ = s.solve()
        assert res["status"][b] == status and res["iterations"][b] == iters, (b, res["status"][b], status, res["iterations"][b], iters)
        np.testing.assert_allclose(x[b], s.get("x"), rtol=0, atol=2e-7)
        np.testing.assert_allclose(u[b], s.get("u"), rtol=0, atol=2e-7)
