"""Pins the CPU oracle against every known-answer constant the reference's tests hold for the
hot path.  CPU only."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle
from tests import problems


def _K0(kat):
    return np.array(kat["K0_rowmajor_2x4"]).reshape(2, 4)


@pytest.mark.parametrize("float_h,scale", [(False, 1e-6), (True, 1.0)])
@pytest.mark.parametrize("is_diag", [True, False])
def test_tvlqr_kat(kats, float_h, scale, is_diag):
    """tvlqr_test.cpp:185-213 / solver_impl_test.cpp:120-148.  With h = 0.01 as a double the
    constants are exact answers (tolerance 1e-12 .. 1e-10); with the reference's `float h`
    plumbing they hold to the reference's own 1e-6 / 1e-5."""
    kat = kats["tvlqr_double_integrator"]
    pr = problems.tvlqr_kat_problem(kat, float_h)
    if is_diag:
        out = oracle.backward_batch(pr["A"], pr["B"], pr["f"], pr["Qdiag"], pr["Rdiag"], None,
                                    pr["q"], pr["r"], 0.0, True)
    else:
        out = oracle.backward_batch(pr["A"], pr["B"], pr["f"], pr["Q"], pr["R"], pr["H"],
                                    pr["q"], pr["r"], 0.0, False)
    assert out["status"][0] == -1
    n, m = pr["n"], pr["m"]
    K0 = out["K"][0, 0].reshape(n, m).T  # column-major (m x n)
    tol = kat["ref_tol"]
    assert np.linalg.norm(K0 - _K0(kat)) < tol["K"] * scale
    assert np.linalg.norm(out["d"][0, 0] - np.array(kat["d0"])) < tol["d"] * scale
    fw = oracle.forward_batch(pr["A"], pr["B"], pr["f"], out["K"], out["d"], out["P"], out["p"], pr["x0"])
    assert np.abs(fw["x"][0, -1] - np.array(kat["xN"])).max() < tol["xN"] * scale
    assert np.abs(fw["y"][0, -1] - np.array(kat["yN"])).max() < tol["yN"] * max(scale, 1e-5)


def test_total_mem_size(kats):
    """tvlqr_test.cpp:71,167: the flat-buffer size helper."""
    N, n, m = 10, 4, 2
    nx = np.full(N + 1, n, dtype=np.int32)
    nu = np.full(N, m, dtype=np.int32)
    per_k = n + m + n + n * n + n * m + n + n + n + m + m + m * n + m + n * n + n + 2 * (n * n + m * m + m * n + n + m)
    term = n + n + n + n + n * n + n + 2
    assert oracle.lib().oracle_tvlqr_TotalMemSize(nx, nu, N, True) == 8 * (N * per_k + term)


def test_cholesky_failure_returns_index():
    """tvlqr.cpp:162-164: a non-PD Quu stops the sweep and returns that knot-point index."""
    pr = problems.random_ltv(1, 6, 4, 2)
    pr["R"][0, 2] = -50.0 * np.eye(2).flatten()
    out = oracle.backward_batch(pr["A"], pr["B"], pr["f"], pr["Q"], pr["R"], pr["H"], pr["q"], pr["r"])
    assert out["status"][0] == 2
    reg = oracle.backward_batch(pr["A"], pr["B"], pr["f"], pr["Q"], pr["R"], pr["H"], pr["q"], pr["r"], reg=100.0)
    assert reg["status"][0] == -1


def test_independent_numpy_riccati():
    """Independent float64 restatement (numpy, textbook Riccati) on a random (12,4) problem:
    guards the 4x4 factorisation the reference's tests never pin."""
    pr = problems.random_ltv(2, 16, 12, 4)
    out = oracle.backward_batch(pr["A"], pr["B"], pr["f"], pr["Q"], pr["R"], pr["H"], pr["q"], pr["r"])
    N, n, m = 16, 12, 4
    for b in range(2):
        M = lambda a, k, r_, c: a[b, k].reshape(c, r_).T
        P = M(pr["Q"], N, n, n); p = pr["q"][b, N]
        for k in range(N - 1, -1, -1):
            A, B, f = M(pr["A"], k, n, n), M(pr["B"], k, n, m), pr["f"][b, k]
            Qxx = M(pr["Q"], k, n, n) + A.T @ P @ A
            Quu = M(pr["R"], k, m, m) + B.T @ P @ B
            Qux = M(pr["H"], k, m, n) + B.T @ P @ A
            t = p + P @ f
            Qx = pr["q"][b, k] + A.T @ t
            Qu = pr["r"][b, k] + B.T @ t
            K = np.linalg.solve(Quu, Qux); d = -np.linalg.solve(Quu, Qu)
            np.testing.assert_allclose(M(out["K"], k, m, n), K, rtol=0, atol=1e-10)
            np.testing.assert_allclose(out["d"][b, k], d, rtol=0, atol=1e-10)
            P = Qxx + K.T @ Quu @ K - K.T @ Qux - Qux.T @ K
            p = Qx - K.T @ Quu @ d - K.T @ Qu + Qux.T @ d
            np.testing.assert_allclose(M(out["P"], k, n, n), P, rtol=1e-11, atol=1e-10)
            np.testing.assert_allclose(out["p"][b, k], p, rtol=1e-11, atol=1e-10)


# ---------------------------------------------------------------- models
def test_double_integrator_model(kats):
    kat = kats["double_integrator_model"]
    L = oracle.lib()
    x = np.array(kat["x"]); u = np.array(kat["u"]); xn = np.zeros(4)
    L.oracle_di_dynamics(xn, x, u, kat["h"], 2)
    assert np.linalg.norm(xn - np.array(kat["xnext"])) < kat["tol"]
    J = np.zeros(24)
    L.oracle_di_jacobian(J, x, u, kat["h"], 2)
    A, B = problems.di_blocks(2, kat["h"])
    np.testing.assert_allclose(J.reshape(6, 4).T, np.hstack([A, B]), atol=1e-15)


def test_pendulum_model(kats):
    kat = kats["pendulum_model"]
    L = oracle.lib()
    mdl = oracle.make_model(oracle.MODEL_PENDULUM)
    x = np.array(kat["x"]); u = np.array(kat["u"]); xn = np.zeros(2); J = np.zeros(6)
    L.oracle_discrete_dynamics(C.byref(mdl), xn, x, u, kat["h"])
    assert np.linalg.norm(xn - np.array(kat["xnext"])) < kat["tol"]
    L.oracle_discrete_jacobian(C.byref(mdl), J, x, u, kat["h"])
    assert np.linalg.norm(J.reshape(3, 2).T - np.array(kat["J_rowmajor_2x3"]).reshape(2, 3)) < kat["tol"]
    # finite-difference check of the chain rule
    eps = 1e-6
    for j in range(3):
        xp, up = x.copy(), u.copy()
        if j < 2: xp[j] += eps
        else: up[0] += eps
        xq = np.zeros(2)
        L.oracle_discrete_dynamics(C.byref(mdl), xq, xp, up, kat["h"])
        np.testing.assert_allclose((xq - xn) / eps, J.reshape(3, 2)[j], atol=1e-5)


def test_bicycle_model(kats):
    kat = kats["bicycle_model"]
    L = oracle.lib()
    mdl = oracle.make_model(oracle.MODEL_BICYCLE)
    xd = kat["x_deg"]
    x = np.array([xd[0], xd[1], xd[2] * np.pi / 180.0, xd[3] * np.pi / 180.0]); u = np.array(kat["u"])
    xdot = np.zeros(4); J = np.zeros(24)
    L.oracle_bicycle_dynamics(C.byref(mdl.bike), xdot, x, u)
    assert np.linalg.norm(xdot - np.array(kat["xdot"])) < kat["tol_xdot"]
    L.oracle_bicycle_jacobian(C.byref(mdl.bike), J, x, u)
    assert np.linalg.norm(J.reshape(6, 4).T - np.array(kat["J_rowmajor_4x6"]).reshape(4, 6)) < kat["tol_J"]


# ---------------------------------------------------------------- merit function / solver loop
def _di_solver(kats, float_h):
    kat = kats["tvlqr_double_integrator"]
    pr = problems.tvlqr_kat_problem(kat, float_h)
    N, n, m = pr["N"], pr["n"], pr["m"]
    s = oracle.ILQR(N, n, m, kat["h"], oracle.DYN_LINEAR, cost_kind=oracle.COST_DIAGONAL)
    s.L.oracle_ilqr_set_linear_dynamics(s.h, pr["A"][0].copy(), pr["B"][0].copy(), pr["f"][0].ctypes.data_as(C.c_void_p))
    for k in range(N + 1):
        Rd = pr["Rdiag"][0, k] if k < N else None
        r = pr["r"][0, k] if k < N else None
        s.L.oracle_ilqr_set_diagonal_cost(s.h, k, pr["Qdiag"][0, k].copy(),
                                          None if Rd is None else Rd.ctypes.data_as(C.c_void_p),
                                          pr["q"][0, k].copy(),
                                          None if r is None else r.ctypes.data_as(C.c_void_p), 0.0)
    s.L.oracle_ilqr_set_initial_state(s.h, pr["x0"][0].copy())
    s.L.oracle_ilqr_initialize(s.h)
    return s, pr


@pytest.mark.parametrize("float_h,rtol", [(False, 1e-12), (True, 1e-6)])
def test_merit_function_kat(kats, float_h, rtol):
    """solver_impl_test.cpp:186-271."""
    kat = kats["merit_function_double_integrator"]
    s, pr = _di_solver(kats, float_h)
    N, n, m = pr["N"], pr["n"], pr["m"]
    x0 = pr["x0"][0]; xf = np.array(kat["xf"], dtype=float)
    for k in range(N):
        theta = k / float(N)
        s.L.oracle_ilqr_set_state(s.h, k, x0 + (xf - x0) * theta)
        s.L.oracle_ilqr_set_input(s.h, k, np.full(m, theta))
    s.L.oracle_ilqr_set_state(s.h, N, xf.copy())
    s.L.oracle_ilqr_copy_trajectory(s.h)
    s.L.oracle_ilqr_calc_cost_gradient(s.h)
    s.L.oracle_ilqr_calc_dynamics_expansions(s.h)
    s.L.oracle_ilqr_calc_expansions(s.h)
    assert s.L.oracle_ilqr_backward_pass(s.h) == -1
    phi1, dphi1 = s.merit(1.0)
    assert abs(phi1 - kat["phi_alpha1"]) / abs(kat["phi_alpha1"]) < rtol
    assert abs(dphi1 - kat["dphi_alpha1"]) / abs(kat["dphi_alpha1"]) < rtol
    eps = 1e-6
    phi_e, _ = s.merit(1.0 + eps, deriv=False)
    assert abs(dphi1 - (phi_e - phi1) / eps) / abs(dphi1) < 1e-6  # :247-255
    phi0, dphi0 = s.merit(0.0)
    assert abs(phi0 - kat["phi_alpha0"]) / abs(kat["phi_alpha0"]) < rtol
    assert abs(dphi0 - kat["dphi_alpha0"]) / abs(kat["dphi_alpha0"]) < rtol


def test_tvlqr_via_solver_and_stationarity(kats):
    """solver_impl_test.cpp:110-155: dense path through SolverImpl + stationarity < 1e-10."""
    kat = kats["tvlqr_double_integrator"]
    s, pr = _di_solver(kats, False)
    assert s.L.oracle_ilqr_backward_pass(s.h) == -1
    assert s.L.oracle_ilqr_backward_pass(s.h) == -1
    K0 = s.get("K")[0].reshape(4, 2).T
    assert np.linalg.norm(K0 - _K0(kat)) < 1e-12
    s.L.oracle_ilqr_linear_rollout(s.h)
    assert np.abs(s.get("x_cand")[-1] - np.array(kat["xN"])).max() < 1e-11
    s.L.oracle_ilqr_calc_cost_gradient(s.h)
    assert s.L.oracle_ilqr_stationarity(s.h) < kat["stationarity_lt"]


def test_forward_pass_alpha_one(kats):
    """solver_impl_test.cpp:273-316: on an LQ problem |dphi(1)| < 1e-8 and alpha == 1.0."""
    s, pr = _di_solver(kats, True)
    N, m = pr["N"], pr["m"]
    for k in range(N):
        s.L.oracle_ilqr_set_input(s.h, k, np.full(m, k / float(N)))
    s.L.oracle_ilqr_open_loop_rollout(s.h)
    s.L.oracle_ilqr_copy_trajectory(s.h)
    s.L.oracle_ilqr_calc_cost_gradient(s.h)
    s.L.oracle_ilqr_calc_dynamics_expansions(s.h)
    s.L.oracle_ilqr_calc_expansions(s.h)
    s.L.oracle_ilqr_backward_pass(s.h)
    _, dphi = s.merit(1.0)
    assert abs(dphi) < 1e-8
    err, alpha = s.forward_pass()
    assert err == 0 and alpha == 1.0


def test_pendulum_solve(kats):
    """test/pendulum_test.cpp:45-115: xN to 1e-5 in <= 10 iterations."""
    kat = kats["pendulum_solve"]
    N = kat["N"]; n, m = 2, 1
    h = np.float32(np.float32(kat["tf"]) / float(N))
    s = oracle.ILQR(N, n, m, h, oracle.DYN_MODEL, oracle.MODEL_PENDULUM, cost_kind=oracle.COST_DIAGONAL)
    xf = np.array(kat["xf_pi"]) * np.pi
    for k in range(N + 1):
        Qd = np.full(n, kat["Qfd"] if k == N else kat["Qd"])
        s.L.oracle_ilqr_set_lqr_cost(s.h, k, Qd, np.full(m, kat["Rd"]), xf.copy(), np.zeros(m))
    s.L.oracle_ilqr_set_initial_state(s.h, np.array(kat["x0"], dtype=float))
    s.L.oracle_ilqr_initialize(s.h)
    for k in range(N):
        s.L.oracle_ilqr_set_input(s.h, k, np.full(m, kat["u_init"]))
    s.L.oracle_ilqr_set_options(s.h, 200, 1e-4, 1e-4, 1e-8, 0)
    status, iters, log = s.solve()
    assert status == 0
    assert iters <= kat["max_iterations"]
    assert np.linalg.norm(s.get("x")[-1] - np.array(kat["xN"])) < kat["tol"]


def test_double_integrator_solve_unconstrained(kats):
    """test/double_integrator_test.cpp:69-168: Success within iterations_max = 3."""
    kat = kats["double_integrator_solve"]
    N = kat["N"]; dim = 2; n, m = 4, 2
    h = np.float32(np.float32(kat["tf"]) / np.float32(N))
    s = oracle.ILQR(N, n, m, h, oracle.DYN_MODEL, oracle.MODEL_DI, model_dim=dim, cost_kind=oracle.COST_DIAGONAL)
    xf = np.array(kat["xf"], dtype=float)
    for k in range(N + 1):
        s.L.oracle_ilqr_set_lqr_cost(s.h, k, np.full(n, kat["Q"]), np.full(m, kat["R"]), xf.copy(), np.zeros(m))
    x0 = np.array(kat["x0"], dtype=float)
    s.L.oracle_ilqr_set_initial_state(s.h, x0)
    s.L.oracle_ilqr_initialize(s.h)
    s.L.oracle_ilqr_set_options(s.h, kat["unconstrained_iterations_max"], 1e-4, 1e-4, 1e-8, 0)
    status, iters, log = s.solve()
    assert status == 0 and iters <= 3
    assert np.linalg.norm(s.get("x")[-1] - xf) < np.linalg.norm(x0 - xf)


def _di_constrained_oracle(kats, kind, x0=None):
    kat = kats["double_integrator_constrained"]
    c = kat[kind]
    N = kat["N"]; dim = kat["dim"]; n, m = 2 * dim, dim
    h = np.float32(np.float32(kat["tf"]) / np.float32(N))
    s = oracle.ILQR(N, n, m, h, oracle.DYN_MODEL, oracle.MODEL_DI, model_dim=dim, cost_kind=oracle.COST_DIAGONAL)
    xf = np.array(kat["xf"], dtype=float)
    for k in range(N + 1):
        s.L.oracle_ilqr_set_lqr_cost(s.h, k, np.full(n, kat["Q"]), np.full(m, kat["R"]), xf.copy(), np.zeros(m))
    x0 = np.array(c["x0"], dtype=float) if x0 is None else np.asarray(x0, dtype=float)
    s.L.oracle_ilqr_set_initial_state(s.h, x0)
    for (k0, k1, cone, G, g) in problems.di_constraint_blocks(kind, N, n, m, xf, kat["u_bnd"]):
        for k in range(k0, k1 + 1):
            s.add_linear_constraint(k, cone, G, g)
    s.L.oracle_ilqr_initialize(s.h)
    s.set_penalty(c["penalty_initial"], c["penalty_scaling"])
    return s, c, xf


@pytest.mark.parametrize("kind", ["goal", "bounds", "soc"])
def test_double_integrator_constrained_iterations(kats, kind):
    """test/double_integrator_test.cpp:255-256, 366-375, 482-492: the AL/conic restatement must reproduce the
    reference's exact iteration counts (3 / 5 / 9), goal distance < 1e-4, and saturated controls."""
    s, c, xf = _di_constrained_oracle(kats, kind)
    status, iters, log = s.solve()
    assert status == 0
    assert iters == c["iterations"], log
    assert np.linalg.norm(s.get("x")[-1] - xf) < c["goal_tol"]
    u0 = s.get("u")[0]
    if kind == "bounds":
        assert np.allclose(u0, c["u0"], atol=c["u0_tol"])
    if kind == "soc":
        assert abs(np.linalg.norm(u0) - c["u0_norm"]) < c["u0_norm_tol"]


def test_cone_projection_properties():
    """cones.cpp:13-202 restatement: projection idempotent, Jacobian == finite difference of the projection,
    Hessian == finite difference of J(x)^T b."""
    import ctypes as C
    L = oracle.lib()
    rng = np.random.default_rng(5)
    for trial in range(200):
        dim = int(rng.integers(2, 6))
        x = rng.normal(size=dim)
        if trial % 3 == 0:
            x[-1] = abs(x[-1]) * 0.3    # mostly outside the cone
        b = rng.normal(size=dim)
        px = np.zeros(dim); ppx = np.zeros(dim)
        for cone in (0, 1, 2, 3):
            L.oracle_cone_projection(cone, dim, x.ctypes.data, px.ctypes.data)
            L.oracle_cone_projection(cone, dim, px.ctypes.data, ppx.ctypes.data)
            assert np.allclose(px, ppx, atol=1e-12)
            J = np.zeros((dim, dim))
            L.oracle_cone_jacobian(cone, dim, x.ctypes.data, J.ctypes.data)
            J = J.T   # column-major
            eps = 1e-6
            Jfd = np.zeros((dim, dim))
            skip = False
            for j in range(dim):
                xp = x.copy(); xp[j] += eps; xm = x.copy(); xm[j] -= eps
                a = np.zeros(dim); c = np.zeros(dim)
                L.oracle_cone_projection(cone, dim, xp.ctypes.data, a.ctypes.data)
                L.oracle_cone_projection(cone, dim, xm.ctypes.data, c.ctypes.data)
                Jfd[:, j] = (a - c) / (2 * eps)
            if cone == 2 and np.min(np.abs(x)) < 1e-4:
                skip = True
            if cone == 3:
                a_ = np.linalg.norm(x[:-1])
                if min(abs(a_ - x[-1]), abs(a_ + x[-1])) < 1e-3:
                    skip = True
            if not skip:
                assert np.allclose(J, Jfd, atol=1e-6), (cone, x)
            if cone == 3 and not skip:
                H = np.zeros((dim, dim))
                L.oracle_cone_hessian(cone, dim, x.ctypes.data, b.ctypes.data, H.ctypes.data)
                Hfd = np.zeros((dim, dim))
                for j in range(dim):
                    xp = x.copy(); xp[j] += eps; xm = x.copy(); xm[j] -= eps
                    Jp = np.zeros((dim, dim)); Jm = np.zeros((dim, dim))
                    L.oracle_cone_jacobian(cone, dim, xp.ctypes.data, Jp.ctypes.data)
                    L.oracle_cone_jacobian(cone, dim, xm.ctypes.data, Jm.ctypes.data)
                    Hfd[:, j] = (Jp.T.T @ b - Jm.T.T @ b) / (2 * eps)   # column-major J => J^T b == Jcm @ b
                assert np.allclose(H.T, Hfd, atol=1e-5), (x, b)


def test_pendulum_goal_constrained(kats):
    """test/pendulum_test.cpp:117-203: goal constraint c = xf - x (EQUALITY) at k = N; Success, distance to goal
    < 1e-4 in <= 10 iterations."""
    kat = kats["pendulum_goal_constrained"]
    N = kat["N"]; n, m = 2, 1
    h = np.float32(np.float32(kat["tf"]) / float(N))
    s = oracle.ILQR(N, n, m, h, oracle.DYN_MODEL, oracle.MODEL_PENDULUM, cost_kind=oracle.COST_DIAGONAL)
    xf = np.array(kat["xf_pi"]) * np.pi
    for k in range(N + 1):
        Qd = np.full(n, kat["Qfd"] if k == N else kat["Qd"])
        s.L.oracle_ilqr_set_lqr_cost(s.h, k, Qd, np.full(m, kat["Rd"]), xf.copy(), np.zeros(m))
    s.L.oracle_ilqr_set_initial_state(s.h, np.array(kat["x0"], dtype=float))
    G = np.zeros((n, n + m)); G[:, :n] = -np.eye(n)
    s.add_linear_constraint(N, oracle.CONE_EQUALITY, G, -xf)
    s.L.oracle_ilqr_initialize(s.h)
    for k in range(N):
        s.L.oracle_ilqr_set_input(s.h, k, np.full(m, kat["u_init"]))
    s.L.oracle_ilqr_set_options(s.h, kat["iterations_max"], 1e-4, 1e-4, 1e-8, 0)
    status, iters, log = s.solve()
    assert status == 0
    assert iters <= kat["max_iterations"]
    assert np.linalg.norm(s.get("x")[-1] - xf) < kat["goal_tol"]


def test_cone_golden_vectors(kats):
    """src/altro/solver/test/cones_test.cpp:9-211: projection, Jacobian and Hessian constants."""
    kat = kats["cones"]
    L = oracle.lib()
    tol = kat["tol"]
    dim = 4
    x = np.array(kat["inequality_projection"]["x"])
    px = np.zeros(dim); J = np.zeros((dim, dim)); H = np.zeros((dim, dim))
    L.oracle_cone_projection(oracle.CONE_EQUALITY, dim, x.ctypes.data, px.ctypes.data)
    assert np.linalg.norm(px) < tol
    L.oracle_cone_projection(oracle.CONE_IDENTITY, dim, x.ctypes.data, px.ctypes.data)
    assert np.linalg.norm(px - x) < tol
    L.oracle_cone_projection(oracle.CONE_INEQUALITY, dim, x.ctypes.data, px.ctypes.data)
    assert np.linalg.norm(px - kat["inequality_projection"]["px"]) < tol
    L.oracle_cone_jacobian(oracle.CONE_INEQUALITY, dim, x.ctypes.data, J.ctypes.data)
    assert np.linalg.norm(J - np.diag(kat["inequality_projection"]["jac_diag"])) < tol
    L.oracle_cone_jacobian(oracle.CONE_EQUALITY, dim, x.ctypes.data, J.ctypes.data)
    assert np.linalg.norm(J) < tol
    L.oracle_cone_jacobian(oracle.CONE_IDENTITY, dim, x.ctypes.data, J.ctypes.data)
    assert np.linalg.norm(J - np.eye(dim)) < tol
    b = np.array(kat["b"], dtype=float)
    for cone in (oracle.CONE_EQUALITY, oracle.CONE_IDENTITY, oracle.CONE_INEQUALITY):
        H[:] = 1.0
        L.oracle_cone_hessian(cone, dim, x.ctypes.data, b.ctypes.data, H.ctypes.data)
        assert np.linalg.norm(H) == 0.0
    mag = np.linalg.norm(x)
    for where, scale in kat["soc_scales"].items():
        xs = x.copy(); xs[3] = mag * scale
        L.oracle_cone_projection(oracle.CONE_SOC, dim, xs.ctypes.data, px.ctypes.data)
        L.oracle_cone_jacobian(oracle.CONE_SOC, dim, xs.ctypes.data, J.ctypes.data)
        L.oracle_cone_hessian(oracle.CONE_SOC, dim, xs.ctypes.data, b.ctypes.data, H.ctypes.data)
        if where == "in":
            assert np.linalg.norm(px - xs) < tol and np.linalg.norm(J - np.eye(dim)) < tol and np.linalg.norm(H) < tol
        elif where == "below":
            assert np.linalg.norm(px) < tol and np.linalg.norm(J) < tol and np.linalg.norm(H) < tol
        else:
            assert np.linalg.norm(px - kat["soc_outside_projection"]) < tol
            assert np.linalg.norm(J.T - np.array(kat["soc_outside_jacobian"]).reshape(4, 4)) < tol
            assert np.linalg.norm(H.T - np.array(kat["soc_outside_hessian"]).reshape(4, 4)) < tol
            assert np.linalg.norm(H - H.T) < tol


def _al_knot(kat, cone, z):
    L = oracle.lib()
    L.oracle_al_knot_eval.restype = C.c_double
    L.oracle_al_knot_eval.argtypes = [C.c_int] * 4 + [C.c_void_p] * 5 + [C.c_double] + [C.c_void_p] * 7
    n, m, p = kat["n"], kat["m"], kat["p"]
    x = np.array(kat["x"], dtype=float); u = np.array(kat["u"], dtype=float)
    J = np.array(kat["J"], dtype=float)
    # the reference's constraint is nonlinear; the AL terms only see its value and Jacobian at (x, u)
    g = J @ np.concatenate([x, u]) - np.array(kat["c"], dtype=float)
    Gc = np.ascontiguousarray(J.T)
    z = np.array(z, dtype=float)
    lx = np.zeros(n); lu = np.zeros(m); lxx = np.zeros((n, n)); luu = np.zeros((m, m)); lux = np.zeros((n, m))
    hess = np.zeros((n + m, n + m)); val = np.zeros(p)
    cost = L.oracle_al_knot_eval(cone, p, n, m, Gc.ctypes.data, g.ctypes.data, x.ctypes.data, u.ctypes.data,
                                 z.ctypes.data, kat["rho"], lx.ctypes.data, lu.ctypes.data, lxx.ctypes.data,
                                 luu.ctypes.data, lux.ctypes.data, hess.ctypes.data, val.ctypes.data)
    return dict(cost=cost, lx=lx, lu=lu, lxx=lxx.T, luu=luu.T, lux=lux.T, hess=hess.T, val=val)


def test_knotpoint_al_golden_vectors(kats):
    """src/altro/solver/test/knotpoint_data_test.cpp:233-523: AL cost, gradient and Hessian constants for the
    orthant, zero and second-order cones (out of / below / inside the cone)."""
    kat = kats["knotpoint_al"]
    tol = kat["tol"]
    rho = kat["rho"]; c = np.array(kat["c"], dtype=float)
    # INEQUALITY
    k = kat["inequality"]
    r = _al_knot(kat, oracle.CONE_INEQUALITY, k["z"])
    assert np.linalg.norm(r["val"] - c) < 1e-12
    zt = np.minimum(np.array(k["z"]) - rho * c, 0.0)
    assert abs(r["cost"] - zt @ zt / (2 * rho)) < tol
    assert np.linalg.norm(r["lx"] - k["lx"]) < tol and np.linalg.norm(r["lu"] - k["lu"]) < tol
    assert np.linalg.norm(r["lxx"]) == 0.0 and np.linalg.norm(r["lux"]) == 0.0
    assert np.allclose(r["luu"], k["luu_const"], rtol=1e-12)
    # EQUALITY
    k = kat["equality"]
    r = _al_knot(kat, oracle.CONE_EQUALITY, k["z"])
    zt = np.array(k["z"]) - rho * c
    assert abs(r["cost"] - zt @ zt / (2 * rho)) < tol
    assert np.linalg.norm(r["lx"] - k["lx"]) < tol and np.linalg.norm(r["lu"] - k["lu"]) < tol
    assert np.linalg.norm(r["lxx"] - np.array(k["lxx"]).reshape(3, 3)) < 1e-13
    assert np.linalg.norm(r["lux"]) == 0.0 and np.allclose(r["luu"], k["luu_const"], rtol=1e-12)
    # SOC out of cone
    k = kat["soc_out_of_cone"]
    r = _al_knot(kat, oracle.CONE_SOC, k["z"])
    assert abs(r["cost"] - k["alcost"]) < tol
    assert np.linalg.norm(r["lx"] - k["lx"]) < tol and np.linalg.norm(r["lu"] - k["lu"]) < tol
    assert np.linalg.norm(r["hess"] - np.array(k["hess"]).reshape(5, 5)) < k["hess_tol"]
    # SOC below the cone
    k = kat["soc_below_cone"]
    r = _al_knot(kat, oracle.CONE_SOC, k["z"])
    zb = np.array(k["z"]) - rho * c
    assert np.linalg.norm(zb[:2]) < -zb[2]
    assert abs(r["cost"] - k["alcost"]) < tol
    assert np.linalg.norm(r["lx"]) < tol and np.linalg.norm(r["lu"]) < tol and np.linalg.norm(r["hess"]) < 1e-6
    # SOC in the cone
    k = kat["soc_in_cone"]
    r = _al_knot(kat, oracle.CONE_SOC, k["z"])
    assert abs(r["cost"] - k["alcost"]) < tol
    assert np.linalg.norm(r["lx"] - k["lx"]) < tol and np.linalg.norm(r["lu"] - k["lu"]) < tol
    assert np.linalg.norm(r["lxx"] - np.array(k["lxx"]).reshape(3, 3)) < 1e-13
    assert np.linalg.norm(r["lux"]) == 0.0 and np.allclose(r["luu"], k["luu_const"], rtol=1e-12)


def test_pendulum_alilqr_hand_sequenced(kats):
    """solver/test/alilqr_test.cpp:112-215: the reference's own AL-iLQR test drives SolverImpl by hand -- phi(0) of
    the open-loop rollout (AL terms included) = 10.632455092693577, six iterations with the line search at
    (c1, c2) = (1e-4, 0.1) leave the pendulum 0.04186387 from the goal, a dual + penalty update and six more shrink
    that more than five-fold, a dual + two penalty updates and six more bring it under 1e-4."""
    kat = kats["pendulum_alilqr_hand_sequenced"]
    N = kat["N"]; n, m = 2, 1
    h = np.float32(np.float32(kat["tf"]) / float(N))
    s = oracle.ILQR(N, n, m, h, oracle.DYN_MODEL, oracle.MODEL_PENDULUM, cost_kind=oracle.COST_DIAGONAL)
    L = s.L
    xf = np.array(kat["xf_pi"]) * np.pi
    for k in range(N + 1):
        Qd = np.full(n, kat["Qfd"] if k == N else kat["Qd"])
        L.oracle_ilqr_set_lqr_cost(s.h, k, Qd, np.full(m, kat["Rd"]), xf.copy(), np.zeros(m))
    L.oracle_ilqr_set_initial_state(s.h, np.array(kat["x0"], dtype=float))
    G = np.zeros((n, n + m)); G[:, :n] = -np.eye(n)
    s.add_linear_constraint(N, oracle.CONE_EQUALITY, G, -xf)
    L.oracle_ilqr_initialize(s.h)
    for k in range(N):
        L.oracle_ilqr_set_input(s.h, k, np.full(m, kat["u_init"]))
    L.oracle_ilqr_open_loop_rollout(s.h)
    L.oracle_ilqr_copy_trajectory(s.h)
    phi0, _ = s.merit(0.0, deriv=False)
    assert abs(phi0 - kat["phi0"]) < kat["phi0_tol"]

    L.oracle_ilqr_set_linesearch_tolerances(s.h, kat["ls_c1"], kat["ls_c2"])

    def block():
        L.oracle_ilqr_refresh_expansions(s.h)
        dist = None
        for _ in range(kat["iterations_per_block"]):
            L.oracle_ilqr_calc_expansions(s.h)
            L.oracle_ilqr_backward_pass(s.h)
            err, alpha = s.forward_pass()
            if err == 1:      # MeritFunctionGradientTooSmall: the reference's later blocks break here
                break
            assert err == 0
            dist = np.linalg.norm(s.get("x_cand")[-1] - xf)
            L.oracle_ilqr_copy_trajectory(s.h)
        return dist if dist is not None else np.linalg.norm(s.get("x")[-1] - xf)

    d1 = block()
    assert abs(d1 - kat["dist_after_block1"]) < kat["dist_after_block1_tol"], d1
    L.oracle_ilqr_dual_update(s.h)
    L.oracle_ilqr_penalty_update(s.h)
    d2 = block()
    assert d2 < d1 / kat["block2_ratio"], (d1, d2)
    L.oracle_ilqr_dual_update(s.h)
    L.oracle_ilqr_penalty_update(s.h)
    L.oracle_ilqr_penalty_update(s.h)
    d3 = block()
    assert d3 < kat["final_dist_tol"], d3


def test_bicycle_turn90(kats):
    """test/bicycle_test.cpp:53-138: the unconstrained 90-degree turn with the backtracking line search ends within
    1e-2 of the goal."""
    kat = kats["bicycle_turn90"]
    N = kat["N"]; n, m = 4, 2
    h = np.float32(np.float32(kat["tf"]) / float(N))
    s = oracle.ILQR(N, n, m, h, oracle.DYN_MODEL, oracle.MODEL_BICYCLE, cost_kind=oracle.COST_DIAGONAL)
    xf = np.array([1.0, 2.0, np.pi / 2, 0.0])
    for k in range(N + 1):
        Qd = np.full(n, kat["Qfd"] if k == N else kat["Qd"])
        s.L.oracle_ilqr_set_lqr_cost(s.h, k, Qd, np.full(m, kat["Rd"]), xf.copy(), np.zeros(m))
    s.L.oracle_ilqr_set_initial_state(s.h, np.array(kat["x0"], dtype=float))
    s.L.oracle_ilqr_initialize(s.h)
    for k in range(N):
        s.L.oracle_ilqr_set_input(s.h, k, np.array(kat["u_init"], dtype=float))
    s.L.oracle_ilqr_set_options(s.h, kat["iterations_max"], 1e-4, 1e-4, 1e-8, kat["use_backtracking"])
    status, iters, log = s.solve()
    assert np.linalg.norm(s.get("x")[-1] - xf) < kat["goal_tol"]


def test_quadrotor_test_model_jacobian_by_central_differences():
    """The 12-state quadrotor (oracle/models_oracle.c) is this repo's own test model for NONLINEAR dynamics at the (12, 4)
    shape -- the reference ships none that large -- so nothing of the reference pins it: its analytic Jacobian is checked here
    against central differences of its own dynamics, continuous and discretised (midpoint rule + chain rule, the reference's
    test_utils.cpp:84-132), and a hover input keeps the hovering state a fixed point."""
    import ctypes as C
    L = oracle.lib()
    L.oracle_quadrotor_dynamics.argtypes = [C.c_void_p] * 3
    L.oracle_quadrotor_jacobian.argtypes = [C.c_void_p] * 3
    rng = np.random.default_rng(7)
    mdl = oracle.make_model(oracle.MODEL_QUADROTOR)
    h = np.float32(0.02)
    for trial in range(5):
        x = rng.normal(size=12) * 0.4
        u = np.array([0.5 * 9.81, 0.0, 0.0, 0.0]) + rng.normal(size=4) * np.array([1.0, 0.01, 0.01, 0.01])

        def f(z):
            out = np.zeros(12)
            xx, uu = np.ascontiguousarray(z[:12]), np.ascontiguousarray(z[12:])
            L.oracle_quadrotor_dynamics(out.ctypes.data, xx.ctypes.data, uu.ctypes.data)
            return out

        def F(z):
            out = np.zeros(12)
            L.oracle_discrete_dynamics(C.byref(mdl), out, np.ascontiguousarray(z[:12]), np.ascontiguousarray(z[12:]), h)
            return out
        z = np.concatenate([x, u])
        J = np.zeros(12 * 16)
        L.oracle_quadrotor_jacobian(J.ctypes.data, x.ctypes.data, u.ctypes.data)
        Jd = np.zeros(12 * 16)
        L.oracle_discrete_jacobian(C.byref(mdl), Jd, x, u, h)
        for (fun, Jan) in ((f, J.reshape(16, 12).T), (F, Jd.reshape(16, 12).T)):
            num = np.zeros((12, 16))
            for c in range(16):
                e = np.zeros(16); e[c] = 1e-6 * max(1.0, abs(z[c]))
                num[:, c] = (fun(z + e) - fun(z - e)) / (2 * e[c])
            assert np.abs(num - Jan).max() <= 2e-7 * max(1.0, np.abs(Jan).max()), (trial, np.abs(num - Jan).max())
    hover_x = np.zeros(12); hover_x[:3] = [1.0, -2.0, 3.0]
    hover_u = np.array([0.5 * 9.81, 0, 0, 0.0])
    out = np.zeros(12)
    L.oracle_discrete_dynamics(C.byref(mdl), out, hover_x, hover_u, h)
    assert np.abs(out - hover_x).max() < 1e-15


def test_quaternion_quadrotor_test_model_jacobian_by_central_differences():
    """The 13-state quaternion quadrotor (oracle/models_oracle.c, round 6): the state dimension one past the (12, 4) tile.  Like its
    12-state sibling nothing of the reference pins it, so its analytic Jacobian is held to central differences of its own dynamics,
    continuous and discretised (midpoint rule + chain rule, test_utils.cpp:84-132); hover with the identity attitude is a fixed
    point, and the quaternion's norm changes only at second order in the step (qdot is orthogonal to q)."""
    import ctypes as C
    L = oracle.lib()
    L.oracle_quadrotor13_dynamics.argtypes = [C.c_void_p] * 3
    L.oracle_quadrotor13_jacobian.argtypes = [C.c_void_p] * 3
    rng = np.random.default_rng(13)
    mdl = oracle.make_model(oracle.MODEL_QUADROTOR13)
    h = np.float32(0.02)
    n, w = 13, 17
    for trial in range(5):
        x = rng.normal(size=n) * 0.4
        x[3:7] = np.array([1.0, 0.0, 0.0, 0.0]) + 0.3 * rng.normal(size=4)
        x[3:7] /= np.linalg.norm(x[3:7])
        u = np.array([0.5 * 9.81, 0.0, 0.0, 0.0]) + rng.normal(size=4) * np.array([1.0, 0.01, 0.01, 0.01])

        def f(z):
            out = np.zeros(n)
            xx, uu = np.ascontiguousarray(z[:n]), np.ascontiguousarray(z[n:])
            L.oracle_quadrotor13_dynamics(out.ctypes.data, xx.ctypes.data, uu.ctypes.data)
            return out

        def F(z):
            out = np.zeros(n)
            L.oracle_discrete_dynamics(C.byref(mdl), out, np.ascontiguousarray(z[:n]), np.ascontiguousarray(z[n:]), h)
            return out
        z = np.concatenate([x, u])
        J = np.zeros(n * w)
        L.oracle_quadrotor13_jacobian(J.ctypes.data, x.ctypes.data, u.ctypes.data)
        Jd = np.zeros(n * w)
        L.oracle_discrete_jacobian(C.byref(mdl), Jd, x, u, h)
        for (fun, Jan) in ((f, J.reshape(w, n).T), (F, Jd.reshape(w, n).T)):
            num = np.zeros((n, w))
            for c in range(w):
                e = np.zeros(w); e[c] = 1e-6 * max(1.0, abs(z[c]))
                num[:, c] = (fun(z + e) - fun(z - e)) / (2 * e[c])
            assert np.abs(num - Jan).max() <= 2e-7 * max(1.0, np.abs(Jan).max()), (trial, np.abs(num - Jan).max())
        assert abs(np.dot(f(z)[3:7], x[3:7])) < 1e-15                       # qdot is orthogonal to q
        assert abs(np.linalg.norm(F(z)[3:7]) - 1.0) < 5e-3
    hover_x = np.zeros(n); hover_x[:3] = [1.0, -2.0, 3.0]; hover_x[3] = 1.0
    hover_u = np.array([0.5 * 9.81, 0, 0, 0.0])
    out = np.zeros(n)
    L.oracle_discrete_dynamics(C.byref(mdl), out, hover_x, hover_u, h)
    assert np.abs(out - hover_x).max() < 1e-15
