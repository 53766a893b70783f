"""The iLQR loop with PER-KNOT-POINT dimensions on the batched ABI (missing #4): the reference sets dimensions per index
range (ALTROSolver::SetDimension(n, m, k_start, k_stop), altro_solver.cpp:26-47; KnotPointData sizes every block from its own n, m,
knotpoint_data.cpp:64-121) and solves.  altro_hip_batch_create_dims + set_dynamics / set_quadratic_cost / set_initial_state /
set_input_guess / add_linear_constraint / ilqr_solve on plan GENERIC, arrays packed [batch][k][block_k].

The checker: the oracle's restatement of SolverImpl has ONE (n, m), so it solves the EQUIVALENT zero-padded problem -- every knot
point padded to (max nx, max nu) with states that stay at zero (zero rows / columns in A, B, f, Q, H, q) and inputs that cost
1/2 u^2 and move nothing (identity block in R, zero columns in B, H): same cost, same gains on the real entries, same merit
function, hence the same line-search decisions; the padded entries of its iterates are zero.  Tolerances: merit phi 1e-11, phi'
1e-9, candidate trajectories 1e-10 (sums reduced over the wave on the device), whole solves: same status and iterations,
trajectories 1e-8."""
import os
import tempfile

import numpy as np
import pytest

import altro_amd
from oracle import oracle
from tests import cpp_build
from tests import problems

pytestmark = pytest.mark.gpu

NX = np.array([5, 5, 5, 7, 7, 7, 7, 4, 4, 4, 4, 4, 4])
NU = np.array([2, 2, 3, 3, 3, 3, 1, 1, 2, 2, 2, 2])
N = len(NU)
NMAX, MMAX = int(NX.max()), int(NU.max())


def make_problem(batch, seed=401):
    """Random LTV dynamics between spaces of changing dimension, a dense cost; blocks as lists over k of [batch, rows, cols]."""
    p = dict(A=[], B=[], f=[], Q=[], R=[], H=[], q=[], r=[])
    for k in range(N):
        n, m, n2 = NX[k], NU[k], NX[k + 1]
        A = 0.25 * problems.normal((batch, n2, n), seed + 7 * k)
        for i in range(min(n, n2)):
            A[:, i, i] += 0.9
        p["A"].append(A)
        p["B"].append(0.4 * problems.normal((batch, n2, m), seed + 7 * k + 1))
        p["f"].append(0.05 * problems.normal((batch, n2), seed + 7 * k + 2))
        M = problems.normal((batch, m, m), seed + 7 * k + 3)
        p["R"].append(0.1 * np.eye(m) + 0.02 * M @ np.swapaxes(M, -1, -2))
        p["H"].append(0.03 * problems.normal((batch, m, n), seed + 7 * k + 4))
        p["r"].append(0.3 * problems.normal((batch, m), seed + 7 * k + 5))
    for k in range(N + 1):
        n = NX[k]
        L = problems.normal((batch, n, n), seed + 1000 + 3 * k)
        p["Q"].append(np.eye(n) + 0.1 * L @ np.swapaxes(L, -1, -2))
        p["q"].append(0.3 * problems.normal((batch, n), seed + 1001 + 3 * k))
    p["c"] = problems.uniform01((batch, N + 1), seed + 5000)
    p["x0"] = problems.normal((batch, NX[0]), seed + 5001)
    p["u0"] = [0.1 * problems.normal((batch, NU[k]), seed + 6000 + k) for k in range(N)]
    return p


def packed(blocks):
    """[batch][k][block_k column-major] as one flat array per problem."""
    cols = [np.ascontiguousarray(np.swapaxes(b, -1, -2)).reshape(b.shape[0], -1) if b.ndim == 3 else b for b in blocks]
    return np.ascontiguousarray(np.concatenate(cols, axis=1))


def make_hip(p, batch, bounds):
    bt = altro_amd.Batch.with_dims(NX, NU, batch)
    assert bt.plan == altro_amd.PLAN_GENERIC
    bt.set_dynamics(packed(p["A"]), packed(p["B"]), packed(p["f"]))
    bt.set_quadratic_cost(packed(p["Q"]), packed(p["R"]), packed(p["H"]), packed(p["q"]), packed(p["r"]), p["c"])
    bt.set_initial_state(p["x0"])
    bt.set_input_guess(packed(p["u0"]))
    for (k0, k1, G, g) in bounds:
        bt.add_linear_constraint(k0, k1, altro_amd.CONE_EQUALITY if k0 == N else altro_amd.CONE_INEQUALITY, G, g)
    return bt


def make_oracle(p, b, bounds):
    s = oracle.ILQR(N, NMAX, MMAX, 0.01, oracle.DYN_LINEAR, cost_kind=oracle.COST_QUADRATIC)
    A = np.zeros((N, NMAX * NMAX)); B = np.zeros((N, NMAX * MMAX)); f = np.zeros((N, NMAX))
    for k in range(N):
        n, m, n2 = NX[k], NU[k], NX[k + 1]
        Ak = np.zeros((NMAX, NMAX)); Ak[:n2, :n] = p["A"][k][b]
        Bk = np.zeros((NMAX, MMAX)); Bk[:n2, :m] = p["B"][k][b]
        A[k] = Ak.T.reshape(-1); B[k] = Bk.T.reshape(-1); f[k, :n2] = p["f"][k][b]
    fc = np.ascontiguousarray(f)
    s.L.oracle_ilqr_set_linear_dynamics(s.h, np.ascontiguousarray(A), np.ascontiguousarray(B), fc.ctypes.data)
    for k in range(N + 1):
        n = NX[k]
        Q = np.zeros((NMAX, NMAX)); Q[:n, :n] = p["Q"][k][b]
        q = np.zeros(NMAX); q[:n] = p["q"][k][b]
        R = np.eye(MMAX); H = np.zeros((MMAX, NMAX)); r = np.zeros(MMAX)
        if k < N:
            m = NU[k]
            R[:m, :m] = p["R"][k][b]; H[:m, :n] = p["H"][k][b]; r[:m] = p["r"][k][b]
        Rc, Hc, rc_ = np.ascontiguousarray(R.T.reshape(-1)), np.ascontiguousarray(H.T.reshape(-1)), np.ascontiguousarray(r)   # (kept alive over the call)
        s.L.oracle_ilqr_set_quadratic_cost(s.h, k, np.ascontiguousarray(Q.T.reshape(-1)), Rc.ctypes.data, Hc.ctypes.data, np.ascontiguousarray(q),
                                           rc_.ctypes.data, float(p["c"][b, k]))
    x0 = np.zeros(NMAX); x0[:NX[0]] = p["x0"][b]
    s.L.oracle_ilqr_set_initial_state(s.h, np.ascontiguousarray(x0))
    for (k0, k1, G, g) in bounds:
        for k in range(k0, k1 + 1):
            n, m = NX[k], (NU[k] if k < N else 0)
            Gp = np.zeros((G.shape[0], NMAX + MMAX)); Gp[:, :n] = G[:, :n]; Gp[:, NMAX:NMAX + m] = G[:, n:n + m]
            s.add_linear_constraint(k, oracle.CONE_EQUALITY if k == N else oracle.CONE_INEQUALITY, Gp, g)
    s.L.oracle_ilqr_initialize(s.h)
    for k in range(N):
        u = np.zeros(MMAX); u[:NU[k]] = p["u0"][k][b]
        s.L.oracle_ilqr_set_input(s.h, k, np.ascontiguousarray(u))
    return s


def unpad_x(xp):
    return np.concatenate([xp.reshape(N + 1, NMAX)[k, :NX[k]] for k in range(N + 1)])


def unpad_u(up):
    return np.concatenate([up.reshape(N, MMAX)[k, :NU[k]] for k in range(N)])


def input_bounds():
    """|u| <= 0.25 on the knot points whose input has 3 entries (k = 2..5), u_0 <= 0.1 where it has one (k = 6, 7)."""
    out = []
    n, m = 7, 3   # k = 3..5; k = 2 has n = 5
    for (k0, k1) in ((2, 2), (3, 5)):
        n = NX[k0]
        G = np.zeros((2 * m, n + m)); g = np.full(2 * m, 0.25)
        for i in range(m):
            G[i, n + i] = 1.0; G[m + i, n + i] = -1.0
        out.append((k0, k1, G, g))
    G = np.zeros((1, 7 + 1)); G[0, 7] = 1.0
    out.append((6, 6, G, np.array([0.1])))
    G = np.zeros((1, 4 + 1)); G[0, 4] = 1.0
    out.append((7, 7, G, np.array([0.1])))
    return out


def terminal_pin():
    """x_N[0] = 0.3, x_N[2] = -0.2: an EQUALITY block of the terminal knot point alone (G is p x nx[N])."""
    G = np.zeros((2, NX[N])); G[0, 0] = 1.0; G[1, 2] = 1.0
    return (N, N, G, np.array([0.3, -0.2]))


def test_merit_and_stationarity_with_varying_dimensions():
    batch = 5
    p = make_problem(batch)
    bt = make_hip(p, batch, [])
    bt.open_loop_rollout(); bt.accept(); bt.expand(); bt.backward()
    assert (bt.get("status") == -1).all()
    alphas = np.linspace(0.2, 1.1, batch)
    phi, dphi = bt.merit(alphas)
    xc, uc = bt.get("x"), bt.get("u")
    st = bt.stationarity()
    for b in (0, 2, 4):
        s = make_oracle(p, b, [])
        s.L.oracle_ilqr_open_loop_rollout(s.h); s.L.oracle_ilqr_copy_trajectory(s.h)
        s.L.oracle_ilqr_calc_dynamics_expansions(s.h); s.L.oracle_ilqr_calc_cost_gradient(s.h)
        s.L.oracle_ilqr_calc_expansions(s.h)
        assert s.L.oracle_ilqr_backward_pass(s.h) == -1
        p_ref, dp_ref = s.merit(alphas[b])
        assert abs(phi[b] - p_ref) <= 1e-11 * max(1.0, abs(p_ref)), (b, phi[b], p_ref)
        assert abs(dphi[b] - dp_ref) <= 1e-9 * max(1.0, abs(dp_ref)), (b, dphi[b], dp_ref)
        np.testing.assert_allclose(xc[b], unpad_x(s.get("x_cand")), rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(uc[b], unpad_u(s.get("u_cand")), rtol=1e-10, atol=1e-10)
        # the padded iterates stay at zero: the embedding is exact
        xp = s.get("x_cand").reshape(N + 1, NMAX)
        assert all(np.all(xp[k, NX[k]:] == 0.0) for k in range(N + 1))
        ref_st = s.L.oracle_ilqr_stationarity(s.h)
        assert abs(st[b] - ref_st) <= 1e-8 * max(1.0, ref_st)


@pytest.mark.parametrize("constrained", [False, True])
def test_whole_solves_with_varying_dimensions(constrained):
    batch = 9
    p = make_problem(batch, seed=977)
    bounds = input_bounds() + [terminal_pin()] if constrained else []
    bt = make_hip(p, batch, bounds)
    res = bt.ilqr_solve(iterations_max=60, tol_stationarity=1e-4, penalty_initial=1.0, penalty_scaling=10.0)
    assert (res["status"] == 0).all(), res["status"]
    x, u = bt.get_nominal()
    x3, u3 = bt.get_knot(3)
    assert x3.shape == (batch, 7) and u3.shape == (batch, 3)
    if constrained:
        off_u = np.concatenate([[0], np.cumsum(NU)])
        for k in range(2, 6):
            assert np.abs(u[:, off_u[k]:off_u[k + 1]]).max() <= 0.25 + 1e-4
        assert (res["iterations"] > 2).all()          # the bounds bind: more than the LQ problem's sweeps
        off_x = np.concatenate([[0], np.cumsum(NX)])
        assert np.abs(x[:, off_x[N] + 0] - 0.3).max() < 1e-4 and np.abs(x[:, off_x[N] + 2] + 0.2).max() < 1e-4
    else:
        assert (res["iterations"] <= 3).all()
    for b in (0, 4, 8):
        s = make_oracle(p, b, bounds)
        if constrained:
            s.set_penalty(1.0, 10.0)
        s.L.oracle_ilqr_set_options(s.h, 60, 1e-4, 1e-4, 1e-8, 0)
        status, iters, log = s.solve()
        assert status == 0 and iters == res["iterations"][b], (b, status, iters, res["iterations"][b])
        np.testing.assert_allclose(x[b], unpad_x(s.get("x")), rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(u[b], unpad_u(s.get("u")), rtol=1e-7, atol=1e-7)
        off_x = np.concatenate([[0], np.cumsum(NX)])
        np.testing.assert_allclose(x3[b], x[b, off_x[3]:off_x[4]], rtol=0, atol=0)


@pytest.mark.parametrize("constrained", [False, True])
def test_cpp_altro_solver_with_varying_dimensions(constrained):
    """The same problem through the C++ ALTROSolver (include/altro/altro.hpp): SetDimension / SetExplicitDynamics / SetQuadraticCost per
    knot point, host callbacks, every backward sweep through tvlqr_BackwardPass with per-knot-point nx, nu on the GPU.  Against the
    batched ABI on the same problem (1e-9: both run plan GENERIC's sweep, the host loop sums in index order, the device loop over the
    wave) and the oracle on the padded one."""
    batch = 3
    p = make_problem(batch, seed=977)
    b = 1
    col = lambda M: " ".join(repr(float(v)) for v in np.asarray(M).T.reshape(-1))     # column-major
    vec = lambda v: " ".join(repr(float(x)) for x in np.asarray(v).reshape(-1))
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "problem.txt")
        with open(path, "w") as f:
            f.write("%d\n%s\n%s\n" % (N, " ".join(map(str, NX)), " ".join(map(str, NU))))
            for k in range(N):
                f.write("\n".join([col(p["A"][k][b]), col(p["B"][k][b]), vec(p["f"][k][b]), col(p["R"][k][b]), col(p["H"][k][b]),
                                   vec(p["r"][k][b])]) + "\n")
            for k in range(N + 1):
                f.write("%s\n%s\n%r\n" % (col(p["Q"][k][b]), vec(p["q"][k][b]), float(p["c"][b, k])))
            f.write(vec(p["x0"][b]) + "\n")
            for k in range(N):
                f.write(vec(p["u0"][k][b]) + "\n")
        rc, out, err = cpp_build.run("altro_varying_dims_test", args=[path] + (["constrained"] if constrained else []), timeout=300)
    assert rc == 0 and out.strip().endswith("OK"), out[-1500:] + err[-1500:]
    lines = out.splitlines()
    head = [l for l in lines if l.startswith("status ")][0].split()
    status, iters = int(head[1]), int(head[3])
    xs = np.array([float(v) for v in [l for l in lines if l.startswith("x ")][0].split()[1:]])
    us = np.array([float(v) for v in [l for l in lines if l.startswith("u ")][0].split()[1:]])
    bounds = input_bounds() + [terminal_pin()] if constrained else []
    bt = make_hip(p, batch, bounds)
    res = bt.ilqr_solve(iterations_max=60, tol_stationarity=1e-4, penalty_initial=1.0, penalty_scaling=10.0)
    x, u = bt.get_nominal()
    assert status == 0 and res["status"][b] == 0 and iters == res["iterations"][b], (status, iters, res["status"][b], res["iterations"][b])
    tol = 1e-7 if constrained else 1e-9
    np.testing.assert_allclose(xs, x[b], rtol=tol, atol=tol)
    np.testing.assert_allclose(us, u[b], rtol=10 * tol, atol=10 * tol)
    s = make_oracle(p, b, bounds)
    if constrained:
        s.set_penalty(1.0, 10.0)
    s.L.oracle_ilqr_set_options(s.h, 60, 1e-4, 1e-4, 1e-8, 0)
    ostatus, oiters, _ = s.solve()
    assert ostatus == 0 and oiters == iters
    np.testing.assert_allclose(xs, unpad_x(s.get("x")), rtol=1e-8, atol=1e-8)


@pytest.mark.parametrize("seed,lo,hi,mhi,dtype", [(1, 1, 9, 4, altro_amd.F64), (2, 2, 12, 6, altro_amd.F64), (3, 20, 40, 35, altro_amd.F64),
                                                  (4, 3, 10, 5, altro_amd.F32)])
def test_random_dimension_sequences(seed, lo, hi, mhi, dtype, monkeypatch):
    """Seeded random nx[k] in [lo, hi], nu[k] in [1, mhi] (case 3: past 32, where a lane owns a state row and an input row):
    merit, candidates and whole LQ solves against the padded oracle."""
    import sys
    me = sys.modules[__name__]
    rs = np.random.RandomState(seed)
    Nn = 9
    nx = rs.randint(lo, hi + 1, size=Nn + 1); nu = rs.randint(1, mhi + 1, size=Nn)
    for name, val in (("NX", nx), ("NU", nu), ("N", Nn), ("NMAX", int(nx.max())), ("MMAX", int(nu.max()))):
        monkeypatch.setattr(me, name, val)
    batch = 3
    p = make_problem(batch, seed=1000 + seed)
    f32 = dtype == altro_amd.F32
    bt = altro_amd.Batch.with_dims(NX, NU, batch, dtype=dtype)
    bt.set_dynamics(packed(p["A"]), packed(p["B"]), packed(p["f"]))
    bt.set_quadratic_cost(packed(p["Q"]), packed(p["R"]), packed(p["H"]), packed(p["q"]), packed(p["r"]), p["c"])
    bt.set_initial_state(p["x0"]); bt.set_input_guess(packed(p["u0"]))
    bt.open_loop_rollout(); bt.accept(); bt.expand(); bt.backward()
    assert (bt.get("status") == -1).all()
    phi, dphi = bt.merit(np.full(batch, 0.7))
    xc = bt.get("x")
    res = bt.ilqr_solve(iterations_max=10, tol_stationarity=1e-2 if f32 else 1e-4)
    assert (res["status"] == 0).all() and (res["iterations"] <= 3).all()
    x, u = bt.get_nominal()
    for b in range(batch):
        s = make_oracle(p, b, [])
        s.L.oracle_ilqr_open_loop_rollout(s.h); s.L.oracle_ilqr_copy_trajectory(s.h)
        s.L.oracle_ilqr_calc_dynamics_expansions(s.h); s.L.oracle_ilqr_calc_cost_gradient(s.h)
        s.L.oracle_ilqr_calc_expansions(s.h)
        assert s.L.oracle_ilqr_backward_pass(s.h) == -1
        p_ref, dp_ref = s.merit(0.7)
        assert abs(phi[b] - p_ref) <= (2e-4 if f32 else 1e-11) * max(1.0, abs(p_ref)), (b, phi[b], p_ref)
        assert abs(dphi[b] - dp_ref) <= (2e-3 if f32 else 1e-9) * max(1.0, abs(dp_ref)), (b, dphi[b], dp_ref)
        np.testing.assert_allclose(xc[b], unpad_x(s.get("x_cand")), rtol=2e-4 if f32 else 1e-10, atol=2e-4 if f32 else 1e-10)
        s2 = make_oracle(p, b, [])
        s2.L.oracle_ilqr_set_options(s2.h, 10, 1e-4, 1e-4, 1e-8, 0)
        status, iters, _ = s2.solve()
        assert status == 0 and (f32 or iters == res["iterations"][b])
        tol = 5e-4 if f32 else 1e-8
        np.testing.assert_allclose(x[b], unpad_x(s2.get("x")), rtol=tol, atol=tol)
        np.testing.assert_allclose(u[b], unpad_u(s2.get("u")), rtol=10 * tol, atol=10 * tol)


def test_tracking_cost_and_linear_cost_update_with_varying_dimensions():
    """altro_hip_set_tracking_cost (ALTROSolver::SetLQRCost, altro_solver.cpp:138-172) and altro_hip_update_linear_costs
    (UpdateLinearCosts, altro_solver.cpp:266-281) with packed per-knot-point arrays: the same iterates as the dense blocks they stand for."""
    batch = 4
    p = make_problem(batch, seed=55)
    Qd = [1.0 + problems.uniform01((batch, NX[k]), 700 + k) for k in range(N + 1)]
    Rd = [0.1 + 0.2 * problems.uniform01((batch, NU[k]), 800 + k) for k in range(N)]
    xref = [0.5 * problems.normal((batch, NX[k]), 900 + k) for k in range(N + 1)]
    uref = [0.2 * problems.normal((batch, NU[k]), 950 + k) for k in range(N)]

    def build(tracking, q_override=None):
        bt = altro_amd.Batch.with_dims(NX, NU, batch)
        bt.set_dynamics(packed(p["A"]), packed(p["B"]), packed(p["f"]))
        if tracking:
            bt.set_tracking_cost(packed(Qd), packed(Rd), packed(xref), packed(uref))
        else:
            Q = [np.stack([np.diag(Qd[k][b]) for b in range(batch)]) for k in range(N + 1)]
            R = [np.stack([np.diag(Rd[k][b]) for b in range(batch)]) for k in range(N)]
            H = [np.zeros((batch, NU[k], NX[k])) for k in range(N)]
            q = [-(Qd[k] * xref[k]) for k in range(N + 1)] if q_override is None else q_override
            r = [-(Rd[k] * uref[k]) for k in range(N)]
            c = np.stack([0.5 * (xref[k] * Qd[k] * xref[k]).sum(1) + (0.5 * (uref[k] * Rd[k] * uref[k]).sum(1) if k < N else 0.0)
                          for k in range(N + 1)], axis=1)
            bt.set_quadratic_cost(packed(Q), packed(R), packed(H), packed(q), packed(r), c)
        bt.set_initial_state(p["x0"]); bt.set_input_guess(packed(p["u0"]))
        return bt
    bt, bd = build(True), build(False)
    r1, r2 = bt.ilqr_solve(iterations_max=20, tol_stationarity=1e-6), bd.ilqr_solve(iterations_max=20, tol_stationarity=1e-6)
    assert (r1["status"] == 0).all() and np.array_equal(r1["iterations"], r2["iterations"])
    (x1, u1), (x2, u2) = bt.get_nominal(), bd.get_nominal()
    np.testing.assert_allclose(x1, x2, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(u1, u2, rtol=1e-12, atol=1e-12)
    # new linear state costs on knot points 3..7 (7 and 4 states): the packed range
    qn = [-(Qd[k] * (xref[k] + 0.3)) for k in range(N + 1)]
    bt.update_linear_costs(packed(qn[3:8]), None, None, 3, 7)
    bt.set_input_guess(packed(p["u0"]))
    r3 = bt.ilqr_solve(iterations_max=20, tol_stationarity=1e-6)
    q_all = [-(Qd[k] * xref[k]) for k in range(N + 1)]
    q_all[3:8] = qn[3:8]
    be = build(False, q_all)
    r4 = be.ilqr_solve(iterations_max=20, tol_stationarity=1e-6)
    assert (r3["status"] == 0).all() and np.array_equal(r3["iterations"], r4["iterations"])
    np.testing.assert_allclose(bt.get_nominal()[0], be.get_nominal()[0], rtol=1e-12, atol=1e-12)
    assert np.abs(bt.get_nominal()[0] - x1).max() > 1e-3       # (the update moved the solution)


def test_calls_that_need_one_dimension_say_so():
    bt = altro_amd.Batch.with_dims(NX, NU, 2)
    with pytest.raises(altro_amd.AltroHipError, match="uniform dimensions"):
        bt.set_model(altro_amd.MODEL_BICYCLE, 0.1)
    G = np.zeros((1, 5 + 2)); G[0, 5] = 1.0
    with pytest.raises(altro_amd.AltroHipError, match="differ in dimension"):
        bt.add_linear_constraint(0, 2, altro_amd.CONE_INEQUALITY, G, np.array([1.0]))     # k = 2 has three inputs
    assert bt.add_linear_constraint(0, 1, altro_amd.CONE_INEQUALITY, G, np.array([1.0])) == 0
