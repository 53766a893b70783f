"""Seeded synthetic problem generators shared by tests and bench.py.

PRNG: counter-based splitmix64(seed ^ index) -> U[0,1), seed = 20260928, so that any rank / shard
can regenerate exactly its slice without communication.  Pure numpy (host side only).
"""
import numpy as np

SEED = 20260928
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(idx, seed=SEED):
    z = (np.asarray(idx, dtype=np.uint64) ^ np.uint64(seed)) + np.uint64(0x9E3779B97F4A7C15)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    z = z ^ (z >> np.uint64(31))
    return z


def uniform01(shape, stream, first=0, seed=SEED):
    """U[0,1) doubles; `stream` separates arrays, `first` is the global element offset."""
    count = int(np.prod(shape))
    with np.errstate(over="ignore"):
        idx = np.arange(first, first + count, dtype=np.uint64) + (np.uint64(stream) << np.uint64(40))
        bits = splitmix64(idx, seed)
    return ((bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))).reshape(shape)


def normal(shape, stream, first=0, seed=SEED):
    count = int(np.prod(shape))
    u1 = uniform01((count,), stream, 2 * first, seed)
    u2 = uniform01((count,), stream + 1000, 2 * first, seed)
    return (np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2 * np.pi * u2)).reshape(shape)


def di_blocks(dim, h, n_inputs=None, float_h=True):
    """[A B] and nothing else for the discrete double integrator (test_utils.cpp:18-41).
    n_inputs < dim keeps only the first n_inputs actuated axes."""
    n = 2 * dim
    m = dim if n_inputs is None else n_inputs
    if float_h:
        hf = np.float32(h)
        b = float(np.float32(hf * hf) / np.float32(2))
        hd = float(hf)
    else:
        hd = float(h)
        b = hd * hd / 2
    A = np.eye(n)
    B = np.zeros((n, m))
    for i in range(dim):
        A[i, i + dim] = hd
        if i < m:
            B[i, i] = b
            B[i + dim, i] = hd
    return A, B


def tvlqr_kat_problem(kat, float_h):
    """The reference's TVLQR known-answer problem (tvlqr_test.cpp:23-66), is_diag data."""
    N, dim = kat["N"], kat["dim"]
    n, m = 2 * dim, dim
    A, B = di_blocks(dim, kat["h"], float_h=float_h)
    xeq = np.array(kat["xeq"], dtype=float)
    ueq = np.array(kat["ueq"], dtype=float)
    f = A @ xeq + B @ ueq
    Qd = np.full(n, kat["Qd"]); Rd = np.full(m, kat["Rd"]); Qfd = Qd * kat["Qf_scale"]
    q = np.full(n, kat["q"]); r = np.full(m, kat["r"])
    cm = lambda M: np.asarray(M).flatten(order="F")
    out = dict(N=N, n=n, m=m,
               A=np.tile(cm(A), (1, N, 1)), B=np.tile(cm(B), (1, N, 1)), f=np.tile(f, (1, N, 1)),
               Qdiag=np.concatenate([np.tile(Qd, (1, N, 1)), Qfd[None, None, :]], axis=1),
               Rdiag=np.tile(Rd, (1, N, 1)),
               q=np.tile(q, (1, N + 1, 1)), r=np.tile(r, (1, N, 1)),
               x0=np.array(kat["x0"], dtype=float)[None, :])
    # dense equivalents (is_diag = false path, solver.cpp:365)
    out["Q"] = np.stack([np.diag(v).flatten(order="F") for v in out["Qdiag"][0]])[None]
    out["R"] = np.stack([np.diag(v).flatten(order="F") for v in out["Rdiag"][0]])[None]
    out["H"] = np.zeros((1, N, m * n))
    return out


def random_ltv(batch, N, n, m, first=0, h=0.01, dtype=np.float64):
    """Random LTV-LQ problems (SURVEY 8d, config C4 recipe): A = I + h G, B = h N(0,1),
    Q = I + 0.1 L L^T, R = 0.1 I + 0.01 M M^T, Qf = 10 Q, q,r ~ N(0, 0.1^2), f ~ h N(0,1)."""
    G = normal((batch, N, n, n), 1, first * N * n * n) * 0.3
    A = np.eye(n)[None, None] + h * G
    B = h * normal((batch, N, n, m), 3, first * N * n * m)
    f = h * normal((batch, N, n), 5, first * N * n)
    L = normal((batch, N + 1, n, n), 7, first * (N + 1) * n * n)
    Q = np.eye(n)[None, None] + 0.1 * L @ np.swapaxes(L, -1, -2)
    Q[:, N] *= 10.0
    M = normal((batch, N, m, m), 9, first * N * m * m)
    R = 0.1 * np.eye(m)[None, None] + 0.01 * M @ np.swapaxes(M, -1, -2)
    H = 0.01 * normal((batch, N, m, n), 11, first * N * m * n)
    q = 0.1 * normal((batch, N + 1, n), 13, first * (N + 1) * n)
    r = 0.1 * normal((batch, N, m), 15, first * N * m)
    x0 = 2.0 * uniform01((batch, n), 17, first * n) - 1.0
    colmajor = lambda T: np.ascontiguousarray(np.swapaxes(T, -1, -2)).reshape(T.shape[0], T.shape[1], -1)
    return dict(N=N, n=n, m=m, A=colmajor(A).astype(dtype), B=colmajor(B).astype(dtype),
                f=f.astype(dtype), Q=colmajor(Q).astype(dtype), R=colmajor(R).astype(dtype),
                H=colmajor(H).astype(dtype), q=q.astype(dtype), r=r.astype(dtype),
                x0=x0.astype(dtype))


def c1_double_integrator(batch, N=256, first=0, h=0.01):
    """Config C1 (BASELINE.json configs[1]): DI dim=6 -> n=12, first four axes actuated -> m=4,
    Q=I, R=1e-2 I, Qf=100 I, xref=0, x0 ~ U(-1,1); time-varying STORAGE (blocks repeated per k)."""
    dim, n, m = 6, 12, 4
    A, B = di_blocks(dim, h, n_inputs=m, float_h=True)
    cm = lambda M_: np.asarray(M_).flatten(order="F")
    Q = np.eye(n); R = 1e-2 * np.eye(m); Qf = 100.0 * np.eye(n)
    tile = lambda v, K: np.broadcast_to(v, (batch, K) + v.shape).copy()
    Qs = tile(cm(Q), N + 1)
    Qs[:, N] = cm(Qf)
    return dict(N=N, n=n, m=m, A=tile(cm(A), N), B=tile(cm(B), N), f=np.zeros((batch, N, n)),
                Q=Qs, R=tile(cm(R), N), H=np.zeros((batch, N, m * n)),
                q=np.zeros((batch, N + 1, n)), r=np.zeros((batch, N, m)),
                x0=2.0 * uniform01((batch, n), 21, first * n) - 1.0)


# ---- constrained double integrator (test/double_integrator_test.cpp:170-493) -------------------------
CONE_EQUALITY, CONE_IDENTITY, CONE_INEQUALITY, CONE_SOC = 0, 1, 2, 3


def di_constraint_blocks(kind, N, n=4, m=2, xf=None, u_bnd=1.0):
    """Linear constraint blocks c = G [x;u] - g of the reference's three constrained double-integrator
    tests, as a list of (k_first, k_last_inclusive, cone, G[p, n+m], g[p]).
      goal   : x_N - xf in {0}                                   (double_integrator_test.cpp:188-203,225)
      bounds : goal + [u - ub; -ub - u] <= 0 at k = 0..N-1        (:296-317,333-334)
      soc    : goal + [u; ub] in SOC at k = 0..N-1                (:414-432,447-448)"""
    xf = np.zeros(n) if xf is None else np.asarray(xf, dtype=np.float64)
    w = n + m
    G = np.zeros((n, w)); G[:, :n] = np.eye(n)
    blocks = [(N, N, CONE_EQUALITY, G, xf.copy())]
    if kind == "bounds":
        G = np.zeros((2 * m, w))
        G[:m, n:] = np.eye(m)
        G[m:, n:] = -np.eye(m)
        blocks.append((0, N - 1, CONE_INEQUALITY, G, np.full(2 * m, u_bnd)))
    elif kind == "soc":
        G = np.zeros((m + 1, w))
        G[:m, n:] = np.eye(m)
        g = np.zeros(m + 1); g[m] = -u_bnd
        blocks.append((0, N - 1, CONE_SOC, G, g))
    else:
        assert kind == "goal"
    return blocks


# ---- bicycle tracking MPC (test/bicycle_test.cpp:140-345 with a synthetic reference, SURVEY.md 8d "C3") ----
def bicycle_reference(steps, h=0.1, v=6.3):
    """Seed-free synthetic reference path: the CoG bicycle (L = 2.7, lr = 1.5) rolled out with speed v and
    steering rate 0.05 sin(0.2 t) by the explicit midpoint rule.  (The reference's own test tracks the
    'scotty' path: tests/golden/scotty_reference.json, `scotty()` below, tracked by tests/test_oracle_mpc.py and
    tests/test_gpu_scotty.py.)  Returns x_ref [steps + 1, 4], u_ref [steps, 2]."""
    L, lr = 2.7, 1.5

    def f(x, u):
        beta = np.arctan2(lr * x[3], L)
        s, c = np.sin(x[2] + beta), np.cos(x[2] + beta)
        return np.array([u[0] * c, u[0] * s, u[0] * np.cos(beta) * np.tan(x[3]) / L, u[1]])

    x = np.zeros(4)
    xs, us = [x.copy()], []
    for i in range(steps):
        u = np.array([v, 0.05 * np.sin(0.2 * i * h)])
        xm = x + 0.5 * h * f(x, u)
        x = x + h * f(xm, u)
        xs.append(x.copy()); us.append(u)
    return np.array(xs), np.array(us)


def scotty():
    """The reference's own MPC scenario as data (tests/golden/make_scotty_fixtures.py): the path of test/scotty.json
    (501 points, spacing 0.1 s) and what test/bicycle_test.cpp:266-359 saved after tracking it for 200 receding-horizon
    steps.  Returns x_ref [501, 4], u_ref [501, 2], expected {solve_iters, state_trajectory, input_trajectory,
    tracking_error} as arrays."""
    import json
    import os
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    with open(os.path.join(g, "scotty_reference.json")) as f:
        ref = json.load(f)
    with open(os.path.join(g, "scotty_mpc_expected.json")) as f:
        mpc = json.load(f)
    exp = {k: np.array(mpc[k]) for k in ("solve_iters", "state_trajectory", "input_trajectory", "tracking_error")}
    return np.array(ref["state_trajectory"]), np.array(ref["input_trajectory"]), exp


# ---- (12, 4) iLQR problems with dynamics given as data (plan MFMA16; tests/test_gpu_ilqr_mfma16.py) --------------------
def ilqr12x4_problem(batch, N, with_f, n=12, m=4):
    """Random LTV dynamics + a diagonal tracking cost with per-problem, per-knot-point weights and references."""
    pr = random_ltv(batch, N, n, m)
    return dict(A=pr["A"], B=pr["B"], f=pr["f"] if with_f else None,
                Qd=1.0 + uniform01((batch, N + 1, n), 71), Rd=0.1 + 0.2 * uniform01((batch, N, m), 72),
                xref=normal((batch, N + 1, n), 73) * 0.3, uref=normal((batch, N, m), 74) * 0.1,
                x0=normal((batch, n), 75), u0=normal((batch, N, m), 76) * 0.2)


def ilqr12x4_constraint_blocks(N, n=12, m=4):
    """|u| <= 0.3 at every k < N, x1 <= 1.2 and -x2 <= 1.2 at 1 <= k < N (INEQUALITY), u_0[0] == 0.05 (EQUALITY)."""
    w = n + m
    Gb = np.zeros((2 * m, w)); Gb[:m, n:] = np.eye(m); Gb[m:, n:] = -np.eye(m)
    Gs = np.zeros((2, w)); Gs[0, 1] = 1.0; Gs[1, 2] = -1.0
    Ge = np.zeros((1, w)); Ge[0, 12] = 1.0
    return [(0, N - 1, CONE_INEQUALITY, Gb, np.full(2 * m, 0.3)),
            (1, N - 1, CONE_INEQUALITY, Gs, np.array([1.2, 1.2])),
            (0, 0, CONE_EQUALITY, Ge, np.array([0.05]))]


def quadratic_cost(batch, N, n, m, stream=81):
    """A dense quadratic cost per problem and knot point, in the blocks ALTROSolver::SetQuadraticCost takes
    (altro_solver.cpp:118-136, column-major): Q = I + 0.1 L L^T (n x n, symmetric positive definite), R = 0.1 I + 0.02 M M^T,
    H = 0.03 N(0, 1) (m x n: small enough that [Q H^T; H R] stays positive definite), q, r ~ N(0, 0.3^2), c ~ U(0, 1).
    Shapes: Q [batch][N+1][n*n], R [batch][N][m*m], H [batch][N][m*n], q [batch][N+1][n], r [batch][N][m], c [batch][N+1]."""
    L = normal((batch, N + 1, n, n), stream)
    Q = np.eye(n) + 0.1 * L @ np.swapaxes(L, -1, -2)
    M = normal((batch, N, m, m), stream + 1)
    R = 0.1 * np.eye(m) + 0.02 * M @ np.swapaxes(M, -1, -2)
    H = 0.03 * normal((batch, N, m, n), stream + 2)
    col = lambda X: np.ascontiguousarray(np.swapaxes(X, -1, -2)).reshape(X.shape[0], X.shape[1], -1)   # column-major blocks
    return dict(Q=col(Q), R=col(R), H=col(H), q=0.3 * normal((batch, N + 1, n), stream + 3), r=0.3 * normal((batch, N, m), stream + 4),
                c=uniform01((batch, N + 1), stream + 5))


def quadrotor_ltv(batch, N, h=0.02, seed_stream=131):
    """Time-varying LQ problems from linearisations of the 12-state quadrotor test model (oracle/models_oracle.c) along seeded random
    states near hover: A_k, B_k of the explicit midpoint rule, a dense cost with a cross term.  Chains of integrators (torque ->
    rate -> angle -> velocity -> position): the problem class on which an UNSYMMETRISED cost-to-go recursion lets the antisymmetric
    rounding noise of P double every step (tests/test_gpu_parity.py::test_backward_sweep_keeps_p_symmetric)."""
    import ctypes as C
    from oracle import oracle
    L = oracle.lib()
    mdl = oracle.make_model(oracle.MODEL_QUADROTOR)
    n, m = 12, 4
    xs = 0.3 * normal((batch, N, n), seed_stream)
    us = np.array([0.5 * 9.81, 0, 0, 0]) + normal((batch, N, m), seed_stream + 1) * np.array([0.5, 0.01, 0.01, 0.01])
    A = np.zeros((batch, N, n * n)); B = np.zeros((batch, N, n * m))
    J = np.zeros(n * (n + m))
    for b in range(batch):
        for k in range(N):
            L.oracle_discrete_jacobian(C.byref(mdl), J, np.ascontiguousarray(xs[b, k]), np.ascontiguousarray(us[b, k]), np.float32(h))
            A[b, k] = J[:n * n]; B[b, k] = J[n * n:]
    cost = quadratic_cost(batch, N, n, m, stream=seed_stream + 2)
    R = cost["R"] + np.diag([0.0, 20.0, 20.0, 20.0]).reshape(-1)
    return dict(N=N, n=n, m=m, A=A, B=B, f=np.zeros((batch, N, n)), Q=cost["Q"], R=R, H=cost["H"], q=cost["q"], r=cost["r"],
                x0=normal((batch, n), seed_stream + 8))
