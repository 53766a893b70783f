"""Problem definitions of the dense golden fixtures (tests/golden/dense_fixtures.npz): shared by the generator
(tests/golden/make_dense_fixtures.py, which runs the oracle) and by the tests that replay the fixtures -- the GPU one
(tests/test_gpu_golden.py) must not need the oracle, so nothing here imports it."""
import hashlib

import numpy as np

from tests import problems

P_KNOTS = {}      # (round 2 kept P of c1 / c4 at three knot points only; every knot point is stored now)


def checksum(pr):
    h = hashlib.sha256()
    for k in sorted(pr):
        if isinstance(pr[k], np.ndarray):
            h.update(k.encode()); h.update(np.ascontiguousarray(pr[k]).tobytes())
    return np.frombuffer(h.digest()[:8], dtype=np.uint64)[0]


def tvlqr_problem(name):
    """The TVLQR inputs of fixture `name` (shared with the tests)."""
    if name == "c1":
        return problems.c1_double_integrator(2, N=256)
    if name == "c4":      # configs[4]: fp32 data (the oracle runs in fp64 on the fp32-rounded inputs)
        pr = problems.random_ltv(2, 512, 12, 4)
        return {k: (v.astype(np.float32).astype(np.float64) if isinstance(v, np.ndarray) else v) for k, v in pr.items()}
    if name == "c2shape":
        return problems.random_ltv(3, 100, 2, 1)
    if name == "c3shape":
        return problems.random_ltv(3, 50, 4, 2)
    raise KeyError(name)


def solve_case(name):
    """Solve fixture `name`: model (altro_hip_model id + name), N, n, m, h, diagonal costs, goal, initial states, input
    guess, double-integrator dimension, backtracking line search flag, iterations_max."""
    f32 = np.float32
    if name in ("di_n10", "di_n50"):     # BASELINE.json configs[0]: double_integrator_test.cpp:69-85 at N = 10 and N = 50
        N = 10 if name == "di_n10" else 50
        return dict(model_name="double_integrator", model=1, N=N, n=4, m=2, h=f32(f32(5.0) / f32(N)), Qd=[1.0] * 4, Rd=[1e-2] * 2,
                    Qfd=[1.0] * 4, xf=[0.0] * 4, x0s=np.array([[1.0, 2.0, 0.0, 0.0]]), u0=[0.0, 0.0], dim=2, bt=0, itmax=3)
    if name == "pendulum":
        return dict(model_name="pendulum", model=2, N=50, n=2, m=1, h=f32(f32(3.0) / 50.0), Qd=[1e-2, 1e-2], Rd=[1e-3],
                    Qfd=[1.0, 1.0], xf=[np.pi, 0.0], x0s=np.array([[-0.4, 0.0], [0.0, 0.0], [0.3, 0.0]]), u0=[0.1], dim=0,
                    bt=0, itmax=30)
    if name == "bicycle":
        return dict(model_name="bicycle", model=3, N=30, n=4, m=2, h=f32(f32(3.0) / 30.0), Qd=[1e-2] * 4, Rd=[1e-3] * 2,
                    Qfd=[10.0] * 4, xf=[1, 2, np.pi / 2, 0.0], x0s=0.02 * (np.arange(12).reshape(3, 4) % 7 - 3) / 3.0,
                    u0=[0.5, 0.0], dim=0, bt=1, itmax=30)
    raise KeyError(name)


TVLQR = ("c1", "c4", "c2shape", "c3shape")
SOLVES = ("di_n10", "di_n50", "pendulum", "bicycle")

# ---- round 3: the rows of SURVEY.md section 8 that round 2's fixtures did not replay -----------------------------------
AL_KINDS = ("goal", "bounds", "soc")      # row f2: double_integrator_test.cpp:170-493 (3 / 5 / 9 iterations)


def load_kats():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_kats.json")))


def al_case(kats, kind):
    """Constrained double integrator: the reference's own start (its iteration count is the pin) + four shifted ones."""
    kat = kats["double_integrator_constrained"]
    c = kat[kind]
    N, dim = kat["N"], kat["dim"]
    n, m = 2 * dim, dim
    x0s = np.tile(np.array(c["x0"], dtype=float), (5, 1))
    x0s[1:, 0] += 0.05 * np.array([-4, -1, 2, 4]); x0s[1:, 1] -= 0.04 * np.array([1, 3, 5, 6]); x0s[1:, 2] += 0.02 * np.array([1, 2, 3, 4])
    xf = np.array(kat["xf"], dtype=float)
    return dict(N=N, n=n, m=m, dim=dim, h=np.float32(np.float32(kat["tf"]) / np.float32(N)), Q=kat["Q"], R=kat["R"], xf=xf,
                x0s=x0s, blocks=problems.di_constraint_blocks(kind, N, n, m, xf, kat["u_bnd"]),
                penalty_initial=c["penalty_initial"], penalty_scaling=c["penalty_scaling"], iterations=c["iterations"], itmax=60)


def mpc_case():
    """Row f3: the receding-horizon loop of bicycle_test.cpp:302-337 (tests/mpc_common.py's scenario), 3 vehicles x 4 steps."""
    N, n, m, nsim = 30, 4, 2, 4
    x_ref, u_ref = problems.bicycle_reference(N + nsim + 1)
    off = problems.uniform01((3, 4), 33) - 0.5
    x0s = x_ref[0] + off * np.array([0.2, 0.2, 0.04, 0.0])
    G = np.zeros((2, n + m)); G[0, 3] = 1.0; G[1, 3] = -1.0
    return dict(N=N, n=n, m=m, nsim=nsim, h=np.float32(0.1), QD=1e-2, RD=1e-3, x_ref=x_ref, u_ref=u_ref, x0s=x0s,
                G=G, g=np.full(2, 60 * np.pi / 180.0), u0=np.array([u_ref[0][0], 0.0]))


def mpc_linear_costs(c, first):
    """bicycle_test.cpp:318-328: q_k = -Qd xref, c_k = 1/2 xref' Qd xref (+ 1/2 u0' Rd u0 for k < N)."""
    N, n = c["N"], c["n"]
    q = np.zeros((N + 1, n)); cc = np.zeros(N + 1)
    c_u = 0.5 * float(c["u0"] @ (c["RD"] * c["u0"]))
    for k in range(N + 1):
        xr = c["x_ref"][k + first]
        qk = -(c["QD"] * xr)
        q[k] = qk
        cc[k] = -(0.5 * float(qk @ xr)) + (c_u if k < N else 0.0)
    return q, cc


def lq12_case(constrained):
    """Rows a5-a12 on plan MFMA16: (12, 4) iLQR with dynamics as data, N = 24 (and its constrained variant)."""
    N = 24
    p = problems.ilqr12x4_problem(4, N, True)
    return dict(N=N, p=p, blocks=problems.ilqr12x4_constraint_blocks(N) if constrained else [], itmax=60 if constrained else 10)


def quad12_case(constrained):
    """Row a9 on plan MFMA16: the same (12, 4) problems with the DENSE quadratic cost of ALTROSolver::SetQuadraticCost
    (Q, R, H != 0, q, r, c per problem and knot point: problems.quadratic_cost) instead of the tracking cost."""
    c = lq12_case(constrained)
    c["p"] = dict(c["p"], **problems.quadratic_cost(4, c["N"], 12, 4))
    return c


def quad4_case(constrained):
    """Row a9 on plan LANE: the bicycle (4, 2) with the dense quadratic cost, with and without input bounds |u| <= 0.4."""
    N, n, m = 20, 4, 2
    x0s = 0.02 * (np.arange(3 * 4).reshape(3, 4) % 7 - 3) / 3.0
    blocks = []
    if constrained:
        Gb = np.zeros((2 * m, n + m)); Gb[:m, n:] = np.eye(m); Gb[m:, n:] = -np.eye(m)
        blocks = [(0, N - 1, problems.CONE_INEQUALITY, Gb, np.full(2 * m, 0.4))]
    return dict(N=N, n=n, m=m, h=np.float32(0.1), model=3, model_name="bicycle", dim=0, u0=[0.5, 0.0], x0s=x0s,
                cost=problems.quadratic_cost(3, N, n, m), blocks=blocks, itmax=40)


MERIT_ALPHAS = (0.0, 0.35, 1.0)


def merit_case(name):
    """Rows a5, a7-a9, a11: one merit evaluation with derivative after rollout / accept / expand / backward: phi, phi',
    candidate trajectory, refreshed lx, lu (and A, B for a device model), stationarity."""
    if name == "pendulum":
        c = solve_case("pendulum")
        return dict(c, x0s=np.array([[-0.4, 0.0], [0.2, 0.1], [0.5, -0.3]]))
    if name == "bicycle":
        c = solve_case("bicycle")
        return c
    raise KeyError(name)


REG_CASE = dict(name="c3shape", reg=0.37)     # row f4 / a1's reg argument: tvlqr.cpp:159-164
