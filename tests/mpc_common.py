"""Shared setup of the bicycle tracking-MPC scenario (test/bicycle_test.cpp:140-345) for the oracle and
the device path: N = 30, Qd = 1e-2, Rd = 1e-3, steering bound +-60 deg as an INEQUALITY block at every knot
point, reference = tests/problems.bicycle_reference."""
import numpy as np

from oracle import oracle
from tests import problems

N, n, m = 30, 4, 2
H = np.float32(0.1)
QD, RD = 1e-2, 1e-3
DELTA_MAX = 60 * np.pi / 180.0


def steering_block():
    G = np.zeros((2, n + m)); G[0, 3] = 1.0; G[1, 3] = -1.0      # bicycle_test.cpp:189-202
    return problems.CONE_INEQUALITY, G, np.full(2, DELTA_MAX)


def linear_costs(x_ref, first, u0):
    """bicycle_test.cpp:318-328: q_k = -Qd xref, c_k = 1/2 xref' Qd xref (+ 1/2 u0' Rd u0 for k < N)."""
    q = np.zeros((N + 1, n)); c = np.zeros(N + 1)
    c_u = 0.5 * float(u0 @ (RD * u0))
    for k in range(N + 1):
        xr = x_ref[k + first]
        qk = -(QD * xr)
        q[k] = qk
        c[k] = -(0.5 * float(qk @ xr)) + (c_u if k < N else 0.0)
    return q, c


def make_oracle(x_ref, u_ref, x0):
    s = oracle.ILQR(N, n, m, H, oracle.DYN_MODEL, oracle.MODEL_BICYCLE, cost_kind=oracle.COST_DIAGONAL)
    for k in range(N + 1):
        s.L.oracle_ilqr_set_lqr_cost(s.h, k, np.full(n, QD), np.full(m, RD), np.ascontiguousarray(x_ref[k]),
                                     np.ascontiguousarray(u_ref[min(k, len(u_ref) - 1)]))
    cone, G, g = steering_block()
    for k in range(N + 1):
        s.add_linear_constraint(k, cone, G, g)
    s.L.oracle_ilqr_set_initial_state(s.h, np.ascontiguousarray(x0, dtype=float))
    s.L.oracle_ilqr_initialize(s.h)
    u0 = np.array([u_ref[0][0], 0.0])
    for k in range(N):
        s.L.oracle_ilqr_set_input(s.h, k, u0)
    for k in range(N + 1):
        s.L.oracle_ilqr_set_state(s.h, k, np.ascontiguousarray(x_ref[k]))
    s.L.oracle_ilqr_set_options(s.h, 80, 1e-4, 1e-4, 1e-8, 1)
    return s, u0


def plant(x, u):
    L = oracle.lib()
    mdl = oracle.make_model(oracle.MODEL_BICYCLE)
    xn = np.zeros(n)
    import ctypes as C
    L.oracle_discrete_dynamics(C.byref(mdl), xn, np.ascontiguousarray(x), np.ascontiguousarray(u), H)
    return xn


def oracle_mpc(x_ref, u_ref, x0, nsim):
    """The loop of bicycle_test.cpp:302-337 on the oracle; returns per-step (iterations, u0, x_next, status)."""
    s, u0 = make_oracle(x_ref, u_ref, x0)
    x = np.array(x0, dtype=float)
    out = []
    for it in range(nsim):
        status, iters, log = s.solve()
        u = s.get("u")[0].copy()
        x = plant(x, u)
        out.append((iters, u, x.copy(), status))
        q, c = linear_costs(x_ref, it + 1, u0)
        for k in range(N + 1):
            s.L.oracle_ilqr_update_linear_costs(s.h, k, q[k].ctypes.data, None, float(c[k]))
        s.L.oracle_ilqr_set_initial_state(s.h, x)
        s.L.oracle_ilqr_shift_trajectory(s.h)
    return out
