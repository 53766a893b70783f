#!/usr/bin/env python3
"""Soak: whole (AL-)iLQR solves with PER-KNOT-POINT dimensions on the batched ABI (altro_hip_batch_create_dims, plan GENERIC) against the
oracle on the equivalent zero-padded uniform problem (tests/test_gpu_ragged_ilqr.py has the construction): random dimension sequences,
random input bounds on runs of knot points with equal dimensions, a terminal pin every other case.

    python tests/soak/fuzz_ragged_ilqr.py [cases] [seed]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import altro_amd  # noqa: E402
import tests.test_gpu_ragged_ilqr as T  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(cases):
    Nn = int(rng.integers(3, 16))
    hi = int(rng.choice([6, 10, 18, 40]))
    nx = rng.integers(1, hi + 1, size=Nn + 1)
    nu = rng.integers(1, min(hi, 12) + 1, size=Nn)
    for k in range(1, Nn):                      # runs of equal dimensions, so that blocks over ranges exist
        if rng.random() < 0.5:
            nx[k] = nx[k - 1]; nu[k] = nu[k - 1]
    T.NX, T.NU, T.N, T.NMAX, T.MMAX = nx, nu, Nn, int(nx.max()), int(nu.max())
    batch = int(rng.integers(1, 7))
    p = T.make_problem(batch, seed=int(rng.integers(1, 1 << 30)))
    bounds = []
    constrained = rng.random() < 0.7
    if constrained:
        k = 0
        while k < Nn:
            k1 = k
            while k1 + 1 < Nn and nx[k1 + 1] == nx[k] and nu[k1 + 1] == nu[k]:
                k1 += 1
            if rng.random() < 0.6:
                n, m = int(nx[k]), int(nu[k])
                G = np.zeros((2 * m, n + m)); G[:m, n:] = np.eye(m); G[m:, n:] = -np.eye(m)
                bounds.append((k, k1, G, np.full(2 * m, float(rng.uniform(0.15, 0.6)))))
            k = k1 + 1
        if it % 2 == 0:
            G = np.zeros((1, int(nx[Nn]))); G[0, 0] = 1.0
            bounds.append((Nn, Nn, G, np.array([0.2])))
    if os.environ.get("ONLY") and int(os.environ["ONLY"]) != it:
        continue
    if os.environ.get("ONLY"):     # the same padded problem the oracle gets, on the device's uniform-dimension path: a three-way comparison
        NM, MM = T.NMAX, T.MMAX
        A = np.zeros((batch, Nn, NM * NM)); B = np.zeros((batch, Nn, NM * MM)); f = np.zeros((batch, Nn, NM))
        Q = np.zeros((batch, Nn + 1, NM * NM)); R = np.zeros((batch, Nn, MM * MM)); H = np.zeros((batch, Nn, MM * NM))
        q = np.zeros((batch, Nn + 1, NM)); r = np.zeros((batch, Nn, MM)); x0 = np.zeros((batch, NM)); u0 = np.zeros((batch, Nn, MM))
        for b in range(batch):
            for k in range(Nn + 1):
                n = nx[k]
                Qk = np.zeros((NM, NM)); Qk[:n, :n] = p["Q"][k][b]; Q[b, k] = Qk.T.reshape(-1); q[b, k, :n] = p["q"][k][b]
                if k < Nn:
                    m, n2 = nu[k], nx[k + 1]
                    Ak = np.zeros((NM, NM)); Ak[:n2, :n] = p["A"][k][b]; A[b, k] = Ak.T.reshape(-1)
                    Bk = np.zeros((NM, MM)); Bk[:n2, :m] = p["B"][k][b]; B[b, k] = Bk.T.reshape(-1); f[b, k, :n2] = p["f"][k][b]
                    Rk = np.eye(MM); Rk[:m, :m] = p["R"][k][b]; R[b, k] = Rk.T.reshape(-1)
                    Hk = np.zeros((MM, NM)); Hk[:m, :n] = p["H"][k][b]; H[b, k] = Hk.T.reshape(-1)
                    r[b, k, :m] = p["r"][k][b]; u0[b, k, :m] = p["u0"][k][b]
            x0[b, :nx[0]] = p["x0"][b]
        bu = altro_amd.Batch(Nn, NM, MM, batch, plan=altro_amd.PLAN_GENERIC)
        bu.set_dynamics(A, B, f); bu.set_quadratic_cost(Q, R, H, q, r, p["c"]); bu.set_initial_state(x0); bu.set_input_guess(u0)
        for (k0, k1, G, g) in bounds:
            for k in range(k0, k1 + 1):
                n, m = nx[k], (nu[k] if k < Nn else 0)
                Gp = np.zeros((G.shape[0], NM + MM)); Gp[:, :n] = G[:, :n]; Gp[:, NM:NM + m] = G[:, n:n + m]
                bu.add_linear_constraint(k, k, altro_amd.CONE_EQUALITY if k0 == Nn else altro_amd.CONE_INEQUALITY, Gp, g)
        ru = bu.ilqr_solve(iterations_max=60, tol_stationarity=1e-4, penalty_initial=1.0, penalty_scaling=10.0)
        print("   uniform padded problem on the device: status", ru["status"], "iterations", ru["iterations"])
    bt = T.make_hip(p, batch, bounds)
    res = bt.ilqr_solve(iterations_max=60, tol_stationarity=1e-4, penalty_initial=1.0, penalty_scaling=10.0)
    x, u = bt.get_nominal()
    ok = True
    for b in range(batch):
        s = T.make_oracle(p, b, bounds)
        if constrained:
            s.set_penalty(1.0, 10.0)
        s.L.oracle_ilqr_set_options(s.h, 60, 1e-4, 1e-4, 1e-8, 0)
        status, iters, _ = s.solve()
        same = res["status"][b] == status and res["iterations"][b] == iters
        if same and status == 0:
            same = np.abs(x[b] - T.unpad_x(s.get("x"))).max() < 1e-6 and np.abs(u[b] - T.unpad_u(s.get("u"))).max() < 1e-5
        if not same:
            ok = False
            print("   problem %d: device status %d iterations %d, oracle %d %d" % (b, res["status"][b], res["iterations"][b], status, iters))
    bad += not ok
    print("%s case %3d: N = %2d nx in [%d, %d] nu in [%d, %d] batch = %d blocks %d sweeps %d converged %d/%d" % (
        "ok " if ok else "BAD", it, Nn, nx.min(), nx.max(), nu.min(), nu.max(), batch, len(bounds), int(res["iterations"].max()),
        int((res["status"] == 0).sum()), batch), flush=True)
    bt.close()
print("%d of %d cases differ" % (bad, cases))
sys.exit(1 if bad else 0)
