"""Iteration-by-iteration trace of problems of bench.py's constrained C1 workload (|u| <= 2) on the device against the oracle.

    python tests/soak/straggler_trace.py 118 [more problem indices]

For every i = 1 .. the device solve is repeated from scratch with iterations_max = i (duals reset), so that the state after
i iterations can be read from the per-problem results; the oracle's log has one row per iteration.
"""
import sys

import numpy as np

import altro_amd
from oracle import oracle
from tests import problems

Nf, batch = 256, 4096
idx = [int(a) for a in sys.argv[1:]] or [118]
one = problems.c1_double_integrator(1, N=Nf)
Qd = np.stack([np.ones(12), 100.0 * np.ones(12)])
x0 = (2.0 * problems.uniform01((batch, 12), 21) - 1.0)[idx]
Gb = np.zeros((8, 16)); Gb[:4, 12:] = np.eye(4); Gb[4:, 12:] = -np.eye(4)


def device(itmax):
    bt = altro_amd.Batch(Nf, 12, 4, len(idx))
    bt.set_dynamics(one["A"][0, :1], one["B"][0, :1], None, k_stride_zero=True, batch_stride_zero=True)
    bt.set_tracking_cost(Qd, np.full((1, 4), 1e-2), np.zeros((2, 12)), np.zeros((1, 4)), k_stride_zero=True, batch_stride_zero=True)
    bt.set_initial_state(x0)
    bt.set_input_guess(np.zeros((1, 1, 4)), k_stride_zero=True, batch_stride_zero=True)
    bt.add_linear_constraint(0, Nf - 1, altro_amd.CONE_INEQUALITY, Gb, np.full(8, 2.0))
    res = bt.ilqr_solve(iterations_max=itmax)
    bt.close()
    return res


A = np.ascontiguousarray(np.tile(one["A"][0, :1], (Nf, 1))); B = np.ascontiguousarray(np.tile(one["B"][0, :1], (Nf, 1)))
logs = []
for j in range(len(idx)):
    s = oracle.ILQR(Nf, 12, 4, 0.01, oracle.DYN_LINEAR, cost_kind=oracle.COST_DIAGONAL)
    s.L.oracle_ilqr_set_linear_dynamics(s.h, A, B, None)
    for k in range(Nf + 1):
        s.L.oracle_ilqr_set_lqr_cost(s.h, k, np.ascontiguousarray(Qd[1 if k == Nf else 0]), np.full(4, 1e-2), np.zeros(12), np.zeros(4))
    s.L.oracle_ilqr_set_initial_state(s.h, np.ascontiguousarray(x0[j]))
    for k in range(Nf):
        s.add_linear_constraint(k, oracle.CONE_INEQUALITY, Gb, np.full(8, 2.0))
    s.L.oracle_ilqr_initialize(s.h)
    s.set_penalty(1.0, 10.0)
    s.L.oracle_ilqr_set_options(s.h, 40, 1e-4, 1e-4, 1e-8, 0)
    status, iters, log = s.solve()
    logs.append((status, iters, log))
    print("problem %d: oracle status %d iterations %d" % (idx[j], status, iters))
maxit = max(l[1] for l in logs) + 1
for i in range(1, maxit + 1):
    r = device(i)
    for j in range(len(idx)):
        st, it, log = logs[j]
        row = log[min(i, it) - 1]
        print("it %2d p%-5d dev: status %d iters %2d alpha %.6g phi %.15g stat %.3e feas %.3e rho %g du %d | oracle: alpha %.6g phi %.15g (phi0 %.15g dphi0 %.3e) stat %.3e ls_it %d feas %.3e rho %g"
              % (i, idx[j], r["status"][j], r["iterations"][j], r["alpha"][j], r["phi"][j], r["stationarity"][j], r["feasibility"][j], r["penalty"][j], r["dual_updates"][j],
                 row[0], row[2], row[1], row[3], row[4], int(row[5]), row[6], row[7]))
