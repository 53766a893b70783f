#!/usr/bin/env python3
"""tools/merit_sparse.py -- what ONE line-search round (altro_hip_merit without derivative, constraint rows on) costs over the number
of problems that take part, in the row-layout (DPP) form and in the LDS form (a wave per (problem, trial)), C1's shape with input
bounds.  The late rounds of a constrained solve run for a handful of problems: their cost is one wave's dependent chain, and the
form with the shorter chain wins there whatever it does at full batch.

    python tools/merit_sparse.py [horizon] [batch,batch,...]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import altro_amd  # noqa: E402
from tests import problems  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n, m = 12, 4


def med(f, reps=25):
    f(); f()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); t.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(t))


print(f"# tools/merit_sparse.py: one merit evaluation (phi only) with two input-bound blocks, N = {N}, (12, 4), fp64; median of 25, host clock, ms")
print(f"{'problems':>9s} {'DPP form':>10s} {'LDS form':>10s} {'with derivative: DPP':>21s} {'LDS':>8s} {'no constraints: DPP':>20s} {'LDS':>8s}")
BATCHES = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [2, 8, 32, 128, 512, 1024, 2048, 4096]
for batch in BATCHES:
    one = problems.c1_double_integrator(1, N=N)
    bt = altro_amd.Batch(N, n, m, batch)
    bt.set_dynamics(one["A"][0, :1], one["B"][0, :1], None, k_stride_zero=True, batch_stride_zero=True)
    bt.set_tracking_cost(np.stack([np.ones(n), 100.0 * np.ones(n)]), np.full((1, m), 1e-2), np.zeros((2, n)), np.zeros((1, m)),
                         k_stride_zero=True, batch_stride_zero=True)
    bt.set_initial_state(2.0 * problems.uniform01((batch, n), 21) - 1.0)
    G = np.zeros((2 * m, n + m)); G[:m, n:] = np.eye(m); G[m:, n:] = -np.eye(m)
    bt.add_linear_constraint(0, N - 1, altro_amd.CONE_INEQUALITY, G[:m], np.full(m, 2.0))
    bt.add_linear_constraint(0, N - 1, altro_amd.CONE_INEQUALITY, G[m:], np.full(m, 2.0))
    bt.set_input_guess(np.zeros((1, 1, m)), k_stride_zero=True, batch_stride_zero=True)
    bt.open_loop_rollout(); bt.accept(); bt.expand(); bt.backward()
    alpha = np.full(batch, 0.5)
    out = []
    for deriv in (False, True):
        for form in ("1", "0"):
            os.environ["ALTRO_HIP_MERIT_DPP"] = form
            out.append(med(lambda: bt.merit(alpha, deriv)))
    bt.clear_constraints()
    bt.open_loop_rollout(); bt.accept(); bt.expand(); bt.backward()
    for form in ("1", "0"):
        os.environ["ALTRO_HIP_MERIT_DPP"] = form
        out.append(med(lambda: bt.merit(alpha, False)))
    os.environ.pop("ALTRO_HIP_MERIT_DPP")
    print(f"{batch:9d} {out[0]:10.3f} {out[1]:10.3f} {out[2]:21.3f} {out[3]:8.3f} {out[4]:20.3f} {out[5]:8.3f}", flush=True)
    bt.close()
