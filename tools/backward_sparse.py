#!/usr/bin/env python3
"""tools/backward_sparse.py -- the (12, 4) backward sweep and the two-trial merit pass for a handful of problems (one wave each):
host-clock medians over the number of problems; under rocprofv3 --pmc the counters of one wave's walk over the horizon.

    python tools/backward_sparse.py [horizon] [batch,batch,...]
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
BATCHES = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [1, 4, 64, 1024, 4096]
n, m = 12, 4


def med(f, reps=25):
    f(); f()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); t.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(t))


print(f"# tools/backward_sparse.py: N = {N}, (12, 4), fp64; median of 25, host clock, ms (each call ends with a stream synchronisation)")
print(f"{'problems':>9s} {'backward':>10s} {'forward':>10s} {'rollout':>10s} {'whole LQ solve':>15s}")
for batch in BATCHES:
    one = problems.c1_double_integrator(1, N=N)
    bt = altro_amd.Batch(N, n, m, batch)
    bt.set_dynamics(one["A"][0, :1], one["B"][0, :1], None, k_stride_zero=True, batch_stride_zero=True)
    bt.set_tracking_cost(np.stack([np.ones(n), 100.0 * np.ones(n)]), np.full((1, m), 1e-2), np.zeros((2, n)), np.zeros((1, m)),
                         k_stride_zero=True, batch_stride_zero=True)
    bt.set_initial_state(2.0 * problems.uniform01((batch, n), 21) - 1.0)
    bt.set_input_guess(np.zeros((1, 1, m)), k_stride_zero=True, batch_stride_zero=True)
    bt.open_loop_rollout(); bt.accept(); bt.expand(); bt.backward()
    tb = med(lambda: (bt.backward(), bt.synchronize()))
    tf = med(lambda: (bt.forward_ltv(), bt.synchronize()))
    tr = med(lambda: (bt.open_loop_rollout(), bt.synchronize()))

    def solve():
        bt.set_input_guess(np.zeros((1, 1, m)), k_stride_zero=True, batch_stride_zero=True)
        bt.ilqr_solve(iterations_max=10)
    ts = med(solve, reps=9)
    print(f"{batch:9d} {tb:10.3f} {tf:10.3f} {tr:10.3f} {ts:15.3f}", flush=True)
    bt.close()
