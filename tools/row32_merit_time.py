#!/usr/bin/env python3
"""One MeritFunction evaluation (with derivative) of random LTV tracking problems at (n, m) on the plan AUTO picks, every problem searching:
    python tools/row32_merit_time.py n m [batch] [N] [forms]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import altro_amd  # noqa: E402
from tests import problems  # noqa: E402

n, m = int(sys.argv[1]), int(sys.argv[2])
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
N = int(sys.argv[4]) if len(sys.argv) > 4 else 128
forms = int(sys.argv[5], 0) if len(sys.argv) > 5 else 0
p = problems.ilqr12x4_problem(batch, N, True, n=n, m=m)
bt = altro_amd.Batch(N, n, m, batch)
bt.set_forms(forms)
bt.set_dynamics(p["A"], p["B"], p["f"])
bt.set_tracking_cost(p["Qd"], p["Rd"], p["xref"], p["uref"])
bt.set_initial_state(p["x0"]); bt.set_input_guess(p["u0"])
bt.open_loop_rollout(); bt.accept(); bt.expand(); bt.backward()
alphas = np.full(batch, 0.5)
bt.merit(alphas); bt.synchronize()
ts = []
for _ in range(7):
    bt.synchronize(); t0 = time.perf_counter()
    bt.merit(alphas)
    bt.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("(%d, %d) x %d problems, N = %d, plan %d, forms %#x: merit with derivative %.3f ms (host clock, best of 7)" % (n, m, batch, N, bt.plan, forms, min(ts)))
