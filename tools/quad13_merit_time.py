#!/usr/bin/env python3
"""One MeritFunction evaluation (with derivative) of 13-state quaternion quadrotors on plan MFMA32, every problem searching:
    python tools/quad13_merit_time.py [batch] [forms]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import altro_amd  # noqa: E402
from tests.test_gpu_generic_model import H, N, m, make_case, n  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
forms = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0
c = make_case(batch, seed=7)
bt = altro_amd.Batch(N, n, m, batch)
bt.set_forms(forms)
bt.set_model(altro_amd.MODEL_QUADROTOR13, H)
bt.set_tracking_cost(np.stack([c["Qd"], c["Qfd"]]), c["Rd"][None], np.stack([c["xref"], c["xref"]]), c["uref"][None], k_stride_zero=True, batch_stride_zero=True)
bt.set_initial_state(c["x0"])
bt.set_input_guess(c["u0"][None, None], k_stride_zero=True, batch_stride_zero=True)
bt.open_loop_rollout(); bt.accept(); bt.expand(); bt.backward()
alphas = np.full(batch, 0.5)
for deriv in (True, False):
    bt.merit(alphas, derivative=deriv); bt.synchronize()
    ts = []
    for _ in range(5):
        bt.synchronize(); t0 = time.perf_counter()
        bt.merit(alphas, derivative=deriv)
        bt.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("forms %#x: %d vehicles, N = %d, merit with derivative %s: %.3f ms (host clock, best of 5)" % (forms, batch, N, deriv, min(ts)))
