#!/usr/bin/env python3
"""Speculative backtracking on/off: identical results, fewer merit launches (run twice: ALTRO_HIP_NO_SPECULATION=1 and unset)."""
import os, sys, time, hashlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import altro_amd
from tests import problems
n, m, N, batch = 4, 2, 50, 8192
h = np.float32(0.1)
x_ref, u_ref = problems.bicycle_reference(N + 1)
bt = altro_amd.Batch(N, n, m, batch)
bt.set_model(altro_amd.MODEL_BICYCLE, h)
bt.set_tracking_cost(np.full((1, N + 1, n), 1e-2), np.full((1, N, m), 1e-3), x_ref[None, :N + 1], u_ref[None, :N], batch_stride_zero=True)
G = np.zeros((2, n + m)); G[0, 3] = 1; G[1, 3] = -1
bt.add_linear_constraint(0, N, altro_amd.CONE_INEQUALITY, G, np.full(2, np.pi / 3))
x0 = x_ref[0] + (problems.uniform01((batch, n), 23, 0) - 0.5) * 0.4
bt.set_initial_state(x0)
for rep in range(2):
    bt.set_input_guess(np.array([[[u_ref[0][0], 0.0]]]), k_stride_zero=True, batch_stride_zero=True)
    bt.synchronize()
    t = time.perf_counter(); res = bt.ilqr_solve(iterations_max=80, use_backtracking=True); bt.synchronize(); dt = time.perf_counter() - t
x, u = bt.get_nominal()
dig = hashlib.sha256(x.tobytes() + u.tobytes() + res["iterations"].tobytes() + res["status"].tobytes()).hexdigest()[:16]
print("spec=%s  %.4f s  sweeps %d  merit launches %d  mean iterations %.3f  digest %s"
      % ("off" if os.environ.get("ALTRO_HIP_NO_SPECULATION") else "on", dt, int(res["sweeps"]), int(res["merit_launches"]), res["iterations"].mean(), dig))
