#!/usr/bin/env python3
"""Per-kernel latency of the LANE iLQR kernels on the C3 bicycle problem (run under rocprofv3 --kernel-trace):
    python tools/lane_latency.py BATCH DERIV(0|1) AL(0|1) [N] [pendulum]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import altro_amd
from tests import problems

batch, deriv, al = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
N = int(sys.argv[4]) if len(sys.argv) > 4 else 50
if len(sys.argv) > 5 and sys.argv[5] == "pendulum":
    bt = altro_amd.Batch(N, 2, 1, batch)
    bt.set_model(altro_amd.MODEL_PENDULUM, np.float32(0.03))
    xf = np.array([np.pi, 0.0])
    bt.set_tracking_cost(np.array([[1e-2, 1e-2], [1.0, 1.0]]), np.array([[1e-3]]), np.stack([xf, xf]), np.zeros((1, 1)),
                         k_stride_zero=True, batch_stride_zero=True)
    x0 = np.zeros((batch, 2)); x0[:, 0] = problems.uniform01((batch,), 22) - 0.5
    bt.set_initial_state(x0)
    bt.set_input_guess(np.array([[[0.1]]]), k_stride_zero=True, batch_stride_zero=True)
    bt.open_loop_rollout(); bt.accept(); bt.expand()
    for _ in range(10):
        bt.backward(); bt.forward_ltv()
        bt.merit(1.0, derivative=bool(deriv))
        bt.expand()
    bt.synchronize()
    sys.exit(0)
n, m, h = 4, 2, np.float32(0.1)
x_ref, u_ref = problems.bicycle_reference(N + 1)
bt = altro_amd.Batch(N, n, m, batch)
bt.set_model(altro_amd.MODEL_BICYCLE, h)
bt.set_tracking_cost(np.full((1, N + 1, n), 1e-2), np.full((1, N, m), 1e-3), x_ref[None, :N + 1], u_ref[None, :N],
                     batch_stride_zero=True)
if al:
    G = np.zeros((2, n + m)); G[0, 3] = 1; G[1, 3] = -1
    bt.add_linear_constraint(0, N, altro_amd.CONE_INEQUALITY, G, np.full(2, np.pi / 3))
x0 = x_ref[0] + (problems.uniform01((batch, n), 23, 0) - 0.5) * 0.4
bt.set_initial_state(x0)
bt.set_input_guess(np.array([[[u_ref[0][0], 0.0]]]), k_stride_zero=True, batch_stride_zero=True)
bt.open_loop_rollout(); bt.accept(); bt.expand()
for _ in range(10):
    bt.backward(); bt.forward_ltv()
    bt.merit(1.0, derivative=bool(deriv))
    bt.expand()
bt.synchronize()
