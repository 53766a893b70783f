#!/usr/bin/env python3
"""What the per-launch hipEvents of bench.py's timed region cost: C1 sweeps with profile mode 2 (two events around every launch,
as the bench line's kernel durations need) and with no events, alternating.   python tools/event_overhead.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import altro_amd  # noqa: E402
from tests import problems  # noqa: E402

N, n, m, batch = 256, 12, 4, 4096
one = problems.c1_double_integrator(1, N=N)
bt = altro_amd.Batch(N, n, m, batch)
bt.set_dynamics(one["A"][0, :1], one["B"][0, :1], None, k_stride_zero=True, batch_stride_zero=True)
bt.set_tracking_cost(np.stack([np.ones(n), 100.0 * np.ones(n)]), np.full((1, m), 1e-2), np.zeros((2, n)), np.zeros((1, m)),
                     k_stride_zero=True, batch_stride_zero=True)
bt.set_initial_state(2.0 * problems.uniform01((batch, n), 21) - 1.0)
for rnd in range(3):
    for mode in (0, 2):
        bt.profile(mode)
        for _ in range(10):
            bt.sweep(0.0)
        bt.synchronize()
        t0 = time.perf_counter()
        for _ in range(400):
            bt.sweep(0.0)
        bt.synchronize()
        dt = (time.perf_counter() - t0) / 400 * 1e3
        bt.profile(0)
        print("profile mode %d: %.4f ms per sweep" % (mode, dt))
