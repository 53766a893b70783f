#!/usr/bin/env python3
"""One-off soak: speculative line-search steps on/off give the same bits (several models, batches, both searches)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import altro_amd
from tests import problems
from tests.test_gpu_speculation import _solve_both, _same, _bicycle, _linear_12x4


def pendulum(batch, N=60, spread=1.0):
    n, m, h = 2, 1, np.float32(0.03)

    def make():
        bt = altro_amd.Batch(N, n, m, batch)
        bt.set_model(altro_amd.MODEL_PENDULUM, h)
        xf = np.array([np.pi, 0.0])
        bt.set_tracking_cost(np.array([[1e-2, 1e-2], [1.0, 1.0]]), np.array([[1e-3]]), np.stack([xf, xf]), np.zeros((1, m)),
                             k_stride_zero=True, batch_stride_zero=True)
        x0 = np.zeros((batch, n)); x0[:, 0] = spread * (problems.uniform01((batch,), 22, 0) - 0.5)
        bt.set_initial_state(x0)
        bt.set_input_guess(np.array([[[0.1]]]), k_stride_zero=True, batch_stride_zero=True)
        return bt
    return make


cases = [("bicycle 300", _bicycle(300), dict(iterations_max=50)), ("bicycle 5000", _bicycle(5000, N=30), dict(iterations_max=40)),
         ("pendulum 1000", pendulum(1000), dict(iterations_max=40)), ("pendulum 70", pendulum(70, spread=3.0), dict(iterations_max=60)),
         ("linear 12x4 700", _linear_12x4(700), dict(iterations_max=30)), ("linear 12x4 9", _linear_12x4(9, N=7), dict(iterations_max=30))]
for name, make, opts in cases:
    for bt_ in (True, False):
        off, on = _solve_both(make, use_backtracking=bt_, **opts)
        _same(off, on)
        print("%-16s backtracking=%d  merit launches %4d -> %4d  converged %d/%d" % (
            name, bt_, int(off[0]["merit_launches"]), int(on[0]["merit_launches"]), int((on[0]["status"] == 0).sum()), len(on[0]["status"])))
print("ok")
