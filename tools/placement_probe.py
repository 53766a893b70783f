#!/usr/bin/env python3
"""How the backward / forward kernel times of C1 depend on where the driver places a handle's buffers:
handles created one after another in one process, (a) each destroyed before the next, (b) all kept alive."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import altro_amd
from tests import problems

N, n, m, batch = 256, 12, 4, 4096
one = problems.c1_double_integrator(1, N=N)
x0 = 2.0 * problems.uniform01((batch, n), 21, 0) - 1.0


def make():
    bt = altro_amd.Batch(N, n, m, batch)
    bt.set_dynamics(one["A"][0, :1], one["B"][0, :1], None, k_stride_zero=True, batch_stride_zero=True)
    Q2 = np.stack([one["Q"][0, 0], one["Q"][0, N]])
    bt.set_cost(Q2, one["R"][0, :1], one["H"][0, :1], np.zeros((2, n)), one["r"][0, :1], k_stride_zero=True, batch_stride_zero=True)
    bt.set_initial_state(x0)
    return bt


def measure(bt):
    for _ in range(3):
        bt.sweep()
    bt.profile(True)
    for _ in range(10):
        bt.sweep()
    bt.synchronize()
    nb, ms_b, _ = bt.profile_get(0)
    nf, ms_f, _ = bt.profile_get(1)
    bt.profile(False)
    return ms_b / nb, ms_f / nf


for mode in ("destroy", "keep"):
    keep = []
    out = []
    for i in range(6):
        bt = make()
        out.append(measure(bt))
        if mode == "keep":
            keep.append(bt)
        else:
            bt.close()
    print(mode, " ".join("%.3f/%.3f" % o for o in out))
    for bt in keep:
        bt.close()
