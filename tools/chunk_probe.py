#!/usr/bin/env python3
"""Does a sweep of C1 (N = 256, (12, 4), 4096 problems, fp64) run faster as several handles of fewer problems on their own streams --
each chunk's backward outputs (K, d, P, p) still in the 256 MB memory-side cache when its forward sweep reads them?
    python tools/chunk_probe.py [chunks ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import altro_amd  # noqa: E402
from tests import problems  # noqa: E402

N, n, m, B = 256, 12, 4, 4096


def run(chunks, reps=20):
    per = B // chunks
    pr = problems.random_ltv(64, N, n, m)
    hs = []
    for _ in range(chunks):
        bt = altro_amd.Batch(N, n, m, per, plan=altro_amd.PLAN_MFMA16)
        bt.set_host_batch(64)
        bt.set_dynamics(pr["A"], pr["B"], pr["f"]); bt.set_cost(pr["Q"], pr["R"], pr["H"], pr["q"], pr["r"])
        bt.set_host_batch(0)
        bt.set_initial_state(np.tile(pr["x0"][:64], (per // 64, 1)))
        hs.append(bt)
    for _ in range(3):
        for bt in hs:
            bt.sweep()
        for bt in hs:
            bt.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        for bt in hs:
            bt.sweep()
        for bt in hs:
            bt.synchronize()
        ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))


if __name__ == "__main__":
    for c in [int(v) for v in sys.argv[1:]] or [1, 2, 4, 8, 16]:
        print("chunks %2d x %4d problems: %.3f ms per sweep of %d" % (c, B // c, run(c), B), flush=True)
