#!/usr/bin/env python3
"""Sweep time (backward + forward, ms) of shapes next to the fast ones, plan AUTO against plan GENERIC -- the "shape cliff"
of VERDICT r2 (weak / item 7): (12, 3) or (10, 4) used to get a kernel 8x slower than (12, 4).   python tools/shape_cliff.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import altro_amd  # noqa: E402
from tests import problems  # noqa: E402

PLAN = {1: "GENERIC", 2: "MFMA16", 3: "LANE"}


def sweep_ms(N, n, m, batch, plan):
    pr = problems.random_ltv(64, N, n, m)
    bt = altro_amd.Batch(N, n, m, batch, plan=plan)
    bt.set_host_batch(64) if bt.plan != altro_amd.PLAN_GENERIC else None
    if bt.plan == altro_amd.PLAN_GENERIC:
        pr = {k: (np.tile(v, (batch // 64,) + (1,) * (v.ndim - 1)) if isinstance(v, np.ndarray) else v) for k, v in pr.items()}
    bt.set_dynamics(pr["A"], pr["B"], pr["f"]); bt.set_cost(pr["Q"], pr["R"], pr["H"], pr["q"], pr["r"])
    bt.set_host_batch(0) if bt.plan != altro_amd.PLAN_GENERIC else None
    bt.set_initial_state(np.tile(pr["x0"][:64], (batch // 64, 1)))
    for _ in range(3):
        bt.sweep()
    bt.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        bt.sweep()
    bt.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    used = bt.plan
    bt.close()
    return ms, used


def main():
    N, batch = 128, 4096
    print("# sweep (backward + forward) ms, N = %d, batch = %d, fp64, random LTV problems; plan AUTO vs plan GENERIC" % (N, batch))
    print("%-8s %-8s %10s %10s %8s" % ("(n, m)", "AUTO ->", "AUTO ms", "GENERIC ms", "ratio"))
    for (n, m) in [(12, 4), (12, 3), (11, 4), (10, 4), (8, 2), (7, 3), (6, 3), (6, 2), (5, 3), (5, 1), (4, 2), (4, 3), (3, 2), (2, 1), (1, 1)]:
        a, used = sweep_ms(N, n, m, batch, altro_amd.PLAN_AUTO)
        g, _ = sweep_ms(N, n, m, batch, altro_amd.PLAN_GENERIC)
        print("%-8s %-8s %10.3f %10.3f %8.1f" % ("(%d, %d)" % (n, m), PLAN[used], a, g, g / a))


if __name__ == "__main__":
    main()
