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

PLAN = {1: "GENERIC", 2: "MFMA16", 3: "LANE", 4: "MFMA32"}


def sweep_ms(N, n, m, batch, plan, flags=0):
    pr = problems.random_ltv(64, N, n, m)
    bt = altro_amd.Batch(N, n, m, batch, plan=plan, flags=flags)
    generic_arrays = bt.plan in (altro_amd.PLAN_GENERIC, altro_amd.PLAN_MFMA32)    # (no host-batch tiling on plan GENERIC's arrays)
    bt.set_host_batch(64) if not generic_arrays else None
    if generic_arrays:
        pr = {k: (np.tile(v, (batch // 64,) + (1,) * (v.ndim - 1)) if isinstance(v, np.ndarray) else v) for k, v in pr.items()}
    bt.set_dynamics(pr["A"], pr["B"], pr["f"]); bt.set_cost(pr["Q"], pr["R"], pr["H"], pr["q"], pr["r"])
    bt.set_host_batch(0) if not generic_arrays else None
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
    N = 128
    batches = [int(b) for b in sys.argv[1:]] or [4096]
    for batch in batches:
        print("# sweep (backward + forward) ms, N = %d, batch = %d, fp64, random LTV problems; every plan that takes the shape" % (N, batch))
        print("%-8s %-8s %10s %10s %10s %10s %10s %8s" % ("(n, m)", "AUTO ->", "AUTO ms", "LANE ms", "MFMA16 ms", "GENERIC ms", "GEN+MC ms", "worst/AUTO"))
        shapes = [(31, 1), (30, 2), (28, 4), (26, 6), (24, 8), (24, 4), (22, 6), (20, 8), (20, 4), (18, 6), (17, 4), (16, 8), (16, 4), (15, 4), (14, 7), (14, 4), (13, 5), (13, 4), (13, 1),
                  (12, 8), (12, 5), (10, 6), (8, 8), (6, 6), (5, 5), (12, 4), (12, 3), (11, 4), (10, 4), (8, 2), (7, 3), (6, 3), (6, 2), (6, 1), (5, 3), (5, 2), (5, 1),
                  (4, 3), (4, 2), (4, 1), (3, 3), (3, 2), (2, 1), (1, 1)]
        for (n, m) in shapes:
            if batch > 8192 and (n > 12 or m > 4):
                continue
            a, used = sweep_ms(N, n, m, batch, altro_amd.PLAN_AUTO)
            row = {}
            for name, plan, ok in (("LANE", altro_amd.PLAN_LANE, n <= 6 and m <= 3), ("MFMA16", altro_amd.PLAN_MFMA16, n <= 12 and m <= 4),
                                   ("GENERIC", altro_amd.PLAN_GENERIC, batch <= 8192)):
                row[name] = sweep_ms(N, n, m, batch, plan)[0] if ok else None
            # plan GENERIC with its products on the matrix cores (ALTRO_HIP_GENERIC_MATRIX_CORES: equal to rounding, not bit for bit; opt-in)
            mc = sweep_ms(N, n, m, batch, altro_amd.PLAN_GENERIC, altro_amd.GENERIC_MATRIX_CORES)[0] if (batch <= 8192 and (n > 12 or m > 4)) else None
            f = lambda v: ("%10.3f" % v) if v is not None else "%10s" % "-"
            best = min([v for v in row.values() if v is not None] + [a])
            print("%-8s %-8s %10.3f %s %s %s %s %8.2f" % ("(%d, %d)" % (n, m), PLAN[used], a, f(row["LANE"]), f(row["MFMA16"]), f(row["GENERIC"]), f(mc), a / best))


if __name__ == "__main__":
    main()
