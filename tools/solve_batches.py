#!/usr/bin/env python3
"""Fused vs launch-sequenced C3 solve (bicycle + steering bound, backtracking search) over the batch size: median wall ms.
    python tools/solve_batches.py [repeats]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.solve_ab import c2, c3      # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for name, mk in (("C3 bicycle, backtracking", lambda B: c3(True, 50, B)), ("C3 bicycle, cubic", lambda B: c3(False, 50, B))):
    for B in [int(v) for v in os.environ.get("SOLVE_BATCHES", "256,1024,2048,4096,8192,16384").split(",")]:
        bt, guess, opts = mk(B)
        med = {}
        for mode in ("1", "0"):
            os.environ["ALTRO_HIP_FUSED"] = mode
            ts = []
            for rep in range(R + 1):
                guess(); bt.synchronize()
                t0 = time.perf_counter(); res = bt.ilqr_solve(**opts); ts.append((time.perf_counter() - t0) * 1e3)
            med[mode] = float(np.median(ts[1:]))
        os.environ.pop("ALTRO_HIP_FUSED", None)
        print("%-26s batch %6d   fused %8.2f ms   sequenced %8.2f ms   (sweeps %d)" % (name, B, med["1"], med["0"], int(res["sweeps"])))
