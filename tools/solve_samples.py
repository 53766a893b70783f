#!/usr/bin/env python3
"""All samples (ms) of the fused C3 solve, in run order:  python tools/solve_samples.py [repeats] [fused 0|1]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.solve_ab import c3      # noqa: E402  (runs nothing on import: see the guard there)

R = int(sys.argv[1]) if len(sys.argv) > 1 else 15
os.environ["ALTRO_HIP_FUSED"] = sys.argv[2] if len(sys.argv) > 2 else "1"
bt, guess, opts = c3(True)
ts = []
for rep in range(R + 1):
    guess(); bt.synchronize()
    t0 = time.perf_counter(); bt.ilqr_solve(**opts); ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join("%.1f" % t for t in ts[1:]))
