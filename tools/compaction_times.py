#!/usr/bin/env python3
"""Straggler compaction of the one-launch solve (capi_solve.hip, run_fused): C3 whole solves with the single launch, the default
chunk policy and uniform chunks of n sweeps (the fused_sweeps = -n hook).    python tools/compaction_times.py [batches...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch          # noqa: E402
import altro_amd      # noqa: E402
import bench          # noqa: E402

batches = [int(a) for a in sys.argv[1:]] or [8192, 16384, 32768, 65536]
for batch in batches:
    bt, set_guess = bench.make_lane_batch("c3", batch, 0, 50, 0)
    ref = None
    for name, kw in (("single launch", dict(forms=altro_amd.FORM_NO_COMPACTION)), ("default policy", {}),
                     ("chunks of 2", dict(fused_sweeps=-2)), ("chunks of 8", dict(fused_sweeps=-8))):
        ts = []
        for rep in range(7):
            bt.reset_duals(1.0); set_guess(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = bt.ilqr_solve(iterations_max=80, use_backtracking=True, **kw)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        key = (res["status"].copy(), res["iterations"].copy(), bt.get_knot(50, want_u=False)[0].copy())
        if ref is None:
            ref = key
        same = all(np.array_equal(a, b) for a, b in zip(ref, key))
        it = res["iterations"]
        print("batch %6d  %-15s %8.2f ms (min of 6 after a warm-up; %s)  sweeps %d  still running after 4 / 10 / 20 sweeps: %d / %d / %d   bit-equal to the single launch: %s"
              % (batch, name, min(ts[1:]), " ".join("%.1f" % t for t in ts), res["sweeps"], (it > 4).sum(), (it > 10).sum(), (it > 20).sum(), same), flush=True)
    bt.close()
