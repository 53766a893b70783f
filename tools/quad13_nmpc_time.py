#!/usr/bin/env python3
"""What a batched NMPC step of 13-state quaternion quadrotors costs on plan MFMA32 (device model in the row-layout loop kernels of
kernels/ilqr_row32.hip): tests/test_gpu_generic_model.py's case at a given batch, warm receding-horizon steps timed on the host clock.

    python tools/quad13_nmpc_time.py [batch] [steps] [--source] [--lds]

--source: the model handed over as HIP source (altro_hip_set_model_source: hiprtc); --lds: the wave-per-problem model kernels
(ALTRO_HIP_FORM_GENERIC_MERIT_LDS).
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import altro_amd  # noqa: E402
from tests.test_gpu_generic_model import H, N, QUADROTOR13_SRC, m, make_case, n  # noqa: E402

argv = [a for a in sys.argv if not a.startswith("--")]
batch = int(argv[1]) if len(argv) > 1 else 1024
steps = int(argv[2]) if len(argv) > 2 else 6
c = make_case(batch, seed=7)
bt = altro_amd.Batch(N, n, m, batch)
if "--lds" in sys.argv:
    bt.set_forms(altro_amd.FORM_GENERIC_MERIT_LDS)
if "--source" in sys.argv:
    bt.set_model_source(QUADROTOR13_SRC, H)
else:
    bt.set_model(altro_amd.MODEL_QUADROTOR13, H)
print("model %s, %s model kernels" % ("from source (hiprtc)" if "--source" in sys.argv else "compiled in", "row-layout" if bt.model_row_layout() else "wave-per-problem"))
bt.set_tracking_cost(np.stack([c["Qd"], c["Qfd"]]), c["Rd"][None], np.stack([c["xref"], c["xref"]]), c["uref"][None], k_stride_zero=True, batch_stride_zero=True)
bt.set_initial_state(c["x0"])
bt.set_input_guess(c["u0"][None, None], k_stride_zero=True, batch_stride_zero=True)
t0 = time.perf_counter(); res = bt.ilqr_solve(iterations_max=40, tol_stationarity=1e-3); bt.synchronize()
print("first solve: %.2f ms, %d sweeps, %d merit launches, %d of %d converged" % ((time.perf_counter() - t0) * 1e3, res["sweeps"], res["merit_launches"], int((res["status"] == 0).sum()), batch))
ts = []
for step in range(steps):
    x1, _ = bt.get_knot(1)
    bt.set_initial_state(x1)
    bt.shift_trajectory()
    bt.synchronize(); t0 = time.perf_counter()
    res = bt.ilqr_solve(iterations_max=40, tol_stationarity=1e-3); bt.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
    print("step %d: %.2f ms, %d sweeps, %d merit launches, mean iterations %.2f" % (step, ts[-1], res["sweeps"], res["merit_launches"], float(res["iterations"].mean())))
print("(n, m) = (%d, %d), N = %d, %d vehicles: median warm step %.2f ms" % (n, m, N, batch, sorted(ts)[len(ts) // 2]))
