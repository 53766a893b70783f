"""Helper of tests/test_gpu_parity.py::test_device_pointer_mode (run as a script in a fresh interpreter, torch first)."""
import ctypes as C
import os
import sys

import torch  # noqa: E402  (FIRST: see the test's docstring)
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import altro_amd  # noqa: E402
from tests import problems  # noqa: E402


def host_run(pr, plan):
    N, n, m = pr["N"], pr["n"], pr["m"]
    bt = altro_amd.Batch(N, n, m, pr["A"].shape[0], plan=plan)
    bt.set_dynamics(pr["A"], pr["B"], pr["f"]); bt.set_cost(pr["Q"], pr["R"], pr["H"], pr["q"], pr["r"])
    bt.set_initial_state(pr["x0"]); bt.sweep()
    return {k: bt.get(k) for k in ("K", "d", "P", "p", "x", "u", "y")}, bt.plan


def main():
    N, batch = 17, 70
    for (n, m, plan) in [(12, 4, altro_amd.PLAN_MFMA16), (4, 2, altro_amd.PLAN_LANE), (5, 2, altro_amd.PLAN_GENERIC)]:
        pr = problems.random_ltv(batch, N, n, m)
        req = altro_amd.PLAN_AUTO if plan != altro_amd.PLAN_GENERIC else plan
        ref, got_plan = host_run(pr, req)
        assert got_plan == plan
        bt = altro_amd.Batch(N, n, m, batch, plan=req)
        dev = {k: torch.from_numpy(np.ascontiguousarray(pr[k])).cuda() for k in ("A", "B", "f", "Q", "R", "H", "q", "r", "x0")}
        p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
        torch.cuda.synchronize()    # the handle works on its own stream: the caller's arrays must be complete
        bt.set_pointer_mode(True)
        L, h = bt.L, bt.h
        assert L.altro_hip_set_dynamics(h, p(dev["A"]), p(dev["B"]), p(dev["f"]), 0, 0) == 0
        assert L.altro_hip_set_cost(h, p(dev["Q"]), p(dev["R"]), p(dev["H"]), p(dev["q"]), p(dev["r"]), 0, 0, 0) == 0
        assert L.altro_hip_set_initial_state(h, p(dev["x0"]), 0) == 0
        assert L.altro_hip_sweep(h, 0.0) == 0
        shapes = {"K": (batch, N, m * n), "d": (batch, N, m), "P": (batch, N + 1, n * n), "p": (batch, N + 1, n),
                  "x": (batch, N + 1, n), "u": (batch, N, m), "y": (batch, N + 1, n)}
        for k, shp in shapes.items():
            out = torch.full(shp, float("nan"), dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()    # (the fill runs on torch's stream)
            assert getattr(L, "altro_hip_get_" + k)(h, p(out)) == 0
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), ref[k]), (plan, k)
        if plan != altro_amd.PLAN_GENERIC:   # tracking-cost arithmetic happens on the host: refused in device mode
            try:
                bt.set_tracking_cost(np.ones((batch, N + 1, n)), np.ones((batch, N, m)), np.zeros((batch, N + 1, n)), np.zeros((batch, N, m)))
                raise SystemExit("set_tracking_cost accepted device-pointer mode")
            except altro_amd.AltroHipError:
                pass
        bt.set_pointer_mode(False)
        assert np.array_equal(bt.get("K"), ref["K"])
    print("device pointer mode OK")


if __name__ == "__main__":
    main()
