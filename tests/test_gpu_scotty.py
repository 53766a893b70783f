"""The reference's one saved end-to-end run on the device paths: test/bicycle_test.cpp:266-359 tracks
test/scotty.json for 200 receding-horizon steps and keeps solve_iters / states / inputs / tracking errors in
test/scotty_mpc.json (here tests/golden/scotty_mpc_expected.json, data only).  Rows a12 + f2 + f3 of SURVEY.md section 8 at once:
whole AL-iLQR solves with the steering INEQUALITY, warm-started through UpdateLinearCosts / SetInitialState / ShiftTrajectory.

* plan LANE, a batch of vehicles: copy 0 and copy 1 start where the reference does, the rest are perturbed;
* the C++ ALTROSolver (host callbacks, every backward sweep through tvlqr_BackwardPass on the GPU).

The oracle reproduces the file exactly on the CPU (tests/test_oracle_mpc.py)."""
import os
import tempfile
import time

import numpy as np
import pytest

import altro_amd
from tests import cpp_build
from tests import mpc_common as M
from tests import problems
from tests.test_gpu_mpc import make_hip

pytestmark = pytest.mark.gpu

NSIM = 200
# Device sin / cos / atan2 / tan differ from glibc's in the last ulp; a closed loop of 200 warm-started solves carries
# that through 627 sweeps.  Measured on the GPU (round 5): every one of the 200 iteration counts equal, |x - x_file| 3.2e-14,
# |u - u_file| 3.0e-13.  Asserted: all counts equal, 1e-10.
TOL_X, TOL_U = 1e-10, 1e-10


def run_batch(bt, u0, x_ref, x0s, nsim):
    batch = x0s.shape[0]
    xs = x0s.copy()
    iters, us, xt = [], [], []
    run_batch.solve_s = 0.0
    for it in range(nsim):
        t0 = time.perf_counter()
        res = bt.ilqr_solve(iterations_max=80, use_backtracking=True)
        run_batch.solve_s += time.perf_counter() - t0
        assert (res["status"] == 0).all(), (it, np.flatnonzero(res["status"] != 0))
        _, u = bt.get_knot(0)
        xs = np.stack([M.plant(xs[b], u[b]) for b in range(batch)])
        iters.append(res["iterations"].copy()); us.append(u.copy()); xt.append(xs.copy())
        q, c = M.linear_costs(x_ref, it + 1, u0)
        bt.update_linear_costs(q[None], None, c[None], 0, M.N, batch_stride_zero=True)
        bt.set_initial_state(xs)
        bt.shift_trajectory()
    return np.array(iters), np.array(us), np.array(xt)


def test_plan_lane_reproduces_the_references_saved_mpc_run():
    x_ref, u_ref, exp = problems.scotty()
    batch = 64
    off = problems.uniform01((batch, 4), 57) - 0.5
    x0s = x_ref[0] + off * np.array([0.4, 0.4, 0.05, 0.0])
    x0s[0] = x_ref[0]; x0s[1] = x_ref[0]
    bt, u0 = make_hip(x_ref, u_ref, x0s)
    iters, us, xt = run_batch(bt, u0, x_ref, x0s, NSIM)
    # copy 1 is copy 0: a problem's result does not depend on its place in the batch
    assert np.array_equal(iters[:, 0], iters[:, 1]) and np.array_equal(xt[:, 0], xt[:, 1]) and np.array_equal(us[:, 0], us[:, 1])
    ex, eu = np.abs(xt[:, 0] - exp["state_trajectory"][1:]).max(), np.abs(us[:, 0] - exp["input_trajectory"]).max()
    same = int((iters[:, 0] == exp["solve_iters"]).sum())
    err = np.linalg.norm(xt[:, 0] - x_ref[1:NSIM + 1], axis=1)
    print("scotty on plan LANE: solve_iters equal %d / %d, max |x - x_file| %.3g, max |u - u_file| %.3g, max tracking error %.6f; "
          "%d vehicles x %d steps: %.1f MPC steps/s per vehicle, %.0f vehicle-steps/s (solve calls only)"
          % (same, NSIM, ex, eu, err.max(), batch, NSIM, NSIM / run_batch.solve_s, batch * NSIM / run_batch.solve_s))
    assert same == NSIM, np.flatnonzero(iters[:, 0] != exp["solve_iters"])
    assert ex < TOL_X and eu < TOL_U
    np.testing.assert_allclose(err, exp["tracking_error"], rtol=0, atol=2 * TOL_X)
    assert abs(err.max() - 1.9336227439800688) < 2 * TOL_X
    # the perturbed vehicles: every solve succeeded (asserted in run_batch) and they close in on the same path
    e_end = np.linalg.norm(xt[-1] - x_ref[NSIM], axis=1)
    assert e_end.max() < err[-1] + 0.2
    # three of the perturbed copies against the per-vehicle oracle loop, first 25 steps
    for b in (2, 33, 63):
        steps = M.oracle_mpc(x_ref, u_ref, x0s[b], 25)
        for it, (k_it, u, xn, status) in enumerate(steps):
            assert status == 0
            np.testing.assert_allclose(us[it, b], u, rtol=0, atol=TOL_U)
            np.testing.assert_allclose(xt[it, b], xn, rtol=0, atol=TOL_X)


def test_cpp_altro_solver_reproduces_the_references_saved_mpc_run():
    x_ref, u_ref, exp = problems.scotty()
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "scotty.txt")
        with open(path, "w") as f:
            f.write("%d\n" % len(x_ref))
            for x, u in zip(x_ref, u_ref):
                f.write(" ".join(repr(float(v)) for v in list(x) + list(u)) + "\n")
        rc, out, errtxt = cpp_build.run("bicycle_mpc_test", args=[path, NSIM], timeout=900)
        # the same program with the three tvlqr_* entry points bound to the CPU oracle (tests/cpp/tvlqr_cpu_seam.cpp interposes
        # libaltro_hip.so's definitions): what the seam replaces, timed on this host beside it
        from oracle import oracle as _oracle
        _oracle.lib()
        odir = os.path.join(cpp_build.ROOT, "oracle", "_build")
        rc_cpu, out_cpu, err_cpu = cpp_build.run("bicycle_mpc_test", extra_sources=["tests/cpp/tvlqr_cpu_seam.cpp"],
                                                 extra_link=["-L" + odir, "-loracle", "-Wl,-rpath," + odir], args=[path, NSIM], timeout=900,
                                                 out_name="bicycle_mpc_test_cpu_seam")
    assert rc == 0 and out.strip().endswith("OK"), out[-2000:] + errtxt[-2000:]
    assert rc_cpu == 0 and out_cpu.strip().endswith("OK"), out_cpu[-2000:] + err_cpu[-2000:]
    rows_cpu = [l.split() for l in out_cpu.splitlines() if l.startswith("step ")]
    rate_cpu = [l for l in out_cpu.splitlines() if l.startswith("Average rate")][0]
    rows = [l.split() for l in out.splitlines() if l.startswith("step ")]
    assert len(rows) == NSIM
    iters = np.array([int(r[3]) for r in rows]); status = np.array([int(r[5]) for r in rows])
    us = np.array([[float(v) for v in r[7:9]] for r in rows]); xs = np.array([[float(v) for v in r[10:14]] for r in rows])
    err = np.array([float(r[15]) for r in rows])
    rate = [l for l in out.splitlines() if l.startswith("Average rate")][0]
    same = int((iters == exp["solve_iters"]).sum())
    ex, eu = np.abs(xs - exp["state_trajectory"][1:]).max(), np.abs(us - exp["input_trajectory"]).max()
    print("scotty through ALTROSolver: solve_iters equal %d / %d, max |x - x_file| %.3g, max |u - u_file| %.3g; %s through the GPU seam; "
          "%s with the backward sweeps on this host's CPU (oracle port, one thread)" % (same, NSIM, ex, eu, rate, rate_cpu))
    # the seam changes where the sweep runs, not what the solver does: same iteration counts, same inputs
    assert [r[3] for r in rows_cpu] == [r[3] for r in rows]
    us_cpu = np.array([[float(v) for v in r[7:9]] for r in rows_cpu])
    assert np.abs(us_cpu - us).max() < 1e-9
    assert (status == 0).all()
    # host callbacks (glibc trigonometry) + the device backward sweep in the oracle's operation order
    assert same == NSIM, np.flatnonzero(iters != exp["solve_iters"])
    assert ex < 1e-9 and eu < 1e-9
    np.testing.assert_allclose(err, exp["tracking_error"], rtol=0, atol=1e-9)


def test_cpp_altro_solver_with_a_device_model_reproduces_the_references_saved_mpc_run():
    """The same program with ALTROSolver::SetDeviceModel(ALTRO_HIP_MODEL_BICYCLE) + SetLinearConstraint in the place of the callback
    pairs (tests/cpp/bicycle_mpc_test.cpp, -DDEVICE_MODEL): every Solve() of the 200-step run is altro_hip_ilqr_solve on a resident batch of
    one, warm-started through the class's UpdateLinearCosts / SetInitialState / ShiftTrajectory -- the file's iteration counts, states
    and inputs."""
    x_ref, u_ref, exp = problems.scotty()
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "scotty.txt")
        with open(path, "w") as f:
            f.write("%d\n" % len(x_ref))
            for x, u in zip(x_ref, u_ref):
                f.write(" ".join(repr(float(v)) for v in list(x) + list(u)) + "\n")
        rc, out, errtxt = cpp_build.run("bicycle_mpc_test", args=[path, NSIM], timeout=900, defines=["DEVICE_MODEL"], out_name="bicycle_mpc_test_device_model")
    assert rc == 0 and out.strip().endswith("OK"), out[-2000:] + errtxt[-2000:]
    rows = [l.split() for l in out.splitlines() if l.startswith("step ")]
    assert len(rows) == NSIM
    iters = np.array([int(r[3]) for r in rows]); status = np.array([int(r[5]) for r in rows])
    us = np.array([[float(v) for v in r[7:9]] for r in rows]); xs = np.array([[float(v) for v in r[10:14]] for r in rows])
    rate = [l for l in out.splitlines() if l.startswith("Average rate")][0]
    same = int((iters == exp["solve_iters"]).sum())
    ex, eu = np.abs(xs - exp["state_trajectory"][1:]).max(), np.abs(us - exp["input_trajectory"]).max()
    print("scotty through ALTROSolver::SetDeviceModel: solve_iters equal %d / %d, max |x - x_file| %.3g, max |u - u_file| %.3g; %s" % (same, NSIM, ex, eu, rate))
    assert (status == 0).all()
    assert same == NSIM, np.flatnonzero(iters != exp["solve_iters"])
    assert ex < 1e-8 and eu < 1e-8
