"""CPU check of the oracle's MPC methods (UpdateLinearCosts / ShiftTrajectory / SetInitialState,
altro_solver.cpp:177-190, :266-293) in the caller pattern of test/bicycle_test.cpp:266-337."""
import numpy as np

from tests import mpc_common as M
from tests import problems


def test_oracle_bicycle_mpc_tracks_reference():
    nsim = 12
    x_ref, u_ref = problems.bicycle_reference(M.N + nsim + 1)
    x0 = x_ref[0] + np.array([0.05, -0.05, 0.01, 0.0])
    steps = M.oracle_mpc(x_ref, u_ref, x0, nsim)
    assert all(st[3] == 0 for st in steps)                     # every solve: SolveStatus::Success (:305)
    err = [np.linalg.norm(st[2] - x_ref[i + 1]) for i, st in enumerate(steps)]
    assert err[-1] < err[0] and err[-1] < 0.05                 # converges onto the reference
    assert steps[0][0] >= steps[-1][0]                         # warm-started solves get cheaper
    assert max(abs(st[2][3]) for st in steps) <= M.DELTA_MAX + 1e-4


def test_oracle_shift_matches_reference_semantics():
    """altro_solver.cpp:283-293 copies k+1 -> k in increasing k: x_[N] and u_[N-1] keep their values."""
    x_ref, u_ref = problems.bicycle_reference(M.N + 2)
    s, u0 = M.make_oracle(x_ref, u_ref, x_ref[0])
    for k in range(M.N):
        s.L.oracle_ilqr_set_input(s.h, k, np.array([float(k), -float(k)]))
    s.L.oracle_ilqr_shift_trajectory(s.h)
    xc, uc = s.get("x_cand"), s.get("u_cand")
    assert np.array_equal(xc[:M.N], x_ref[1:M.N + 1]) and np.array_equal(xc[M.N], x_ref[M.N])
    assert np.array_equal(uc[:M.N - 1, 0], np.arange(1, M.N)) and uc[M.N - 1, 0] == M.N - 1


def test_oracle_reproduces_the_references_saved_mpc_run():
    """The one end-to-end run the reference keeps: test/bicycle_test.cpp:266-359 tracks test/scotty.json for 200 receding-
    horizon steps (N = 30, Qd = 1e-2, Rd = 1e-3, steering +-60 deg INEQUALITY at k = 0..N, iterations_max = 80, backtracking,
    u0 = (u_ref[0][0], 0), SetState(x_ref)) and saves solve_iters / state / input / tracking_error to test/scotty_mpc.json
    (here: tests/golden/scotty_mpc_expected.json, data only).  The oracle on the same scenario: ALL 200 iteration counts
    equal, every state and input within 1e-12 (measured 2e-14 / 2e-13: libm's last ulp), tracking error within 1e-12, its
    maximum 1.9336.  h = 0.1f: the saved file's tf = Nsim * h = 20.0 says so (scotty.json's N = 501 is the point count;
    with h = 50 / 501 nine of the 200 counts differ -- see make_scotty_fixtures.py)."""
    x_ref, u_ref, exp = problems.scotty()
    assert M.H == np.float32(0.1)
    steps = M.oracle_mpc(x_ref, u_ref, x_ref[0], 200)
    assert all(st[3] == 0 for st in steps)                                     # EXPECT_EQ(status, Success) :305
    iters = np.array([st[0] for st in steps])
    assert np.array_equal(iters, exp["solve_iters"]), np.flatnonzero(iters != exp["solve_iters"])
    xs = np.array([st[2] for st in steps]); us = np.array([st[1] for st in steps])
    np.testing.assert_allclose(xs, exp["state_trajectory"][1:], rtol=0, atol=1e-12)
    np.testing.assert_allclose(us, exp["input_trajectory"], rtol=0, atol=1e-12)
    err = np.linalg.norm(xs - x_ref[1:201], axis=1)
    np.testing.assert_allclose(err, exp["tracking_error"], rtol=0, atol=1e-12)
    assert abs(err.max() - 1.9336227439800688) < 1e-12
    assert int(iters.sum()) == 627 and int(iters.max()) == 15
