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
