"""The fused solve of plan LANE (kernels/ilqr_fused.hip: whole sweeps of SolverImpl::Solve per launch, no host in
between) against the launch-sequenced loop it replaces: the same device functions in the same order, so NOTHING may
differ -- statuses, iteration counts, step lengths, merit values, residuals, penalties, trajectories and duals are
compared bit for bit with ALTRO_HIP_NO_FUSED=1 -- and neither may the hand-over in the middle of a solve (the first
sweeps fused, the stragglers in the sequenced loop with its speculative line-search steps).  Needs an MI355X."""
import os

import numpy as np
import pytest

import altro_amd
from tests import problems

pytestmark = pytest.mark.gpu

KEYS = ("status", "iterations", "stationarity", "alpha", "phi", "feasibility", "penalty", "dual_updates", "reg_retries")


def _solve(make, env, **opts):
    saved = {k: os.environ.get(k) for k in ("ALTRO_HIP_NO_FUSED", "ALTRO_HIP_FUSED", "ALTRO_HIP_FUSED_SWEEPS", "ALTRO_HIP_NO_SPECULATION")}
    for k in saved:
        os.environ.pop(k, None)
    os.environ.update(env)
    try:
        bt = make()
        res = bt.ilqr_solve(**opts)
        x, u = bt.get_nominal()
        return res, x, u, bt.get("x"), bt.get("K"), bt
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def _same(a, b):
    for key in KEYS:
        assert np.array_equal(np.asarray(a[0][key]), np.asarray(b[0][key])), key
    for i in (1, 2, 3, 4):
        assert np.array_equal(a[i], b[i]), i
    assert int(a[0]["sweeps"]) == int(b[0]["sweeps"])


def _pendulum(batch, N=60):
    def make():
        bt = altro_amd.Batch(N, 2, 1, batch)
        bt.set_model(altro_amd.MODEL_PENDULUM, np.float32(0.05))
        xf = np.array([np.pi, 0.0])
        bt.set_tracking_cost(np.array([[1e-2, 1e-2], [1.0, 1.0]]), np.array([[1e-3]]), np.stack([xf, xf]), np.zeros((1, 1)),
                             k_stride_zero=True, batch_stride_zero=True)
        x0 = np.zeros((batch, 2)); x0[:, 0] = problems.uniform01((batch,), 41) - 0.5
        bt.set_initial_state(x0)
        bt.set_input_guess(np.array([[[0.1]]]), k_stride_zero=True, batch_stride_zero=True)
        return bt
    return make


def _bicycle(batch, N=40, spread=0.6):
    n, m, h = 4, 2, np.float32(0.1)
    x_ref, u_ref = problems.bicycle_reference(N + 1)

    def make():
        bt = altro_amd.Batch(N, n, m, batch)
        bt.set_model(altro_amd.MODEL_BICYCLE, h)
        bt.set_tracking_cost(np.full((1, N + 1, n), 1e-2), np.full((1, N, m), 1e-3), x_ref[None, :N + 1], u_ref[None, :N],
                             batch_stride_zero=True)
        G = np.zeros((2, n + m)); G[0, 3] = 1.0; G[1, 3] = -1.0
        bt.add_linear_constraint(0, N, altro_amd.CONE_INEQUALITY, G, np.full(2, np.pi / 3))
        bt.set_initial_state(x_ref[0] + (problems.uniform01((batch, n), 23, 0) - 0.5) * spread)
        bt.set_input_guess(np.array([[[u_ref[0][0], 0.0]]]), k_stride_zero=True, batch_stride_zero=True)
        return bt
    return make


def _di_cones(kind, batch):
    """the reference's constrained double-integrator problems (goal / bounds / second-order cone), perturbed starts"""
    N, n, m = 10, 4, 2

    def make():
        bt = altro_amd.Batch(N, n, m, batch)
        bt.set_model(altro_amd.MODEL_DOUBLE_INTEGRATOR, np.float32(0.5))
        bt.set_tracking_cost(np.ones((2, n)), np.full((1, m), 1e-2), np.zeros((2, n)), np.zeros((1, m)), k_stride_zero=True,
                             batch_stride_zero=True)
        for (k0, k1, cone, G, g) in problems.di_constraint_blocks(kind, N, n, m, np.zeros(n), 1.0):
            bt.add_linear_constraint(k0, k1, cone, G, g)
        bt.set_initial_state(np.tile([1.0, 2.0, 0.0, 0.0], (batch, 1)) + 0.02 * np.arange(batch)[:, None])
        bt.set_input_guess(np.zeros((1, 1, m)), k_stride_zero=True, batch_stride_zero=True)
        return bt
    return make


@pytest.mark.parametrize("name,make,opts,duals", [
    ("pendulum, cubic line search", _pendulum(333), dict(iterations_max=30), []),
    ("bicycle + steering bound, backtracking", _bicycle(500), dict(iterations_max=40, use_backtracking=True),
     [(0, 0, 2), (20, 0, 2), (40, 0, 2)]),
    ("bicycle + steering bound, cubic", _bicycle(200), dict(iterations_max=40), [(7, 0, 2)]),
    ("double integrator, goal", _di_cones("goal", 70), dict(iterations_max=20, penalty_scaling=100.0), [(10, 0, 4)]),
    ("double integrator, bounds", _di_cones("bounds", 70), dict(iterations_max=20, penalty_scaling=100.0), [(10, 0, 4), (3, 0, 4)]),
    ("double integrator, second-order cone", _di_cones("soc", 70), dict(iterations_max=30, penalty_scaling=100.0),
     [(10, 0, 4), (3, 0, 3)]),
])
def test_fused_solve_is_bit_identical_to_the_sequenced_loop(name, make, opts, duals):
    seq = _solve(make, {"ALTRO_HIP_NO_FUSED": "1", "ALTRO_HIP_NO_SPECULATION": "1"}, **opts)
    fused = _solve(make, {"ALTRO_HIP_FUSED": "1", "ALTRO_HIP_FUSED_SWEEPS": "1000"}, **opts)        # the whole solve in one launch
    _same(seq, fused)
    assert int(fused[0]["merit_launches"]) == 0 and int(seq[0]["merit_launches"]) > 0
    hand = _solve(make, {"ALTRO_HIP_FUSED": "1", "ALTRO_HIP_FUSED_SWEEPS": "2"}, **opts)            # two sweeps fused, the rest sequenced + speculative
    _same(seq, hand)
    for (k, slot, p) in duals:   # the duals went the same way too
        z = seq[5].get_duals(k, slot, p)
        assert np.array_equal(z, fused[5].get_duals(k, slot, p)) and np.array_equal(z, hand[5].get_duals(k, slot, p))


def test_fused_regularisation_retry_matches_sequenced():
    """the retry extension inside the fused kernel: same retries, same gains"""
    N, n, m, batch = 10, 4, 2, 66

    def make():
        bt = altro_amd.Batch(N, n, m, batch)
        bt.set_model(altro_amd.MODEL_DOUBLE_INTEGRATOR, np.float32(0.5))
        bt.set_tracking_cost(np.ones((2, n)), np.full((1, m), -0.5), np.zeros((2, n)), np.zeros((1, m)), k_stride_zero=True,
                             batch_stride_zero=True)
        bt.set_initial_state(np.tile([1.0, 2.0, 0.0, 0.0], (batch, 1)) + 0.01 * np.arange(batch)[:, None])
        bt.set_input_guess(np.zeros((1, 1, m)), k_stride_zero=True, batch_stride_zero=True)
        return bt
    opts = dict(iterations_max=2, reg_retry_max=5, reg_min=0.01, reg_scale=10.0)
    seq = _solve(make, {"ALTRO_HIP_NO_FUSED": "1"}, **opts)
    fused = _solve(make, {"ALTRO_HIP_FUSED": "1"}, **opts)
    _same(seq, fused)
    assert (fused[0]["reg_retries"] >= 3).all() and (fused[5].get("status") == -1).all()


def test_fused_ragged_batch_and_short_horizon():
    for batch, N in ((1, 3), (65, 1), (130, 7)):
        seq = _solve(_pendulum(batch, N), {"ALTRO_HIP_NO_FUSED": "1"}, iterations_max=15)
        fused = _solve(_pendulum(batch, N), {"ALTRO_HIP_FUSED": "1"}, iterations_max=15)
        _same(seq, fused)


@pytest.mark.parametrize("batch", [2100, 4200])
def test_fused_workgroup_sizes(batch):
    """8 / 16 / 32 problems per workgroup (ilqr_fused_group: batch <= 2048 / <= 4096 / larger): the other tests run the
    8-problem kernels; these batches the 16- and 32-problem ones, ragged last workgroup included"""
    for make, opts in ((_pendulum(batch, 24), dict(iterations_max=12)),
                       (_bicycle(batch, 16), dict(iterations_max=10, use_backtracking=True))):
        seq = _solve(make, {"ALTRO_HIP_NO_FUSED": "1"}, **opts)
        fused = _solve(make, {"ALTRO_HIP_FUSED": "1"}, **opts)
        _same(seq, fused)
        assert int(fused[0]["merit_launches"]) == 0


def test_fused_f32_storage():
    """the same kernels instantiated for float: fused and sequenced still agree bit for bit"""
    N, n, m, batch = 30, 4, 2, 150
    x_ref, u_ref = problems.bicycle_reference(N + 1)

    def make():
        bt = altro_amd.Batch(N, n, m, batch, dtype=altro_amd.F32)
        bt.set_model(altro_amd.MODEL_BICYCLE, np.float32(0.1))
        bt.set_tracking_cost(np.full((1, N + 1, n), 1e-2), np.full((1, N, m), 1e-3), x_ref[None, :N + 1], u_ref[None, :N],
                             batch_stride_zero=True)
        G = np.zeros((2, n + m)); G[0, 3] = 1.0; G[1, 3] = -1.0
        bt.add_linear_constraint(0, N, altro_amd.CONE_INEQUALITY, G, np.full(2, np.pi / 3))
        bt.set_initial_state(x_ref[0] + (problems.uniform01((batch, n), 23, 0) - 0.5) * 0.5)
        bt.set_input_guess(np.array([[[u_ref[0][0], 0.0]]]), k_stride_zero=True, batch_stride_zero=True)
        return bt
    for opts in (dict(iterations_max=15, use_backtracking=True), dict(iterations_max=15)):
        seq = _solve(make, {"ALTRO_HIP_NO_FUSED": "1"}, **opts)
        fused = _solve(make, {"ALTRO_HIP_FUSED": "1"}, **opts)
        _same(seq, fused)
        assert int(fused[0]["merit_launches"]) == 0


def test_fused_six_state_double_integrator():
    """(6, 3): the shape whose fused kernel hipcc could not build before the merit evaluation was split"""
    N, n, m, batch = 25, 6, 3, 150

    def make(constrained):
        def mk():
            bt = altro_amd.Batch(N, n, m, batch)
            assert bt.plan == altro_amd.PLAN_MFMA16       # plan AUTO at 150 problems: the padded tile (cheaper sweeps) ...
            bt.set_model(altro_amd.MODEL_DOUBLE_INTEGRATOR, np.float32(0.2))
            assert bt.plan == altro_amd.PLAN_LANE         # ... until the LANE-only compiled-in model arrives on the empty handle
            bt.set_tracking_cost(np.ones((2, n)), np.full((1, m), 1e-2), np.zeros((2, n)), np.zeros((1, m)), k_stride_zero=True,
                                 batch_stride_zero=True)
            if constrained:
                G = np.zeros((2 * m, n + m)); G[:m, n:] = np.eye(m); G[m:, n:] = -np.eye(m)
                bt.add_linear_constraint(0, N - 1, altro_amd.CONE_INEQUALITY, G, np.full(2 * m, 0.6))
            bt.set_initial_state(2.0 * (problems.uniform01((batch, n), 57, 0) - 0.5))
            bt.set_input_guess(np.zeros((1, 1, m)), k_stride_zero=True, batch_stride_zero=True)
            return bt
        return mk
    for constrained, opts in ((False, dict(iterations_max=5)), (True, dict(iterations_max=25, penalty_scaling=10.0, use_backtracking=True)),
                              (True, dict(iterations_max=25, penalty_scaling=100.0))):
        seq = _solve(make(constrained), {"ALTRO_HIP_NO_FUSED": "1"}, **opts)
        fused = _solve(make(constrained), {"ALTRO_HIP_FUSED": "1"}, **opts)
        hand = _solve(make(constrained), {"ALTRO_HIP_FUSED": "1", "ALTRO_HIP_FUSED_SWEEPS": "2"}, **opts)
        _same(seq, fused)
        _same(seq, hand)
        assert int(fused[0]["merit_launches"]) == 0


# ---- straggler compaction (capi_solve.hip, run_fused): chunks of sweeps over the list of still-running problems --------------------
@pytest.mark.parametrize("name,make,opts,chunk", [
    # all three workgroup sizes of a LISTED launch: the list after the first chunk holds > 4096, > 2048, <= 2048 problems
    ("bicycle + steering bound, backtracking, 6000 problems, chunks of 1", _bicycle(6000, spread=1.2), dict(iterations_max=30, use_backtracking=True), 1),
    ("bicycle + steering bound, cubic, 3000 problems, chunks of 2", _bicycle(3000, spread=1.0), dict(iterations_max=30), 2),
    ("bicycle, chunks of 3, ragged batch", _bicycle(777), dict(iterations_max=40, use_backtracking=True), 3),
    ("pendulum (four lanes per problem, eight waves), chunks of 2", _pendulum(2500), dict(iterations_max=30), 2),
    ("double integrator, second-order cone, chunks of 1", _di_cones("soc", 70), dict(iterations_max=30, penalty_scaling=100.0), 1),
])
def test_listed_launches_are_the_single_launch_bit_for_bit(name, make, opts, chunk):
    one = _solve(make, {"ALTRO_HIP_FUSED": "1"}, forms=altro_amd.FORM_NO_COMPACTION, **opts)
    chunks = _solve(make, {"ALTRO_HIP_FUSED": "1", "ALTRO_HIP_FUSED_SWEEPS": str(-chunk)}, **opts)
    _same(one, chunks)
    assert int(chunks[0]["merit_launches"]) == 0     # (no hand-over to the sequenced loop happened)
    one[5].close(); chunks[5].close()


def test_listed_launches_with_the_regularisation_retry():
    N, n, m, batch = 10, 4, 2, 300

    def make():
        bt = altro_amd.Batch(N, n, m, batch)
        bt.set_model(altro_amd.MODEL_DOUBLE_INTEGRATOR, np.float32(0.5))
        R = np.where(np.arange(batch)[:, None, None] % 3 == 0, -0.5, 1e-2) * np.ones((batch, 1, m))   # every third problem needs the retry
        bt.set_tracking_cost(np.ones((batch, 2, n)), R, np.zeros((batch, 2, n)), np.zeros((batch, 1, m)), k_stride_zero=True)
        bt.set_initial_state(np.tile([1.0, 2.0, 0.0, 0.0], (batch, 1)) + 0.01 * np.arange(batch)[:, None])
        bt.set_input_guess(np.zeros((1, 1, m)), k_stride_zero=True, batch_stride_zero=True)
        return bt
    opts = dict(iterations_max=6, reg_retry_max=5, reg_min=0.01, reg_scale=10.0)
    one = _solve(make, {"ALTRO_HIP_FUSED": "1"}, forms=altro_amd.FORM_NO_COMPACTION, **opts)
    chunks = _solve(make, {"ALTRO_HIP_FUSED": "1", "ALTRO_HIP_FUSED_SWEEPS": "-1"}, **opts)
    _same(one, chunks)
    assert (chunks[0]["reg_retries"] >= 1).any()
    one[5].close(); chunks[5].close()


def test_a_batch_beyond_four_workgroups_per_compute_unit_compacts_by_itself():
    """40000 bicycles: 1250 workgroups of 32 -- the default solve runs its first sweeps, then the rest over the list of problems still
    running; the results are the single launch's."""
    make = _bicycle(40000, N=20, spread=1.0)
    opts = dict(iterations_max=40, use_backtracking=True)
    one = _solve(make, {}, forms=altro_amd.FORM_NO_COMPACTION, **opts)
    auto = _solve(make, {}, **opts)
    _same(one, auto)
    one[5].close(); auto[5].close()
