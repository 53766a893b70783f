"""Affine line-search trials (plan MFMA16, dynamics given as data, fp64: kernels/ilqr_merit2_dpp.hip, AFF).  With x+ = A x + B u + f the
closed-loop rollout of SolverImpl::MeritFunction (solver.cpp:273-355) is affine in the step: x_k(alpha) = x_k(0) + alpha s_k with
s_k = dx_k / dalpha, which the sweep's phi(0) evaluation carries anyway.  The rounds after the first step evaluate every knot point
independently from that pair -- a chunk of 16 per wave instead of the horizon -- and add the shares up.

* one evaluation: phi, phi', the candidate trajectory against the rollout form (ALTRO_HIP_AFFINE=0) -- equal to rounding;
* whole constrained solves against the rollout form: same statuses / iterations / dual updates on these problems, trajectories 1e-9;
  and against the oracle like tests/test_gpu_al.py (that file runs with the affine rounds on too);
* the rounds are the same bits whether the sweep's head is the two-trial pass or the one-evaluation sequence, and whether trials are
  evaluated speculatively or one per launch (tests/test_gpu_speculation.py, tests/test_gpu_merit2.py::test_dual_evaluation_...)."""
import os

import numpy as np
import pytest

import altro_amd
from tests import problems

pytestmark = pytest.mark.gpu


def _solve(p, N, blocks, affine, forms=0, **kw):
    old = os.environ.get("ALTRO_HIP_AFFINE")
    os.environ["ALTRO_HIP_AFFINE"] = "1" if affine else "0"
    try:
        batch = p["x0"].shape[0]
        bt = altro_amd.Batch(N, 12, 4, batch)
        bt.set_forms(forms)
        bt.set_dynamics(p["A"], p["B"], p["f"])
        if "Q" in p:
            bt.set_quadratic_cost(p["Q"], p["R"], p["H"], p["q"], p["r"], p["c"])
        else:
            bt.set_tracking_cost(p["Qd"], p["Rd"], p["xref"], p["uref"])
        for (k0, k1, cone, G, g) in blocks:
            bt.add_linear_constraint(k0, k1, cone, G, g)
        bt.set_initial_state(p["x0"]); bt.set_input_guess(p["u0"])
        res = bt.ilqr_solve(**kw)
        x, u = bt.get_nominal()
        return dict(res, x=x, u=u)
    finally:
        if old is None:
            del os.environ["ALTRO_HIP_AFFINE"]
        else:
            os.environ["ALTRO_HIP_AFFINE"] = old


@pytest.mark.parametrize("dense", [False, True])
@pytest.mark.parametrize("batch,N,backtracking", [(41, 24, False), (130, 37, True), (7, 256, False), (3, 15, False)])
def test_constrained_solves_with_affine_rounds_equal_the_rollout_rounds(batch, N, backtracking, dense):
    """Horizons that are not a multiple of the chunk (24, 37, 15), one that is (256), odd horizons, a wave with one problem."""
    p = problems.ilqr12x4_problem(batch, N, True)
    if dense:
        p.update(problems.quadratic_cost(batch, N, 12, 4))
    blocks = problems.ilqr12x4_constraint_blocks(N)
    kw = dict(iterations_max=40, penalty_initial=1.0, penalty_scaling=10.0, use_backtracking=backtracking)
    a = _solve(p, N, blocks, True, **kw)
    b = _solve(p, N, blocks, False, **kw)
    assert a["merit_launches"] > a["sweeps"]                 # rounds past the first step did run
    both = (a["status"] == 0) & (b["status"] == 0)           # (some of these random problems run out of iterations in either form)
    print("converged in both: %d of %d; statuses equal %d; iterations equal %d" % (
        both.sum(), batch, (a["status"] == b["status"]).sum(), (a["iterations"] == b["iterations"]).sum()))
    assert both.sum() >= 1
    assert (a["status"] != b["status"]).sum() <= max(1, batch // 20)
    assert (a["iterations"][both] != b["iterations"][both]).sum() <= max(1, batch // 20)
    same = both & (a["iterations"] == b["iterations"])
    np.testing.assert_allclose(a["x"][same], b["x"][same], rtol=1e-7, atol=1e-7)
    np.testing.assert_allclose(a["u"][same], b["u"][same], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(a["feasibility"][same], b["feasibility"][same], rtol=1e-4, atol=1e-10)


def test_unconstrained_and_soc_problems():
    N, batch = 31, 19
    p = problems.ilqr12x4_problem(batch, N, True)
    a = _solve(p, N, [], True, iterations_max=10)
    b = _solve(p, N, [], False, iterations_max=10)
    assert np.array_equal(a["status"], b["status"]) and np.array_equal(a["iterations"], b["iterations"])
    np.testing.assert_allclose(a["x"], b["x"], rtol=1e-10, atol=1e-10)
    w = 16
    Gs = np.zeros((4, w)); Gs[0, 12] = 1.0; Gs[1, 13] = 1.0; Gs[2, 14] = 1.0
    blocks = [(0, N - 1, altro_amd.CONE_SOC, Gs, np.array([0.0, 0.0, 0.0, -0.35]))]
    kw = dict(iterations_max=40, penalty_initial=1.0, penalty_scaling=10.0)
    a = _solve(p, N, blocks, True, **kw)
    b = _solve(p, N, blocks, False, **kw)
    both = (a["status"] == 0) & (b["status"] == 0)
    assert both.sum() >= batch - 1 and np.array_equal(a["iterations"][both], b["iterations"][both])
    np.testing.assert_allclose(a["x"][both], b["x"][both], rtol=1e-8, atol=1e-8)


@pytest.mark.parametrize("seed", [11, 12])
def test_affine_rounds_used_for_robust_decisions_only_are_the_rollout_form_bit_for_bit(seed):
    """ALTRO_HIP_FORM_AFFINE_EXACT + decision_margin: the affine values decide only where the decision does not hang
    on their last bits (the state machine run at the corners of the margin box: linesearch_sm.h, ls_feed_is_robust); a trial whose
    values the search would keep, and the step it ends on, are evaluated as rollouts.  Then every step length, candidate and decision
    is the rollout form's: statuses, iteration counts, dual updates AND trajectories equal bit for bit -- on random constrained
    problems of tools/fuzz_affine.py's kind, both line searches, non-converging problems included."""
    from tools.fuzz_dpp import blocks_for
    rng = np.random.default_rng(seed)
    total = 0
    for c in range(12):
        n, m = 12, 4
        N = int(rng.integers(3, 70)); batch = int(rng.integers(2, 60))
        p = problems.ilqr12x4_problem(batch, N, bool(rng.integers(0, 2)), n=n, m=m)
        blocks = blocks_for(rng, N, n, m)
        kw = dict(iterations_max=int(rng.integers(3, 40)), use_backtracking=bool(rng.integers(0, 2)), penalty_initial=1.0, penalty_scaling=10.0)
        a = _solve(p, N, blocks, True, forms=altro_amd.FORM_AFFINE_EXACT, decision_margin=1e-9, **kw)
        b = _solve(p, N, blocks, False, **kw)
        for key in ("status", "iterations", "dual_updates", "x", "u", "alpha", "phi", "stationarity", "feasibility"):
            assert np.array_equal(a[key], b[key]), (c, key)
        total += batch
    assert total > 100


def test_margin_guard_sends_borderline_trials_to_the_rollout_form():
    """decision_margin alone: trials whose turn is not robust are evaluated again as rollouts (more merit launches than the unguarded
    rounds take), everything else as in the unguarded form; the converged problems agree with the rollout form's to rounding."""
    N, batch = 37, 60
    p = problems.ilqr12x4_problem(batch, N, True)
    blocks = problems.ilqr12x4_constraint_blocks(N)
    kw = dict(iterations_max=40, penalty_initial=1.0, penalty_scaling=10.0)
    g = _solve(p, N, blocks, True, decision_margin=1e-9, **kw)
    a = _solve(p, N, blocks, True, **kw)
    b = _solve(p, N, blocks, False, **kw)
    assert g["merit_launches"] >= a["merit_launches"]
    both = (g["status"] == 0) & (b["status"] == 0)
    assert both.sum() >= batch // 2
    same = both & (g["iterations"] == b["iterations"])
    assert same.sum() >= both.sum() - max(1, batch // 20)
    np.testing.assert_allclose(g["x"][same], b["x"][same], rtol=1e-7, atol=1e-7)
