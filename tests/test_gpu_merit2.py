"""Plan MFMA16: phi(0) and the line search's first step from ONE pass over the records (wave_merit_dpp_kernel<.., DUAL>), the
candidate's stationarity / feasibility from that same pass, the head of Solve as one pass (ROLLOUT_INIT) -- against the
one-evaluation-per-launch sequence they replace (ALTRO_HIP_MERIT2=0: wave_merit_kernel twice, wave_stationarity_kernel,
rollout + accept + expand), which tests/test_gpu_ilqr_mfma16.py pins to the oracle.  Same expressions in the same
order per trial: the results are compared bit for bit; where the compiler contracts a product-sum differently in the
two kernels a tolerance of 1e-12 relative would still hold, and the test says which it was.
Needs an MI355X."""
import os

import numpy as np
import pytest

import altro_amd
from tests import problems

pytestmark = pytest.mark.gpu

n, m = 12, 4


@pytest.fixture(autouse=True)
def rollout_trials(monkeypatch):
    """This module holds the kernel FORMS against each other bit for bit (two-trial pass vs sequence, row layout vs LDS form); the
    line-search rounds' affine evaluation (tests/test_gpu_affine.py) exists in the row layout only, so here every trial is a rollout."""
    monkeypatch.setenv("ALTRO_HIP_AFFINE", "0")


def _solve(p, N, blocks, dual, dtype=altro_amd.F64, **kw):
    batch = p["x0"].shape[0]
    bt = altro_amd.Batch(N, n, m, batch, dtype=dtype)
    bt.set_dynamics(p["A"], p["B"], p["f"])
    bt.set_tracking_cost(p["Qd"], p["Rd"], p["xref"], p["uref"])
    bt.set_initial_state(p["x0"])
    bt.set_input_guess(p["u0"])
    for (k0, k1, cone, G, g) in blocks:
        bt.add_linear_constraint(k0, k1, cone, G, g)
    old = os.environ.get("ALTRO_HIP_MERIT2")
    os.environ["ALTRO_HIP_MERIT2"] = "1" if dual else "0"
    try:
        res = bt.ilqr_solve(**kw)
    finally:
        if old is None:
            del os.environ["ALTRO_HIP_MERIT2"]
        else:
            os.environ["ALTRO_HIP_MERIT2"] = old
    out = dict(res)
    out["x"], out["u"] = bt.get_nominal()
    out["xc"], out["uc"], out["yc"] = bt.get("x"), bt.get("u"), bt.get("y")
    out["K"], out["d"] = bt.get("K"), bt.get("d")
    bt.close()
    return out


def _same(a, b, what):
    exact = True
    for k in ("status", "iterations", "dual_updates", "reg_retries"):
        assert np.array_equal(a[k], b[k]), (what, k)
    for k in ("phi", "stationarity", "feasibility", "alpha", "penalty", "x", "u", "xc", "uc", "yc", "K", "d"):
        if not np.array_equal(a[k], b[k]):
            exact = False
            scale = max(1.0, float(np.abs(b[k]).max()))
            assert np.abs(a[k] - b[k]).max() <= 1e-12 * scale, (what, k, float(np.abs(a[k] - b[k]).max()), scale)
    return exact


def _problem(batch, N, with_f):
    return problems.ilqr12x4_problem(batch, N, with_f)


@pytest.mark.parametrize("N,with_f", [(24, True), (25, False), (1, True), (2, True), (3, False)])
def test_dual_evaluation_matches_the_sequence_lq(N, with_f):
    """Unconstrained LQ solves, even and odd horizons (the two-deep record ring has a padding step when N is odd;
    the stationarity lag ends on the last LIVE step's buffers)."""
    p = _problem(67, N, with_f)
    a = _solve(p, N, [], True, iterations_max=6)
    b = _solve(p, N, [], False, iterations_max=6)
    assert (a["status"] == 0).all() and a["sweeps"] == b["sweeps"]
    assert a["merit_launches"] <= b["merit_launches"]          # one pass where the sequence takes two (or two trials in one launch)
    exact = _same(a, b, "lq N=%d" % N)
    print("dual vs sequence, N = %d: %s" % (N, "bit-identical" if exact else "within 1e-12"))


def test_dual_evaluation_matches_the_sequence_constrained():
    """Input bounds + state half-spaces + an equality block (the AL rows ride both trials), many sweeps, dual updates,
    line searches that go past the first step (those problems fall back to the single-step kernel and to
    wave_stationarity_kernel): same decisions, same numbers."""
    p, blocks = problems.ilqr12x4_problem(40, 24, True), problems.ilqr12x4_constraint_blocks(24)
    kw = dict(iterations_max=60, penalty_initial=1.0, penalty_scaling=10.0)
    a = _solve(p, 24, blocks, True, **kw)
    b = _solve(p, 24, blocks, False, **kw)
    assert (a["dual_updates"] > 0).all() and (a["iterations"] > 3).any()
    exact = _same(a, b, "constrained")
    print("dual vs sequence, constrained: %s" % ("bit-identical" if exact else "within 1e-12"))


def test_dual_evaluation_backtracking_and_early_out():
    """(a) a search that ends WITHOUT the first step: a problem already at its optimum has |phi'(0)| below the
    tolerance in its second sweep ... forced here by a huge tol_meritfun_gradient so every problem takes that path in
    its first sweep (alpha = 0, candidate = the alpha = 0 evaluation); (b) the backtracking search."""
    p = _problem(33, 24, True)
    a = _solve(p, 24, [], True, iterations_max=3, tol_meritfun_gradient=1e30)
    b = _solve(p, 24, [], False, iterations_max=3, tol_meritfun_gradient=1e30)
    assert (a["alpha"] == 0.0).all()
    assert _same(a, b, "early-out") in (True, False)
    a = _solve(p, 24, [], True, iterations_max=6, use_backtracking=True)
    b = _solve(p, 24, [], False, iterations_max=6, use_backtracking=True)
    assert (a["status"] == 0).all()
    _same(a, b, "backtracking")


def test_dual_evaluation_fp32_storage():
    """fp32 records: the stationarity comes from wave_stationarity_kernel (stored values are rounded ones), the rest is
    the dual pass.  Everything that goes through an fp32 store may differ by one fp32 rounding between the two paths (a
    1e-16 difference in the double decides which way a value rounds): same decisions, values within a few fp32 ulps."""
    p = _problem(48, 24, True)
    a = _solve(p, 24, [], True, dtype=altro_amd.F32, iterations_max=6)
    b = _solve(p, 24, [], False, dtype=altro_amd.F32, iterations_max=6)
    assert np.array_equal(a["status"], b["status"]) and np.array_equal(a["iterations"], b["iterations"])
    for k in ("phi", "stationarity", "x", "u"):
        scale = max(1.0, float(np.abs(b[k]).max()))
        assert np.abs(a[k] - b[k]).max() <= 2e-6 * scale, k


def _with_env(name, value, fn):
    old = os.environ.get(name)
    os.environ[name] = value
    try:
        return fn()
    finally:
        if old is None:
            del os.environ[name]
        else:
            os.environ[name] = old


@pytest.mark.parametrize("batch,N,with_f", [(67, 24, True), (1, 7, True), (2, 1, False), (130, 25, False), (64, 2, True)])
def test_two_trial_pass_against_the_lds_single_step_sequence(batch, N, with_f):
    """wave_merit_dpp_kernel<.., DUAL> (two problems per wave, two trials per problem) against the one-evaluation-per-launch sequence on
    the LDS form (ALTRO_HIP_MERIT2=0 and ALTRO_HIP_MERIT_DPP=0: wave_merit_kernel twice): odd batches (a wave with one problem), odd
    horizons (the padding step) and a batch where a third of the problems start at their optimum and drop out of the second sweep (a wave
    whose two problems are not both running).  (Up to round 4 the comparison form was the two-trial pass's own LDS twin,
    wave_merit2_kernel; it went with the A/B forms that lost -- DESIGN.md section 6.)"""
    p = _problem(batch, N, with_f)
    if batch >= 3 and not with_f:
        p["x0"][::3] = 0.0
    a = _solve(p, N, [], True, iterations_max=6)
    b = _with_env("ALTRO_HIP_MERIT_DPP", "0", lambda: _solve(p, N, [], False, iterations_max=6))
    assert (a["status"] == 0).all()
    _same(a, b, "two-trial vs LDS sequence")


def test_two_trial_pass_fp32_storage_against_the_lds_sequence():
    p = _problem(45, 24, True)
    a = _solve(p, 24, [], True, dtype=altro_amd.F32, iterations_max=6)
    b = _with_env("ALTRO_HIP_MERIT_DPP", "0", lambda: _solve(p, 24, [], False, dtype=altro_amd.F32, iterations_max=6))
    assert np.array_equal(a["status"], b["status"]) and np.array_equal(a["iterations"], b["iterations"])
    for k in ("phi", "stationarity", "x", "u"):
        scale = max(1.0, float(np.abs(b[k]).max()))
        assert np.abs(a[k] - b[k]).max() <= 2e-6 * scale, k


def _soc_and_terminal_blocks(N):
    """A second-order-cone block on the inputs at every k < N, an input-bound block beside it, and a terminal half-space block."""
    w = n + m
    Gs = np.zeros((4, w)); Gs[0, 12] = 1.0; Gs[1, 13] = 1.0; Gs[2, 14] = 1.0
    Gb = np.zeros((2 * m, w)); Gb[:m, n:] = np.eye(m); Gb[m:, n:] = -np.eye(m)
    Gt = np.zeros((3, w)); Gt[0, 0] = 1.0; Gt[1, 1] = -1.0; Gt[2, 5] = 1.0
    return [(0, N - 1, altro_amd.CONE_SOC, Gs, np.array([0.0, 0.0, 0.0, -0.35])),
            (0, N - 1, altro_amd.CONE_INEQUALITY, Gb, np.full(2 * m, 0.3)),
            (N, N, altro_amd.CONE_INEQUALITY, Gt, np.array([0.4, 0.4, 0.2]))]


@pytest.mark.parametrize("which,batch,N", [("bounds", 41, 24), ("bounds", 6, 5), ("soc", 23, 12)])
def test_two_trial_pass_with_constraint_blocks_against_the_lds_sequence(which, batch, N):
    """The constraint rows in the row layout (dpp_al_rows / dpp_al_col) of the two-trial pass against the LDS single-step sequence:
    inequality and equality blocks, two blocks per knot point, a second-order cone, a terminal block; whole solves with dual updates and
    searches that go past the first step."""
    p = problems.ilqr12x4_problem(batch, N, True)
    blocks = problems.ilqr12x4_constraint_blocks(N) if which == "bounds" else _soc_and_terminal_blocks(N)
    kw = dict(iterations_max=40, penalty_initial=1.0, penalty_scaling=10.0)
    a = _solve(p, N, blocks, True, **kw)
    b = _with_env("ALTRO_HIP_MERIT_DPP", "0", lambda: _solve(p, N, blocks, False, **kw))
    assert (a["dual_updates"] > 0).any()
    _same(a, b, "two-trial + constraint rows vs LDS sequence")


def _merit_direct(p, N, blocks, dtype=altro_amd.F64):
    """altro_hip_merit (IK_MERIT) at per-problem steps after one expansion + backward pass: phi, phi', candidate, gradient."""
    batch = p["x0"].shape[0]
    bt = altro_amd.Batch(N, n, m, batch, dtype=dtype)
    bt.set_dynamics(p["A"], p["B"], p["f"])
    bt.set_tracking_cost(p["Qd"], p["Rd"], p["xref"], p["uref"])
    bt.set_initial_state(p["x0"])
    bt.set_input_guess(p["u0"])
    for (k0, k1, cone, G, g) in blocks:
        bt.add_linear_constraint(k0, k1, cone, G, g)
    bt.open_loop_rollout(); bt.accept(); bt.expand(); bt.backward()
    alphas = np.linspace(0.0, 1.4, batch)
    phi, dphi = bt.merit(alphas)
    out = dict(phi=phi, dphi=dphi, x=bt.get("x"), u=bt.get("u"), y=bt.get("y"))
    out["stationarity"] = bt.stationarity()      # reads the gradient the evaluation left in the cost records
    bt.backward()
    out["d"] = bt.get("d")                       # ... and so does the next backward pass
    phi1, _ = bt.merit(0.6, derivative=False)
    out["phi_noderiv"], out["x2"] = phi1, bt.get("x")
    bt.close()
    return out


@pytest.mark.parametrize("which,batch,N,dtype", [("none", 35, 24, altro_amd.F64), ("none", 1, 5, altro_amd.F64), ("bounds", 22, 9, altro_amd.F64),
                                                 ("soc", 9, 12, altro_amd.F64), ("bounds", 12, 8, altro_amd.F32)])
def test_dpp_single_step_form_is_bit_identical_to_the_lds_form(which, batch, N, dtype):
    """wave_merit_dpp_kernel<.., DUAL = false> (the line-search rounds and altro_hip_merit; default) against wave_merit_kernel
    (ALTRO_HIP_MERIT_DPP=0): one evaluation per problem at its own step, with and without the derivative."""
    p = problems.ilqr12x4_problem(batch, N, True)
    blocks = [] if which == "none" else problems.ilqr12x4_constraint_blocks(N) if which == "bounds" else _soc_and_terminal_blocks(N)
    a = _with_env("ALTRO_HIP_MERIT_DPP", "2", lambda: _merit_direct(p, N, blocks, dtype))      # 2: also where the launcher would not
    b = _with_env("ALTRO_HIP_MERIT_DPP", "0", lambda: _merit_direct(p, N, blocks, dtype))
    for k in a:
        assert np.array_equal(a[k], b[k]), (k, float(np.abs(np.asarray(a[k], dtype=float) - np.asarray(b[k], dtype=float)).max()))
    assert np.isfinite(a["phi"]).all() and np.abs(a["dphi"]).max() > 0


@pytest.mark.parametrize("dual", [True, False])
@pytest.mark.parametrize("which,batch,N,kw", [("bounds", 41, 24, dict()), ("soc", 14, 12, dict()), ("bounds", 19, 10, dict(use_backtracking=True)),
                                              ("none", 30, 16, dict(use_backtracking=True))])
def test_dpp_line_search_rounds_are_bit_identical_to_the_lds_form(which, batch, N, kw, dual):
    """Whole solves whose searches go past the first step (speculative trials in the rounds, the spec_pre launch of the
    one-evaluation-per-launch sequence with dual = False): the same numbers with the rounds on either form."""
    p = problems.ilqr12x4_problem(batch, N, True)
    blocks = [] if which == "none" else problems.ilqr12x4_constraint_blocks(N) if which == "bounds" else _soc_and_terminal_blocks(N)
    opts = dict(iterations_max=40, penalty_initial=1.0, penalty_scaling=10.0, **kw)
    a = _with_env("ALTRO_HIP_MERIT_DPP", "2", lambda: _solve(p, N, blocks, dual, **opts))
    b = _with_env("ALTRO_HIP_MERIT_DPP", "0", lambda: _solve(p, N, blocks, dual, **opts))
    c = _solve(p, N, blocks, dual, **opts)                    # the default rule (DPP form where it is the faster one)
    if which != "none":
        assert a["merit_launches"] > 2 * a["sweeps"]        # the rounds did run
    for k in ("status", "iterations", "dual_updates", "phi", "stationarity", "feasibility", "alpha", "penalty", "x", "u", "xc", "uc", "yc", "K", "d"):
        assert np.array_equal(a[k], b[k]), (k, float(np.abs(np.asarray(a[k], dtype=float) - np.asarray(b[k], dtype=float)).max()))
        assert np.array_equal(c[k], b[k]), k


@pytest.mark.parametrize("which,batch,N,dtype", [("bounds", 41, 24, altro_amd.F64), ("bounds", 6, 5, altro_amd.F64), ("soc", 23, 12, altro_amd.F64),
                                                 ("soc", 3, 4, altro_amd.F64), ("bounds", 17, 8, altro_amd.F32)])
def test_dpp_expansion_is_bit_identical_to_the_lds_form(which, batch, N, dtype):
    """wave_expand_dpp_kernel (four (problem, knot point) pairs per wave; the Gauss-Newton block as DPP outer products) against
    wave_expand_kernel (ALTRO_HIP_EXPAND_DPP=0): first the backward pass right after one expansion (K, d read every entry of the
    cost records), then whole solves -- batches that are not multiples of four, problems that drop out, second-order cones with
    their curvature term, a terminal block."""
    p = problems.ilqr12x4_problem(batch, N, True)
    blocks = problems.ilqr12x4_constraint_blocks(N) if which == "bounds" else _soc_and_terminal_blocks(N)

    def one_expansion():
        bt = altro_amd.Batch(N, n, m, batch, dtype=dtype)
        bt.set_dynamics(p["A"], p["B"], p["f"])
        bt.set_tracking_cost(p["Qd"], p["Rd"], p["xref"], p["uref"])
        bt.set_initial_state(p["x0"])
        bt.set_input_guess(p["u0"])
        for (k0, k1, cone, G, g) in blocks:
            bt.add_linear_constraint(k0, k1, cone, G, g)
        bt.open_loop_rollout(); bt.accept(); bt.expand(); bt.backward()
        out = bt.get("K"), bt.get("d"), bt.get("P"), bt.get("p")
        bt.close()
        return out

    ea = _with_env("ALTRO_HIP_EXPAND_DPP", "1", one_expansion)
    eb = _with_env("ALTRO_HIP_EXPAND_DPP", "0", one_expansion)
    for x, y in zip(ea, eb):
        assert np.array_equal(x, y)
    kw = dict(iterations_max=40, penalty_initial=1.0, penalty_scaling=10.0)
    a = _with_env("ALTRO_HIP_EXPAND_DPP", "1", lambda: _solve(p, N, blocks, True, dtype=dtype, **kw))
    b = _with_env("ALTRO_HIP_EXPAND_DPP", "0", lambda: _solve(p, N, blocks, True, dtype=dtype, **kw))
    assert (a["dual_updates"] > 0).any()
    for k in ("status", "iterations", "dual_updates", "phi", "stationarity", "feasibility", "alpha", "penalty", "x", "u", "xc", "uc", "yc", "K", "d"):
        assert np.array_equal(a[k], b[k]), (k, float(np.abs(np.asarray(a[k], dtype=float) - np.asarray(b[k], dtype=float)).max()))


@pytest.mark.parametrize("which,batch,N,dtype", [("bounds", 41, 24, altro_amd.F64), ("soc", 23, 12, altro_amd.F64), ("soc", 3, 4, altro_amd.F64),
                                                 ("bounds", 17, 8, altro_amd.F32)])
def test_dpp_dual_update_and_feasibility_are_bit_identical_to_the_lds_form(which, batch, N, dtype):
    """wave_dual_update_dpp_kernel and wave_feasibility_dpp_kernel (four problems per wave) against wave_dual_update_kernel and
    the feasibility loop of wave_stationarity_kernel (ALTRO_HIP_ALROWS_DPP=0): whole solves, the duals included."""
    p = problems.ilqr12x4_problem(batch, N, True)
    blocks = problems.ilqr12x4_constraint_blocks(N) if which == "bounds" else _soc_and_terminal_blocks(N)
    kw = dict(iterations_max=40, penalty_initial=1.0, penalty_scaling=10.0)
    for dual in (True, False):       # (False: the one-evaluation-per-launch sequence, where every sweep runs IK_STATIONARITY)
        a = _with_env("ALTRO_HIP_ALROWS_DPP", "1", lambda: _solve(p, N, blocks, dual, dtype=dtype, **kw))
        b = _with_env("ALTRO_HIP_ALROWS_DPP", "0", lambda: _solve(p, N, blocks, dual, dtype=dtype, **kw))
        assert (a["dual_updates"] > 0).any()
        for k in ("status", "iterations", "dual_updates", "phi", "stationarity", "feasibility", "alpha", "penalty", "x", "u", "xc", "uc", "yc", "K", "d"):
            assert np.array_equal(a[k], b[k]), (dual, k)


@pytest.mark.parametrize("N", [5, 6, 7, 9])
def test_reported_feasibility_is_the_trajectorys_own(N):
    """AltroStats::primal_feasibility of a solve (written by the two-trial pass for the accepted step) against the constraint
    violation recomputed from the returned trajectory -- in particular for ODD horizons, where the sweep's padding step (the
    horizon is walked two knot points per trip) evaluates knot point N - 1 a second time on a point that is not on the
    trajectory: nothing of that step may reach the maximum."""
    batch = 24
    p = problems.ilqr12x4_problem(batch, N, True)
    blocks = problems.ilqr12x4_constraint_blocks(N)
    r = _solve(p, N, blocks, True, iterations_max=40, penalty_initial=1.0, penalty_scaling=10.0)
    x, u = r["x"], r["u"]
    viol = np.zeros(batch)
    for (k0, k1, cone, G, g) in blocks:
        for k in range(k0, k1 + 1):
            w = np.concatenate([x[:, k], u[:, k] if k < N else np.zeros((batch, m))], axis=1)
            c = w @ G.T - g
            v = np.abs(c) if cone == altro_amd.CONE_EQUALITY else np.maximum(c, 0.0)
            viol = np.maximum(viol, v.max(axis=1))
    np.testing.assert_allclose(r["feasibility"], viol, rtol=1e-9, atol=1e-12)
