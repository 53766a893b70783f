"""MeritFunction in the row layout of kernels/ilqr_row32.hip (round 6): plan MFMA32's shapes (12 < n <= 31 or 4 < m <= 8, n + m <= 32),
one kernel per shape, two problems per wave, every product a chain of DPP broadcasts against coefficients staged in LDS, no barriers,
against generic_merit_kernel (one wave per problem, vectors through LDS: ALTRO_HIP_FORM_GENERIC_MERIT_LDS), which it replaces in
altro_hip_merit and in every solve on these shapes.  Same sums in the same order: held to 1e-13 relative here (the printout says
whether a shape was bit-identical); whole solves against the oracle like plan GENERIC's own tests (solver.cpp:273-355)."""
import numpy as np
import pytest

import altro_amd
from tests import problems
from tests.test_gpu_ilqr_generic import make_oracle

pytestmark = pytest.mark.gpu

SHAPES = [(13, 4), (16, 4), (14, 7), (17, 1), (20, 8), (24, 8), (28, 4), (31, 1), (12, 5), (5, 8), (12, 4), (3, 2), (1, 1), (29, 3)]


def build(p, N, n, m, batch, dense, forms, blocks=(), plan=altro_amd.PLAN_AUTO):
    bt = altro_amd.Batch(N, n, m, batch, plan=plan)
    bt.set_forms(forms)
    bt.set_dynamics(p["A"], p["B"], p["f"])
    if dense:
        bt.set_quadratic_cost(p["Q"], p["R"], p["H"], p["q"], p["r"], p["c"])
    else:
        bt.set_tracking_cost(p["Qd"], p["Rd"], p["xref"], p["uref"])
    bt.set_initial_state(p["x0"]); bt.set_input_guess(p["u0"])
    for (k0, k1, cone, G, g) in blocks:
        bt.add_linear_constraint(k0, k1, cone, G, g)
    return bt


def problem(batch, N, n, m, dense):
    p = problems.ilqr12x4_problem(batch, N, True, n=n, m=m)
    if dense:
        p.update(problems.quadratic_cost(batch, N, n, m))
    return p


def blocks_for(N, n, m, rows):
    """An input box, a dense orthant block of `rows` rows at 1 <= k < N, a pinned first input, a terminal state box."""
    w = n + m
    rng = np.random.default_rng(100 * n + m)
    Gu = np.zeros((2 * m, w)); Gu[:m, n:] = np.eye(m); Gu[m:, n:] = -np.eye(m)
    Gd = rng.normal(size=(rows, w)) * (rng.random((rows, w)) < 0.5); Gd[np.abs(Gd).sum(axis=1) == 0, 0] = 1.0
    Ge = np.zeros((1, w)); Ge[0, n] = 1.0
    Gx = np.zeros((min(2 * n, 32), w)); h2 = Gx.shape[0] // 2
    Gx[:h2, :h2] = np.eye(h2); Gx[h2:2 * h2, :h2] = -np.eye(h2)
    return [(0, N - 1, altro_amd.CONE_INEQUALITY, Gu, np.full(2 * m, 0.4)),
            (1, N - 1, altro_amd.CONE_INEQUALITY, Gd, np.full(rows, 1.5)),
            (0, 0, altro_amd.CONE_EQUALITY, Ge, np.array([0.05])),
            (N, N, altro_amd.CONE_INEQUALITY, Gx, np.full(Gx.shape[0], 0.8))]


def evaluate(bt, batch):
    bt.open_loop_rollout(); bt.accept(); bt.expand(); bt.backward()
    assert (bt.get("status") == -1).all()
    out = {}
    for name, alphas, deriv in (("a", np.linspace(0.05, 1.1, batch), True), ("zero", np.zeros(batch), True), ("noderiv", np.full(batch, 0.5), False)):
        phi, dphi = bt.merit(alphas, derivative=deriv)
        out["phi_" + name] = phi.copy()
        if deriv:
            out["dphi_" + name] = dphi.copy()
        out["x_" + name] = bt.get("x").copy(); out["u_" + name] = bt.get("u").copy(); out["y_" + name] = bt.get("y").copy()
        if deriv:
            _, _, lx, lu = bt.get_expansion()
            out["lx_" + name] = lx.copy(); out["lu_" + name] = lu.copy()
            # Stationarity / Feasibility of that candidate (solver.cpp:207-231): row32_stationarity_kernel against generic_stationarity_kernel
            out["stat_" + name] = np.asarray(bt.stationarity(), dtype=np.float64).copy()
            out["feas_" + name] = bt.feasibility().copy()
    return out


@pytest.mark.parametrize("constrained", [False, True])
def test_row_layout_merit_equals_the_lds_form_on_every_kind_of_shape(constrained):
    N = 9
    exact = []
    for (n, m) in SHAPES:
        for dense in (False, True):
            batch = 5 if (n + m) % 2 else 6          # odd batches: the last wave's second half has no problem
            p = problem(batch, N, n, m, dense)
            blocks = blocks_for(N, n, m, min(32, 3 + (n * 7 + m) % 30)) if constrained else ()
            res = {}
            for name, forms in (("row", 0), ("lds", altro_amd.FORM_GENERIC_MERIT_LDS)):
                bt = build(p, N, n, m, batch, dense, forms, blocks, plan=altro_amd.PLAN_AUTO if (n > 12 or m > 4) else altro_amd.PLAN_GENERIC)
                assert bt.plan in (altro_amd.PLAN_MFMA32, altro_amd.PLAN_GENERIC)
                res[name] = evaluate(bt, batch)
                bt.close()
            same = True
            for key in res["row"]:
                a, b = res["row"][key], res["lds"][key]
                np.testing.assert_allclose(a, b, rtol=1e-13, atol=1e-13 * max(1.0, float(np.abs(b).max())), err_msg="%s (%d, %d) dense %s" % (key, n, m, dense))
                same = same and np.array_equal(a, b)
            exact.append(same)
    print("row-layout merit == LDS form bit for bit on %d of %d (shape, cost) cases%s" % (sum(exact), len(exact), " with constraint blocks" if constrained else ""))


def test_inactive_problems_and_odd_batches_store_nothing_of_their_own():
    """altro_hip_merit on a line-search round touches only the problems that search: here through a solve whose problems finish at
    different sweeps (the active mask changes every round) -- the row-layout kernel's halves shadow each other; results equal the LDS
    form's problem by problem."""
    N, n, m, batch = 20, 13, 4, 7
    p = problem(batch, N, n, m, False)
    p["x0"] = p["x0"] * np.linspace(0.2, 3.0, batch)[:, None]
    blocks = blocks_for(N, n, m, 6)
    out = {}
    for name, forms in (("row", 0), ("lds", altro_amd.FORM_GENERIC_MERIT_LDS)):
        bt = build(p, N, n, m, batch, False, forms, blocks)
        r = bt.ilqr_solve(iterations_max=50, penalty_initial=1.0, penalty_scaling=10.0)
        out[name] = (r["status"].copy(), r["iterations"].copy(), bt.get_nominal()[0].copy(), r["merit_launches"])
        bt.close()
    assert np.array_equal(out["row"][0], out["lds"][0]) and np.array_equal(out["row"][1], out["lds"][1])
    assert len(set(out["row"][1].tolist())) > 1
    np.testing.assert_allclose(out["row"][2], out["lds"][2], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("n,m,dense", [(13, 4, False), (20, 8, True), (28, 4, False)])
def test_whole_solves_on_the_row_layout_equal_the_oracle(n, m, dense):
    """Every problem: the LDS form's status and iteration count (the two kernels return the same values, so a solve cannot tell them
    apart); every problem the oracle solves: its status, iteration count and trajectory.  (A problem that converges nowhere may stop a
    sweep apart from the oracle on plan GENERIC's loop -- tests/soak/README.md -- whichever merit kernel runs.)"""
    N, batch = 12, 6
    p = problem(batch, N, n, m, dense)
    blocks = blocks_for(N, n, m, 5)[:3]
    blocks[0] = blocks[0][:4] + (np.full(2 * m, 0.8),)          # (a looser input box: most problems converge)
    out = {}
    for name, forms in (("row", 0), ("lds", altro_amd.FORM_GENERIC_MERIT_LDS)):
        bt = build(p, N, n, m, batch, dense, forms, blocks)
        assert bt.plan == altro_amd.PLAN_MFMA32
        res = bt.ilqr_solve(iterations_max=60, penalty_initial=1.0, penalty_scaling=10.0)
        x, u = bt.get_nominal()
        out[name] = (res["status"].copy(), res["iterations"].copy(), x.copy(), u.copy())
        bt.close()
    assert np.array_equal(out["row"][0], out["lds"][0]) and np.array_equal(out["row"][1], out["lds"][1])
    st, its, x, u = out["row"]
    nconv = 0
    for b in range(batch):
        s = make_oracle(p, b, N, n, m, dense)
        for (k0, k1, cone, G, g) in blocks:
            for k in range(k0, k1 + 1):
                s.add_linear_constraint(k, cone, G, g)
        s.L.oracle_ilqr_initialize(s.h)
        for k in range(N):
            s.L.oracle_ilqr_set_input(s.h, k, np.ascontiguousarray(p["u0"][b, k]))
        s.set_penalty(1.0, 10.0)
        s.L.oracle_ilqr_set_options(s.h, 60, 1e-4, 1e-4, 1e-8, 0)
        status, iters, _ = s.solve()
        if status != 0:
            assert st[b] != 0
            continue
        assert st[b] == status and its[b] == iters, (b, st[b], status, its[b], iters)
        nconv += 1
        np.testing.assert_allclose(x[b], s.get("x"), rtol=1e-7, atol=1e-7)
        np.testing.assert_allclose(u[b], s.get("u"), rtol=1e-6, atol=1e-6)
    assert nconv >= 1


@pytest.mark.parametrize("n,m,dense", [(13, 4, False), (17, 3, True), (24, 8, False), (31, 1, True), (5, 8, False)])
def test_unconstrained_solves_are_the_lds_forms_bit_for_bit(n, m, dense):
    """Without constraint blocks a solve on these shapes runs row32_rollout_init_kernel (rollout + CopyTrajectory + first expansion in
    one pass), the row-layout merit kernel and row32_stationarity_kernel where ALTRO_HIP_FORM_GENERIC_MERIT_LDS runs plan GENERIC's five
    kernels: the same values, so the same iterates -- nominal trajectory, gains, expansion and statistics array_equal -- and the
    oracle's iteration count."""
    N, batch = 11, 5
    p = problem(batch, N, n, m, dense)
    out = {}
    for name, forms in (("row", 0), ("lds", altro_amd.FORM_GENERIC_MERIT_LDS)):
        bt = build(p, N, n, m, batch, dense, forms)
        res = bt.ilqr_solve(iterations_max=20)
        x, u = bt.get_nominal()
        _, _, lx, lu = bt.get_expansion()
        out[name] = dict(status=res["status"].copy(), iterations=res["iterations"].copy(), x=x.copy(), u=u.copy(), K=bt.get("K").copy(), d=bt.get("d").copy(),
                         lx=lx.copy(), lu=lu.copy(), stat=res["stationarity"].copy(), cost=res["cost"].copy() if "cost" in res else np.zeros(1))
        bt.close()
    for key in out["row"]:
        assert np.array_equal(out["row"][key], out["lds"][key]), key
    assert (out["row"]["status"] == 0).all()
    s = make_oracle(p, 0, N, n, m, dense)
    s.L.oracle_ilqr_set_options(s.h, 20, 1e-4, 1e-4, 1e-8, 0)
    status, iters, _ = s.solve()
    assert status == 0 and iters == out["row"]["iterations"][0]
    np.testing.assert_allclose(out["row"]["x"][0], s.get("x"), rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("constrained", [False, True])
def test_long_horizons_walk_in_chunks(constrained):
    """N = 70: the stationarity kernel walks the horizon in chunks of 32 knot points (their maxima meet in row32_stat_reduce_kernel), the
    constrained expansion in chunks of 16 -- against the LDS forms: stationarity, feasibility and a whole solve."""
    N, n, m, batch = 70, 13, 4, 5
    p = problem(batch, N, n, m, False)
    blocks = blocks_for(N, n, m, 7) if constrained else ()
    out = {}
    for name, forms in (("row", 0), ("lds", altro_amd.FORM_GENERIC_MERIT_LDS)):
        bt = build(p, N, n, m, batch, False, forms, blocks)
        res = evaluate(bt, batch)
        r = bt.ilqr_solve(iterations_max=40, penalty_initial=1.0, penalty_scaling=10.0)
        res["status"] = r["status"].copy(); res["iterations"] = r["iterations"].copy(); res["x"] = bt.get_nominal()[0].copy()
        res["stationarity"] = r["stationarity"].copy()
        out[name] = res
        bt.close()
    for key in out["row"]:
        a, b = out["row"][key], out["lds"][key]
        if key in ("x",):
            np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-9, err_msg=key)
        elif key in ("status", "iterations"):
            assert np.array_equal(a, b), key
        else:
            np.testing.assert_allclose(a, b, rtol=1e-13, atol=1e-13 * max(1.0, float(np.abs(b).max())), err_msg=key)


def test_a_handle_created_as_plan_generic_runs_the_same_kernels():
    """The row-layout kernels serve plan MFMA32's shapes on plan GENERIC's arrays -- also on a handle created with ALTRO_HIP_PLAN_GENERIC
    (whose TVLQR sweeps stay plan GENERIC's own, bit-identical to the CPU path): evaluations and a constrained solve against the LDS form."""
    N, n, m, batch = 15, 14, 7, 4
    p = problem(batch, N, n, m, True)
    blocks = blocks_for(N, n, m, 9)
    out = {}
    for name, forms in (("row", 0), ("lds", altro_amd.FORM_GENERIC_MERIT_LDS)):
        bt = build(p, N, n, m, batch, True, forms, blocks, plan=altro_amd.PLAN_GENERIC)
        assert bt.plan == altro_amd.PLAN_GENERIC
        res = evaluate(bt, batch)
        r = bt.ilqr_solve(iterations_max=30, penalty_initial=1.0, penalty_scaling=10.0)
        res["status"] = r["status"].copy(); res["iterations"] = r["iterations"].copy(); res["xsol"] = bt.get_nominal()[0].copy()
        out[name] = res
        bt.close()
    for key in out["row"]:
        assert np.array_equal(out["row"][key], out["lds"][key]), key
