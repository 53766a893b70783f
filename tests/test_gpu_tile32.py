"""Plan MFMA32 (kernels/tvlqr_tile32.hip): the TVLQR pair on 2 x 2 tiles of v_mfma_f64_16x16x4 for the shapes one step past the
(12, 4) tile -- 12 < n <= 31, m <= 8, n + m <= 32, and n <= 12 with 4 < m <= 8 -- against the CPU oracle, through the C ABI.

Tolerances: K, d within 1e-8 absolute of the CPU path (the north star's statement; measured 1e-14), every block of K, d, P, p, x, u, y
within 1e-9 of its own scale, the Cholesky-failure index equal, K_k = Qux and d_k = -Qu left at a failed factorisation like
tvlqr.cpp:159-166 leaves them.  The plan sums in the matrix pipe's order, so it is not the bit-for-bit plan: that stays
ALTRO_HIP_PLAN_GENERIC, whose arrays (and iLQR loop kernels) this plan shares."""
import numpy as np
import pytest

import altro_amd
from oracle import oracle
from tests import problems
from tests.test_gpu_parity import relerr, run_hip, run_oracle

pytestmark = pytest.mark.gpu

ALL_SHAPES = [(n, m) for n in range(5, 32) for m in range(1, 9) if n + m <= 32 and not (n <= 12 and m <= 4)]
# one of every (terms-of-four, tile rows, tile columns, input chunks) class, the edges of the shape range, and the shapes the round's
# review names: the quaternion quadrotor (13, 4), the 7-joint arm (14, 7), (16, 4), (24, 8), (28, 4)
SHAPES = [(13, 4), (14, 7), (16, 4), (24, 8), (28, 4), (13, 1), (13, 3), (15, 1), (15, 8), (16, 1), (16, 8), (17, 3), (20, 8), (21, 5),
          (25, 7), (29, 3), (31, 1), (5, 5), (5, 8), (8, 8), (9, 7), (12, 5), (12, 8), (19, 2)]


def check(out, ref, tol=1e-9):
    assert (out["status"] == ref["status"]).all()
    assert np.abs(out["K"] - ref["K"]).max() < 1e-8 and np.abs(out["d"] - ref["d"]).max() < 1e-8   # the north star's statement
    for k in ("K", "d", "P", "p", "x", "u", "y"):
        assert relerr(out[k], ref[k]) < tol, (k, relerr(out[k], ref[k]))
    assert relerr(out["delta_V"], ref["dV"]) < tol


@pytest.mark.parametrize("n,m", SHAPES)
def test_tile32_vs_oracle(n, m):
    pr = problems.random_ltv(5, 20, n, m)
    out = run_hip(pr, altro_amd.PLAN_MFMA32)
    assert out["bt"].plan == altro_amd.PLAN_MFMA32
    check(out, run_oracle(pr))
    P = out["P"].reshape(5, 21, n, n)
    assert np.array_equal(P[:, :20], np.swapaxes(P[:, :20], 2, 3))     # the carried cost-to-go is the stored one: exactly symmetric


def test_tile32_every_shape():
    """All 156 shapes the plan takes (one kernel instantiation each): a short horizon against the oracle."""
    for n, m in ALL_SHAPES:
        pr = problems.random_ltv(2, 6, n, m)
        out = run_hip(pr, altro_amd.PLAN_MFMA32)
        ref = run_oracle(pr)
        for k in ("K", "d", "P", "p", "x", "u", "y"):
            assert relerr(out[k], ref[k]) < 1e-9, (n, m, k)
        out["bt"].close()


@pytest.mark.parametrize("n,m", [(13, 4), (20, 6)])
def test_tile32_without_affine_term_and_short_horizons(n, m):
    pr = problems.random_ltv(3, 9, n, m)
    pr["f"] = np.zeros_like(pr["f"])
    bt = altro_amd.Batch(9, n, m, 3, plan=altro_amd.PLAN_MFMA32)
    bt.set_dynamics(pr["A"], pr["B"], None); bt.set_cost(pr["Q"], pr["R"], pr["H"], pr["q"], pr["r"]); bt.set_initial_state(pr["x0"])
    bt.sweep()
    ref = run_oracle(pr)
    for k in ("K", "d", "P", "p", "x", "u", "y"):
        assert relerr(bt.get(k), ref[k]) < 1e-9, k
    for N in (1, 2):
        pr = problems.random_ltv(2, N, n, m)
        check(run_hip(pr, altro_amd.PLAN_MFMA32), run_oracle(pr))


@pytest.mark.parametrize("n,m", [(13, 4), (16, 4), (24, 8), (9, 6)])
def test_tile32_failed_factorisation(n, m):
    """Quu indefinite at chosen knot points: the sweep stops where the CPU path stops (`return k`, tvlqr.cpp:163), with K_k = Qux,
    d_k = -Qu there, everything from that knot point to the end equal to the oracle's, everything before it untouched (zero)."""
    N, batch = 12, 8
    pr = problems.random_ltv(batch, N, n, m)
    rng = np.random.default_rng(5)
    for b in range(0, batch, 2):
        k = int(rng.integers(1, N - 1))
        pr["R"][b, k].reshape(m, m)[:] -= 50.0 * np.eye(m)
    bt = altro_amd.Batch(N, n, m, batch, plan=altro_amd.PLAN_MFMA32)
    bt.set_dynamics(pr["A"], pr["B"], pr["f"]); bt.set_cost(pr["Q"], pr["R"], pr["H"], pr["q"], pr["r"])
    bt.backward(0.0)
    ref = oracle.backward_batch(pr["A"], pr["B"], pr["f"], pr["Q"], pr["R"], pr["H"], pr["q"], pr["r"], 0.0, False)
    st = bt.get("status")
    assert (st == ref["status"]).all() and (st[0::2] >= 0).all() and (st[1::2] == -1).all()
    K, d, P = bt.get("K"), bt.get("d"), bt.get("P")
    for b in range(batch):
        k0 = max(int(st[b]), 0)
        assert relerr(K[b:b + 1, k0:], ref["K"][b:b + 1, k0:]) < 1e-9 and relerr(d[b:b + 1, k0:], ref["d"][b:b + 1, k0:]) < 1e-9
        assert relerr(P[b:b + 1, k0 + 1:], ref["P"][b:b + 1, k0 + 1:]) < 1e-9
        if st[b] >= 0:
            assert not K[b, :k0].any() and not P[b, :k0 + 1].any()       # never written below the failing knot point
    # with enough regularisation the same batch factors: reg enters the pivots only (tvlqr.cpp:159-164), the cost-to-go update keeps
    # the unregularised Quu (tvlqr.cpp:174)
    bt.backward(60.0)
    ref = oracle.backward_batch(pr["A"], pr["B"], pr["f"], pr["Q"], pr["R"], pr["H"], pr["q"], pr["r"], 60.0, False)
    assert (bt.get("status") == -1).all() and (ref["status"] == -1).all()
    for k in ("K", "d", "P", "p"):
        assert relerr(bt.get(k), ref[k]) < 1e-9, k
    assert relerr(bt.get("delta_V"), ref["dV"]) < 1e-9


def test_tile32_is_what_auto_picks_and_generic_stays_exact():
    A, G, T = altro_amd.PLAN_AUTO, altro_amd.PLAN_GENERIC, altro_amd.PLAN_MFMA32
    for (n, m, kw, want) in [(13, 4, {}, T), (14, 7, {}, T), (28, 4, {}, T), (31, 1, {}, T), (8, 8, {}, T), (12, 5, {}, T),
                             (12, 4, {}, altro_amd.PLAN_MFMA16), (32, 1, {}, G), (20, 9, {}, G), (28, 5, {}, G), (4, 8, {}, G),
                             (13, 4, {"dtype": altro_amd.F32}, G), (13, 4, {"flags": altro_amd.STORE_QBLOCKS}, G),
                             (13, 4, {"flags": altro_amd.GENERIC_MATRIX_CORES}, G), (13, 4, {"plan": G}, G)]:
        bt = altro_amd.Batch(6, n, m, 4, **kw)
        assert bt.plan == want, (n, m, kw, bt.plan)
        bt.close()
    for n, m, dt in [(12, 4, altro_amd.F64), (33, 1, altro_amd.F64), (13, 9, altro_amd.F64), (13, 4, altro_amd.F32)]:
        with pytest.raises(altro_amd.AltroHipError, match="MFMA32"):
            altro_amd.Batch(6, n, m, 4, plan=T, dtype=dt)
    pr = problems.random_ltv(4, 10, 13, 4)
    ref = run_oracle(pr)
    out = run_hip(pr, G)
    for k in ("K", "d", "P", "p", "x", "u", "y"):
        assert np.array_equal(out[k], ref[k]), k          # plan GENERIC: the CPU path's bits, as before
    out = run_hip(pr, A)
    assert out["bt"].plan == T
    check(out, ref)


def test_tile32_diagonal_cost_runs_the_exact_kernel():
    """A diagonal cost keeps its diagonals packed in plan GENERIC's arrays (knotpoint_data.cpp:92-95): the backward sweep of such a
    handle is plan GENERIC's own kernel (bit for bit the oracle), the forward sweep the tile plan's."""
    n, m, N = 14, 3, 10
    pr = problems.random_ltv(4, N, n, m)
    rng = np.random.default_rng(3)
    pr["Qdiag"] = rng.uniform(0.5, 2.0, (4, N + 1, n)); pr["Rdiag"] = rng.uniform(0.1, 1.0, (4, N, m))
    out = run_hip(pr, altro_amd.PLAN_MFMA32, is_diag=True)
    ref = run_oracle(pr, is_diag=True)
    for k in ("K", "d", "P", "p"):
        assert np.array_equal(out[k], ref[k]), k
    for k in ("x", "u", "y"):
        assert relerr(out[k], ref[k]) < 1e-12, k


def test_tile32_full_size_13x4():
    """4096 quaternion-quadrotor-sized problems x 128 knot points (the shape-cliff table's size): every status -1, everything finite,
    P symmetric, and a seeded sample of problems against the oracle."""
    N, n, m, batch = 128, 13, 4, 4096
    base = problems.random_ltv(16, N, n, m)
    rep = lambda a: np.ascontiguousarray(np.tile(a, (batch // 16,) + (1,) * (a.ndim - 1)))
    pr = {k: (rep(v) if isinstance(v, np.ndarray) else v) for k, v in base.items()}
    rng = np.random.default_rng(11)
    pr["x0"] = rng.uniform(-1, 1, (batch, n))
    pr["q"] = pr["q"] + 0.01 * rng.standard_normal(pr["q"].shape)      # problems differ beyond the sixteen base ones
    out = run_hip(pr, altro_amd.PLAN_AUTO)
    assert out["bt"].plan == altro_amd.PLAN_MFMA32
    assert (out["status"] == -1).all()
    for k in ("K", "d", "P", "p", "x", "u", "y"):
        assert np.isfinite(out[k]).all(), k
    sample = [0, 17, 1023, 2048, 4095]
    sub = {k: (np.ascontiguousarray(v[sample]) if isinstance(v, np.ndarray) else v) for k, v in pr.items()}
    ref = run_oracle(sub)
    for k in ("K", "d", "P", "p", "x", "u", "y"):
        assert relerr(out[k][sample], ref[k]) < 1e-9, k


@pytest.mark.parametrize("n,m,dense", [(13, 4, False), (16, 5, True), (24, 8, False)])
def test_tile32_whole_lq_solves(n, m, dense):
    """The iLQR loop of plan GENERIC (kernels/ilqr_generic.hip) on a handle of this plan -- same arrays, the sweeps on the matrix
    cores: whole solves of an LQ problem end with the oracle's status and iteration count, trajectories 1e-9."""
    from tests.test_gpu_ilqr_generic import make_oracle
    N, batch = 18, 11
    p = problems.ilqr12x4_problem(batch, N, True, n=n, m=m)
    if dense:
        p.update(problems.quadratic_cost(batch, N, n, m))
    bt = altro_amd.Batch(N, n, m, batch)
    assert bt.plan == altro_amd.PLAN_MFMA32
    bt.set_dynamics(p["A"], p["B"], p["f"])
    if dense:
        bt.set_quadratic_cost(p["Q"], p["R"], p["H"], p["q"], p["r"], p["c"])
    else:
        bt.set_tracking_cost(p["Qd"], p["Rd"], p["xref"], p["uref"])
    bt.set_initial_state(p["x0"]); bt.set_input_guess(p["u0"])
    res = bt.ilqr_solve(iterations_max=10, tol_stationarity=1e-4)
    assert (res["status"] == 0).all() and (res["iterations"] <= 3).all()
    x, u = bt.get_nominal()
    for b in [0, 5, 10]:
        s = make_oracle(p, b, N, n, m, dense)
        s.L.oracle_ilqr_set_options(s.h, 10, 1e-4, 1e-4, 1e-8, 0)
        status, iters, log = s.solve()
        assert status == 0 and iters == res["iterations"][b]
        np.testing.assert_allclose(x[b], s.get("x"), rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(u[b], s.get("u"), rtol=1e-8, atol=1e-8)
