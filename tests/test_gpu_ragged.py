"""Per-knot-point dimensions on the batched ABI (altro_hip_batch_create_dims; VERDICT r3 missing #6): the reference's kernel
boundary takes nx[k], nu[k] per knot point throughout (tvlqr.cpp:65-248: A_k is nx[k+1] x nx[k], B_k nx[k+1] x nu[k], K_k
nu[k] x nx[k]; ALTROSolver::SetDimension sets them per range, altro_solver.cpp:26-47).  A batch of seeded random problems with
shrinking, growing and mixed state dimensions and an input dimension that changes every step goes through the device sweeps
(plan GENERIC) and, problem by problem, through oracle/tvlqr_oracle.c on the reference's own pointer tables: gains, cost-to-go,
expected decrease, status, closed-loop trajectory and duals bit for bit (fp64), 2e-4 relative in fp32.  Needs an MI355X."""
import ctypes as C

import numpy as np
import pytest

import altro_amd
from oracle import oracle
from tests import problems

pytestmark = pytest.mark.gpu

DIMS = {
    "shrinking": ([6, 6, 5, 5, 5, 4, 3, 3, 2, 2], [2, 3, 1, 2, 2, 3, 1, 2, 1]),
    "growing": ([2, 3, 3, 4, 6, 6, 7], [1, 2, 1, 3, 2, 4]),
    "mixed": ([4, 7, 3, 9, 9, 2, 5, 5], [2, 1, 3, 4, 1, 2, 2]),
}


def make(nx, nu, batch, is_diag, seed):
    """packed per-problem arrays [batch, sum_k block_k] in the reference's per-knot-point blocks (column-major)"""
    N = len(nu)
    parts = {k: [[] for _ in range(batch)] for k in ("A", "B", "f", "Q", "R", "H", "q", "r")}
    rs = 0
    for b in range(batch):
        for k in range(N + 1):
            n = nx[k]
            rs += 1
            if is_diag:
                parts["Q"][b].append(1.0 + problems.uniform01((n,), seed, rs * 101))
            else:
                L = 0.3 * problems.normal((n, n), seed + 1, rs * 103)
                parts["Q"][b].append((np.eye(n) + L @ L.T).T.reshape(-1))
            parts["q"][b].append(0.2 * problems.normal((n,), seed + 2, rs * 107))
            if k == N:
                break
            m, n2 = nu[k], nx[k + 1]
            A = 0.4 * problems.normal((n2, n), seed + 3, rs * 109)
            A[:min(n, n2), :min(n, n2)] += np.eye(min(n, n2))
            parts["A"][b].append(A.T.reshape(-1))
            parts["B"][b].append((0.5 * problems.normal((n2, m), seed + 4, rs * 113)).T.reshape(-1))
            parts["f"][b].append(0.1 * problems.normal((n2,), seed + 5, rs * 127))
            if is_diag:
                parts["R"][b].append(0.2 + 0.2 * problems.uniform01((m,), seed + 6, rs * 131))
            else:
                M = 0.2 * problems.normal((m, m), seed + 7, rs * 137)
                parts["R"][b].append((0.3 * np.eye(m) + M @ M.T).T.reshape(-1))
                parts["H"][b].append((0.03 * problems.normal((m, n), seed + 8, rs * 139)).T.reshape(-1))
            parts["r"][b].append(0.1 * problems.normal((m,), seed + 9, rs * 149))
    out = {k: (np.stack([np.concatenate(v) for v in parts[k]]) if parts[k][0] else None) for k in parts}
    out["x0"] = 2.0 * problems.uniform01((batch, nx[0]), seed + 10) - 1.0
    return out


def oracle_one(nx, nu, p, b, is_diag):
    """oracle_tvlqr_BackwardPass / _ForwardPass on pointer tables into problem b's packed arrays (scratch sized for any dimensions)"""
    L = oracle.lib()
    N = len(nu)
    nxa, nua = (C.c_int * (N + 1))(*nx), (C.c_int * N)(*nu)
    dp = C.POINTER(C.c_double)

    def table(arr, sizes):
        t = (dp * len(sizes))()
        at = 0
        for k, sz in enumerate(sizes):
            t[k] = arr[at:].ctypes.data_as(dp) if arr is not None else None
            at += sz
        return t

    n, m, n2 = np.array(nx[:N]), np.array(nu), np.array(nx[1:])
    nn = np.array(nx)
    src = {k: (np.ascontiguousarray(p[k][b]) if p[k] is not None else None) for k in ("A", "B", "f", "Q", "R", "H", "q", "r")}
    out = {"K": np.zeros(int((m * n).sum())), "d": np.zeros(int(m.sum())), "P": np.zeros(int((nn * nn).sum())), "p": np.zeros(int(nn.sum())),
           "x": np.zeros(int(nn.sum())), "u": np.zeros(int(m.sum())), "y": np.zeros(int(nn.sum()))}
    big = 33 * 33
    scratch = [np.zeros(N * big) for _ in range(10)]
    dV = np.zeros(2)
    tA, tB, tf = table(src["A"], n2 * n), table(src["B"], n2 * m), table(src["f"], n2)
    tQ, tq = table(src["Q"], nn if is_diag else nn * nn), table(src["q"], nn)
    tR, tr = table(src["R"], m if is_diag else m * m), table(src["r"], m)
    tH = table(src["H"], m * n) if not is_diag else (dp * N)()
    tK, td, tP, tp = table(out["K"], m * n), table(out["d"], m), table(out["P"], nn * nn), table(out["p"], nn)
    tS = [table(s, [big] * N) for s in scratch]
    L.oracle_tvlqr_BackwardPass.restype = C.c_int
    L.oracle_tvlqr_BackwardPass.argtypes = None
    st = L.oracle_tvlqr_BackwardPass(nxa, nua, C.c_int(N), tA, tB, tf, tQ, tR, tH, tq, tr, C.c_double(0.0), tK, td, tP, tp,
                                     dV.ctypes.data_as(dp), *tS, C.c_bool(False), C.c_bool(bool(is_diag)))
    tx, tu, ty = table(out["x"], nn), table(out["u"], m), table(out["y"], nn)
    L.oracle_tvlqr_ForwardPass.restype = C.c_int
    L.oracle_tvlqr_ForwardPass.argtypes = None
    x0 = np.ascontiguousarray(p["x0"][b])
    L.oracle_tvlqr_ForwardPass(nxa, nua, C.c_int(N), tA, tB, tf, tK, td, tP, tp, x0.ctypes.data_as(dp), tx, tu, ty)
    out["delta_V"], out["status"] = dV, st
    return out


@pytest.mark.parametrize("which", ["shrinking", "growing", "mixed"])
@pytest.mark.parametrize("is_diag", [False, True])
def test_sweeps_with_per_knot_point_dimensions_are_bit_identical_to_the_oracle(which, is_diag):
    nx, nu = DIMS[which]
    batch = 37
    p = make(nx, nu, batch, is_diag, seed=900 + len(nx))
    bt = altro_amd.Batch.with_dims(nx, nu, batch)
    assert bt.plan == altro_amd.PLAN_GENERIC
    bt.set_dynamics(p["A"], p["B"], p["f"])
    bt.set_cost(p["Q"], p["R"], p["H"], p["q"], p["r"], is_diag=is_diag)
    bt.set_initial_state(p["x0"])
    bt.sweep()
    got = {k: bt.get(k) for k in ("K", "d", "P", "p", "x", "u", "y", "delta_V", "status")}
    st = bt.stats()
    assert st.problems == batch and st.cholesky_failures == 0
    for b in (0, 1, 17, batch - 1):
        ref = oracle_one(nx, nu, p, b, is_diag)
        assert ref["status"] == -1 and got["status"][b] == -1
        for k in ("K", "d", "P", "p", "delta_V", "x", "u", "y"):
            assert np.array_equal(got[k][b], ref[k]), (which, is_diag, b, k, float(np.abs(got[k][b] - ref[k]).max()))
    nN = nx[-1]
    assert abs(st.max_abs_xN - np.abs(got["x"][:, -nN:]).max()) == 0.0
    bt.close()


@pytest.mark.parametrize("is_diag", [False, True])
def test_matrix_core_products_with_per_knot_point_dimensions(is_diag):
    """ALTRO_HIP_GENERIC_MATRIX_CORES on a handle with per-knot-point dimensions: the tiles are fed zeros past every block's own edges,
    so the changing shapes need nothing special; equal to the bit-exact form to 1e-12."""
    nx, nu = DIMS["mixed"]
    batch = 11
    p = make(nx, nu, batch, is_diag, seed=333)
    got = {}
    for name, flags in (("exact", 0), ("mc", altro_amd.GENERIC_MATRIX_CORES)):
        bt = altro_amd.Batch.with_dims(nx, nu, batch, flags=flags)
        bt.set_dynamics(p["A"], p["B"], p["f"])
        bt.set_cost(p["Q"], p["R"], p["H"], p["q"], p["r"], is_diag=is_diag)
        bt.set_initial_state(p["x0"])
        bt.sweep()
        got[name] = {k: bt.get(k) for k in ("K", "d", "P", "p", "x", "u", "y", "delta_V", "status")}
        bt.close()
    assert np.array_equal(got["exact"]["status"], got["mc"]["status"])
    for k in ("K", "d", "P", "p", "x", "u", "y", "delta_V"):
        scale = max(1.0, float(np.abs(got["exact"][k]).max()))
        assert np.abs(got["exact"][k] - got["mc"][k]).max() <= 1e-12 * scale, k


def test_fp32_handle_and_shared_problem():
    """fp32 storage (2e-4 relative) and batch_stride_zero (one problem's arrays for the whole batch)"""
    nx, nu = DIMS["mixed"]
    batch = 9
    p = make(nx, nu, 1, False, seed=77)
    bt = altro_amd.Batch.with_dims(nx, nu, batch, dtype=altro_amd.F32)
    bt.set_dynamics(p["A"], p["B"], p["f"], batch_stride_zero=True)
    bt.set_cost(p["Q"], p["R"], p["H"], p["q"], p["r"], batch_stride_zero=True)
    bt.set_initial_state(p["x0"], batch_stride_zero=True)
    bt.sweep()
    ref = oracle_one(nx, nu, p, 0, False)
    for k in ("K", "d", "P", "x", "u"):
        g = bt.get(k)
        assert np.array_equal(g[0], g[batch - 1])
        scale = max(1.0, float(np.abs(ref[k]).max()))
        assert np.abs(g[3] - ref[k]).max() <= 2e-4 * scale, k
    bt.close()


def test_what_a_ragged_handle_refuses():
    nx, nu = DIMS["shrinking"]
    with pytest.raises(altro_amd.AltroHipError, match="outside"):
        altro_amd.Batch.with_dims([3, 300, 3], [1, 1], 4)     # (dimensions up to 256 since round 5; 40 used to be refused)
    altro_amd.Batch.with_dims([3, 40, 3], [1, 1], 4).close()
    bt = altro_amd.Batch.with_dims(nx, nu, 4)
    p = make(nx, nu, 4, False, seed=5)
    with pytest.raises(altro_amd.AltroHipError, match="k_stride_zero"):
        bt.set_dynamics(p["A"], p["B"], p["f"], k_stride_zero=True)
    with pytest.raises(altro_amd.AltroHipError, match="uniform dimensions"):
        bt.set_input_guess(np.zeros((1, 1, 2)), k_stride_zero=True, batch_stride_zero=True)
    with pytest.raises(altro_amd.AltroHipError, match="uniform dimensions"):   # (the iLQR loop itself runs: tests/test_gpu_ragged_ilqr.py)
        bt.set_model(altro_amd.MODEL_PENDULUM, 0.05)
    bt.close()
