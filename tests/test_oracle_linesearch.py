"""Pins oracle/linesearch_oracle.c against (a) the constants in the reference's line-search tests
(src/linesearch/test/linesearch_tests.cpp:134-270) and (b) the REAL reference code, compiled from
/root/reference/src/linesearch into oracle/_ref/liblinesearch_ref.so (`make -C oracle ref`).  CPU."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle


def run_oracle(fn, c1=1e-4, c2=0.9, try_cubic_first=0, backtracking=0):
    L = oracle.lib()
    ls = oracle.LineSearch()
    L.oracle_ls_defaults(C.byref(ls))
    ls.c1, ls.c2, ls.try_cubic_first, ls.use_backtracking = c1, c2, try_cubic_first, backtracking
    evals = []

    def cb(a, phi, dphi, ctx):
        p, dp = fn(a)
        evals.append(a)
        phi[0] = p
        if dphi:
            dphi[0] = dp

    phi0, dphi0 = fn(0.0)
    alpha = L.oracle_ls_run(C.byref(ls), oracle.MERIT_FN(cb), None, 1.0, phi0, dphi0)
    return alpha, ls, evals


def quad(a, c):
    return lambda x: (a * (x - c) ** 2, 2 * a * (x - c))


def cubic(c):
    return lambda x: ((x - c) ** 2 - (x - c) ** 3, 2 * (x - c) - 3 * (x - c) ** 2)


def test_reference_quadratic_cases():
    alpha, ls, _ = run_oracle(quad(1.0, 1.0))
    assert ls.n_iters == 1 and alpha == 1.0 and ls.status == 1
    alpha, ls, _ = run_oracle(quad(1.0, 1.1))
    assert ls.n_iters == 1 and alpha == 1.0
    alpha, ls, _ = run_oracle(quad(1.0, 1.1), c2=0.01)
    assert ls.n_iters == 3 and alpha == pytest.approx(1.1, rel=1e-15)
    alpha, ls, _ = run_oracle(quad(1.0, 0.8), c2=0.1)
    assert alpha == pytest.approx(0.8, rel=1e-15) and ls.status == 1
    alpha, ls, _ = run_oracle(quad(-1.0, -0.1))
    assert alpha == 2.0 and ls.sufficient_decrease and not ls.curvature and ls.status == 7


@pytest.mark.parametrize("c,c2,iters", [(1.0, 0.9, 1), (1.2, 1e-3, 3), (1.8, 0.01, 4), (0.8, 0.01, 2), (0.01, 0.01, 2)])
def test_reference_cubic_cases(c, c2, iters):
    alpha, ls, _ = run_oracle(cubic(c), c2=c2)
    assert ls.n_iters == iters
    assert alpha == pytest.approx(c, abs=1e-6)
    assert ls.status == 1


def test_against_real_reference_linesearch():
    R = oracle.ref_linesearch()
    if R is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    rng = np.random.default_rng(7)
    n_checked = 0
    for trial in range(400):
        kind = trial % 4
        if kind == 0:
            a, c = rng.uniform(0.1, 5), rng.uniform(-0.5, 3)
            fn = quad(a, c)
        elif kind == 1:
            fn = cubic(rng.uniform(0.01, 2.5))
        elif kind == 2:
            w, s = rng.uniform(0.5, 6), rng.uniform(0.1, 3)
            fn = lambda x, w=w, s=s: (-s * np.sin(w * x) + 0.3 * x * x, -s * w * np.cos(w * x) + 0.6 * x)
        else:
            k, c = rng.uniform(1, 30), rng.uniform(0.001, 0.5)
            fn = lambda x, k=k, c=c: (np.cosh(k * (x - c)), k * np.sinh(k * (x - c)))
        phi0, dphi0 = fn(0.0)
        for tcf in (0, 1):
            for bt in (0, 1):
                alpha, ls, evals = run_oracle(fn, try_cubic_first=tcf, backtracking=bt)
                revals = []

                def cb(a, phi, dphi, ctx):
                    p, dp = fn(a)
                    revals.append(a)
                    phi[0] = p
                    if dphi:
                        dphi[0] = dp

                st, it, ph, dph = C.c_int(), C.c_int(), C.c_double(), C.c_double()
                ralpha = R.ref_ls_run(oracle.MERIT_FN(cb), None, 1.0, phi0, dphi0, tcf, bt,
                                      C.byref(st), C.byref(it), C.byref(ph), C.byref(dph))
                assert st.value == ls.status
                assert it.value == ls.n_iters
                assert revals == evals          # identical sequence of trial steps, bit for bit
                assert (np.isnan(ralpha) and np.isnan(alpha)) or ralpha == alpha
                if st.value != 3:
                    assert ph.value == ls.phi
                n_checked += 1
    assert n_checked == 1600
