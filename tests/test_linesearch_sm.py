"""CPU: the batched solver's resumable line-search state machine (altro_amd/csrc/linesearch_sm.h, product
code, run here on the host through the C ABI) must visit exactly the trial steps of the reference's
CubicLineSearch -- checked against the REAL reference (oracle/_ref, compiled from
/root/reference/src/linesearch) and against the constants of the reference's own tests."""
import ctypes as C

import numpy as np
import pytest

import altro_amd
from oracle import oracle
from tests.test_oracle_linesearch import cubic, quad


def test_reference_test_constants():
    r = altro_amd.linesearch_host(quad(1.0, 1.0), 1.0, *quad(1.0, 1.0)(0.0))
    assert (r["iters"], r["alpha"], r["status"]) == (1, 1.0, 1)
    r = altro_amd.linesearch_host(quad(1.0, 1.1), 1.0, *quad(1.0, 1.1)(0.0), c2=0.01)
    assert r["iters"] == 3 and r["alpha"] == pytest.approx(1.1, rel=1e-15)
    r = altro_amd.linesearch_host(quad(-1.0, -0.1), 1.0, *quad(-1.0, -0.1)(0.0))
    assert r["alpha"] == 2.0 and r["status"] == 7
    for c, c2, iters in [(1.0, 0.9, 1), (1.2, 1e-3, 3), (1.8, 0.01, 4), (0.8, 0.01, 2), (0.01, 0.01, 2)]:
        r = altro_amd.linesearch_host(cubic(c), 1.0, *cubic(c)(0.0), c2=c2)
        assert r["iters"] == iters and r["alpha"] == pytest.approx(c, abs=1e-6) and r["status"] == 1


def test_state_machine_equals_real_reference():
    R = oracle.ref_linesearch()
    rng = np.random.default_rng(11)
    checked = 0
    for trial in range(500):
        kind = trial % 5
        if kind == 0:
            fn = quad(rng.uniform(0.1, 5), rng.uniform(-0.5, 3))
        elif kind == 1:
            fn = cubic(rng.uniform(0.01, 2.5))
        elif kind == 2:
            w, s_ = rng.uniform(0.5, 6), rng.uniform(0.1, 3)
            fn = lambda x, w=w, s_=s_: (-s_ * np.sin(w * x) + 0.3 * x * x, -s_ * w * np.cos(w * x) + 0.6 * x)
        elif kind == 3:
            k, c = rng.uniform(1, 30), rng.uniform(0.001, 0.5)
            fn = lambda x, k=k, c=c: (np.cosh(k * (x - c)), k * np.sinh(k * (x - c)))
        else:   # flat / nearly linear: exercises the expand-to-alpha_max path
            g = rng.uniform(0.01, 2)
            fn = lambda x, g=g: (-g * x + 1e-3 * x ** 3, -g + 3e-3 * x * x)
        phi0, dphi0 = fn(0.0)
        for tcf in (0, 1):
            for bt in (0, 1):
                mine = altro_amd.linesearch_host(fn, 1.0, phi0, dphi0, bool(tcf), bool(bt))
                # the CPU oracle restatement (itself pinned to the reference) always runs
                L = oracle.lib()
                ls = oracle.LineSearch(); L.oracle_ls_defaults(C.byref(ls))
                ls.try_cubic_first, ls.use_backtracking = tcf, bt
                oevals = []

                def ocb(a, phi, dphi, ctx):
                    p, dp = fn(a); oevals.append(a); phi[0] = p
                    if dphi: dphi[0] = dp
                oalpha = L.oracle_ls_run(C.byref(ls), oracle.MERIT_FN(ocb), None, 1.0, phi0, dphi0)
                assert mine["evals"] == oevals and mine["status"] == ls.status and mine["iters"] == ls.n_iters
                assert mine["alpha"] == oalpha or (np.isnan(oalpha) and np.isnan(mine["alpha"]))
                if R is not None:
                    revals = []

                    def rcb(a, phi, dphi, ctx):
                        p, dp = fn(a); revals.append(a); phi[0] = p
                        if dphi: dphi[0] = dp
                    st, it, ph, dph = C.c_int(), C.c_int(), C.c_double(), C.c_double()
                    ralpha = R.ref_ls_run(oracle.MERIT_FN(rcb), None, 1.0, phi0, dphi0, tcf, bt, C.byref(st),
                                          C.byref(it), C.byref(ph), C.byref(dph))
                    assert mine["evals"] == revals and mine["status"] == st.value and mine["iters"] == it.value
                    assert mine["alpha"] == ralpha
                    if st.value != 3:
                        assert mine["phi"] == ph.value
                checked += 1
    assert checked == 2000
