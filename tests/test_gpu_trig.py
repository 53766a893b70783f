"""The device's fp64 sincos (altro_amd/csrc/models.h: sincos_hd) against near-exact references: every nonlinear model's
rollout and Jacobian goes through it, and the parity tolerances of the pendulum / bicycle tests (1e-10) rest on it being
within a couple of ulp of the host's."""
import ctypes as C

import numpy as np
import pytest

import altro_amd
from tests import problems

pytestmark = pytest.mark.gpu


def _device_sincos(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    s = np.zeros_like(x); c = np.zeros_like(x)
    rc = altro_amd.lib().altro_hip_selftest_sincos(0, x.ctypes.data_as(C.c_void_p), x.size, s.ctypes.data_as(C.c_void_p),
                                                   c.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return s, c


def _ulps(got, ref):
    """error in units of the last place of the reference (np.longdouble: 64-bit mantissa on x86, a near-exact reference)"""
    ref64 = ref.astype(np.float64)
    ulp = np.spacing(np.abs(ref64))
    return np.abs(got.astype(np.longdouble) - ref) / ulp


def test_sincos_accuracy():
    u = problems.uniform01((400000,), 91)
    xs = [
        (u - 0.5) * 2.0 * np.pi,                       # one period
        (u - 0.5) * 200.0,                             # the range trajectories live in
        (u - 0.5) * 2.0 * 1048575.0,                   # up to the switch-over to the library routine
        (u - 0.5) * 2.0e9,                             # beyond it
        np.round((u - 0.5) * 4000.0) * (np.pi / 2) + (u[::-1] - 0.5) * 1e-6,    # around the multiples of pi/2
        np.round((u - 0.5) * 4000.0) * (np.pi / 2),    # the doubles nearest to them
        (u - 0.5) * 1e-3, (u - 0.5) * 1e-9,            # tiny arguments
    ]
    worst = 0.0
    for x in xs:
        s, c = _device_sincos(x)
        xl = x.astype(np.longdouble)
        es, ec = _ulps(s, np.sin(xl)), _ulps(c, np.cos(xl))
        worst = max(worst, float(es.max()), float(ec.max()))
        assert es.max() <= 2.0 and ec.max() <= 2.0, (float(es.max()), float(ec.max()), float(np.abs(x).max()))
        assert np.all(np.abs(s * s + c * c - 1.0) < 1e-15)
    print("worst error %.3f ulp" % worst)


def test_sincos_special_values():
    x = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e300, -1e300, 1048576.0, -1048576.0, 1048575.999])
    s, c = _device_sincos(x)
    assert s[0] == 0.0 and c[0] == 1.0 and np.signbit(s[1]) and s[1] == 0.0 and c[1] == 1.0
    assert np.isnan(s[2]) and np.isnan(c[2]) and np.isnan(s[3]) and np.isnan(s[4]) and np.isnan(c[4])
    xl = x[5:].astype(np.longdouble)
    assert np.all(_ulps(s[5:], np.sin(xl)) <= 2.0) and np.all(_ulps(c[5:], np.cos(xl)) <= 2.0)
