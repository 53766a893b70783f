"""altro_amd -- MI355X-native ALTRO iLQR inner loop.

The product is the C-ABI shared library (include/altro_hip/altro_hip.h, built from altro_amd/csrc by
`python -m altro_amd.build`).  This module is only the ctypes binding a Python caller would write:
it holds no arithmetic and no CPU fallback -- if the library or a HIP device is missing, calls fail.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libaltro_hip.so")

F64, F32 = 0, 1
PLAN_AUTO, PLAN_GENERIC, PLAN_MFMA16, PLAN_LANE = 0, 1, 2, 3
STORE_QBLOCKS = 0x1
TVLQR_SUCCESS = -1

# every symbol include/altro_hip/altro_hip.h declares (tests check the .so exports all of them)
C_ABI_SYMBOLS = [
    "altro_hip_version", "altro_hip_last_error", "altro_hip_device_count", "altro_hip_device_info",
    "altro_hip_batch_create", "altro_hip_batch_destroy", "altro_hip_batch_plan",
    "altro_hip_batch_device_bytes", "altro_hip_set_dynamics", "altro_hip_set_cost",
    "altro_hip_set_initial_state", "altro_hip_backward", "altro_hip_forward_ltv", "altro_hip_sweep",
    "altro_hip_synchronize", "altro_hip_get_K", "altro_hip_get_d", "altro_hip_get_P",
    "altro_hip_get_p", "altro_hip_get_x", "altro_hip_get_u", "altro_hip_get_y",
    "altro_hip_get_delta_V", "altro_hip_get_status", "altro_hip_get_qblocks",
    "altro_hip_stats_reduce", "altro_hip_profile_enable", "altro_hip_profile_reset",
    "altro_hip_profile_get", "altro_hip_algorithmic_bytes",
]


class AltroHipError(RuntimeError):
    pass


class Stats(C.Structure):
    _fields_ = [("problems", C.c_int64), ("cholesky_failures", C.c_int64),
                ("sum_delta_V0", C.c_double), ("sum_delta_V1", C.c_double),
                ("max_abs_xN", C.c_double)]


_lib = None


def lib():
    """Loads the HIP library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AltroHipError("%s is missing: run `python -m altro_amd.build` (hipcc, gfx950). "
                                "There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp, i, d = C.c_void_p, C.c_int, C.c_double
        L.altro_hip_last_error.restype = C.c_char_p
        L.altro_hip_device_info.argtypes = [i, C.c_char_p, i, C.POINTER(i), C.POINTER(i)]
        L.altro_hip_batch_create.argtypes = [C.POINTER(vp), i, i, i, i, i, i, C.c_uint, i, vp]
        L.altro_hip_batch_destroy.argtypes = [vp]
        L.altro_hip_batch_destroy.restype = None
        L.altro_hip_batch_plan.argtypes = [vp]
        L.altro_hip_batch_device_bytes.argtypes = [vp]
        L.altro_hip_batch_device_bytes.restype = C.c_size_t
        L.altro_hip_set_dynamics.argtypes = [vp, vp, vp, vp, i, i]
        L.altro_hip_set_cost.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i]
        L.altro_hip_set_initial_state.argtypes = [vp, vp, i]
        L.altro_hip_backward.argtypes = [vp, d]
        L.altro_hip_forward_ltv.argtypes = [vp]
        L.altro_hip_sweep.argtypes = [vp, d]
        L.altro_hip_synchronize.argtypes = [vp]
        for g in ("K", "d", "P", "p", "x", "u", "y", "delta_V", "status", "qblocks"):
            getattr(L, "altro_hip_get_" + g).argtypes = [vp, vp]
        L.altro_hip_stats_reduce.argtypes = [vp, C.POINTER(Stats)]
        L.altro_hip_profile_enable.argtypes = [vp, i]
        L.altro_hip_profile_reset.argtypes = [vp]
        L.altro_hip_profile_get.argtypes = [vp, i, C.POINTER(i), C.POINTER(d), C.POINTER(C.c_char_p)]
        L.altro_hip_algorithmic_bytes.argtypes = [vp, i]
        L.altro_hip_algorithmic_bytes.restype = d
        L.altro_hip_selftest_mfma_f64.argtypes = [i]
        L.altro_hip_selftest_mfma_f64.restype = d
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise AltroHipError("altro_hip error %d: %s" % (rc, lib().altro_hip_last_error().decode()))


def _in(a):
    if a is None:
        return None, None
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.c_void_p)


class Batch:
    """`batch` independent TVLQR / iLQR problems on one MI355X (mirror of the C handle)."""

    def __init__(self, N, n, m, batch, dtype=F64, plan=PLAN_AUTO, flags=0, device=0, stream=None):
        self.L = lib()
        self.N, self.n, self.m, self.batch = N, n, m, batch
        self.h = C.c_void_p()
        _check(self.L.altro_hip_batch_create(C.byref(self.h), N, n, m, batch, dtype, plan, flags,
                                             device, stream))
        self.plan = self.L.altro_hip_batch_plan(self.h)

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.altro_hip_batch_destroy(self.h)
            self.h = None

    __del__ = close

    def device_bytes(self):
        return self.L.altro_hip_batch_device_bytes(self.h)

    def set_dynamics(self, A, B, f=None, k_stride_zero=False, batch_stride_zero=False):
        (a, pa), (b, pb), (c, pc) = _in(A), _in(B), _in(f)
        _check(self.L.altro_hip_set_dynamics(self.h, pa, pb, pc, int(k_stride_zero), int(batch_stride_zero)))

    def set_cost(self, Q, R, H, q, r, is_diag=False, k_stride_zero=False, batch_stride_zero=False):
        keep = [_in(v) for v in (Q, R, H, q, r)]
        _check(self.L.altro_hip_set_cost(self.h, *[k[1] for k in keep], int(is_diag),
                                         int(k_stride_zero), int(batch_stride_zero)))

    def set_initial_state(self, x0, batch_stride_zero=False):
        a, pa = _in(x0)
        _check(self.L.altro_hip_set_initial_state(self.h, pa, int(batch_stride_zero)))

    def backward(self, reg=0.0):
        _check(self.L.altro_hip_backward(self.h, float(reg)))

    def forward_ltv(self):
        _check(self.L.altro_hip_forward_ltv(self.h))

    def sweep(self, reg=0.0):
        _check(self.L.altro_hip_sweep(self.h, float(reg)))

    def synchronize(self):
        _check(self.L.altro_hip_synchronize(self.h))

    def get(self, name):
        N, n, m, B = self.N, self.n, self.m, self.batch
        shape = {"K": (B, N, m * n), "d": (B, N, m), "P": (B, N + 1, n * n), "p": (B, N + 1, n),
                 "x": (B, N + 1, n), "u": (B, N, m), "y": (B, N + 1, n), "delta_V": (B, 2),
                 "qblocks": (B, N, n * n + m * m + m * n + n + m)}
        if name == "status":
            out = np.zeros(B, dtype=np.int32)
        else:
            out = np.zeros(shape[name], dtype=np.float64)
        _check(getattr(self.L, "altro_hip_get_" + name)(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def stats(self):
        s = Stats()
        _check(self.L.altro_hip_stats_reduce(self.h, C.byref(s)))
        return s

    def profile(self, enable=True):
        _check(self.L.altro_hip_profile_enable(self.h, int(enable)))
        _check(self.L.altro_hip_profile_reset(self.h))

    def profile_get(self, slot):
        n, ms, name = C.c_int(), C.c_double(), C.c_char_p()
        _check(self.L.altro_hip_profile_get(self.h, slot, C.byref(n), C.byref(ms), C.byref(name)))
        return n.value, ms.value, name.value.decode()

    def algorithmic_bytes(self, slot):
        return self.L.altro_hip_algorithmic_bytes(self.h, slot)
