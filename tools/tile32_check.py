"""Plan MFMA32 (kernels/tvlqr_tile32.hip) against the CPU oracle over a list of shapes: largest error of K, d, P, p, delta_V,
x, u, y per shape, the Cholesky-failure index on an indefinite batch, and a sweep time.  Development tool (the assertions live in
tests/test_gpu_tile32.py).   python tools/tile32_check.py [n,m ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import altro_amd
from oracle import oracle
from tests import problems

SHAPES = [(13, 4), (16, 4), (14, 7), (24, 8), (28, 4), (13, 1), (15, 1), (13, 3), (16, 1), (17, 3), (20, 8), (21, 5), (25, 7),
          (29, 3), (31, 1), (5, 8), (8, 8), (9, 7), (12, 5), (12, 8), (15, 8), (16, 8), (19, 2)]


def relerr(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    if a.ndim >= 3:
        ax = tuple(range(2, a.ndim))
        return float((np.abs(a - b).max(axis=ax) / np.maximum(1.0, np.abs(b).max(axis=ax))).max())
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


def one(n, m, N=24, batch=6, with_f=True, quiet=False):
    pr = problems.random_ltv(batch, N, n, m)
    if not with_f:
        pr["f"] = np.zeros_like(pr["f"])
    bt = altro_amd.Batch(N, n, m, batch, plan=altro_amd.PLAN_MFMA32)
    assert bt.plan == altro_amd.PLAN_MFMA32
    bt.set_dynamics(pr["A"], pr["B"], pr["f"] if with_f else None)
    bt.set_cost(pr["Q"], pr["R"], pr["H"], pr["q"], pr["r"])
    bt.set_initial_state(pr["x0"])
    bt.backward(0.0)
    st = bt.get("status")
    ref = oracle.backward_batch(pr["A"], pr["B"], pr["f"], pr["Q"], pr["R"], pr["H"], pr["q"], pr["r"], 0.0, False)
    errs = {k: relerr(bt.get(k), ref[k]) for k in ("K", "d", "P", "p")}
    errs["abs_K"] = float(np.abs(bt.get("K") - ref["K"]).max())
    errs["abs_d"] = float(np.abs(bt.get("d") - ref["d"]).max())
    errs["dV"] = relerr(bt.get("delta_V"), ref["dV"])
    ok_status = bool((st == ref["status"]).all())
    bt.forward_ltv()
    fw = oracle.forward_batch(pr["A"], pr["B"], pr["f"], ref["K"], ref["d"], ref["P"], ref["p"], pr["x0"])
    for k in ("x", "u", "y"):
        errs[k] = relerr(bt.get(k), fw[k])
    worst = max(errs.values())
    if not quiet or not (worst < 1e-9 and ok_status):
        print("(%2d,%2d) f=%d status_ok=%s worst=%.2e  " % (n, m, with_f, ok_status, worst) +
              " ".join("%s=%.1e" % kv for kv in errs.items()), flush=True)
    return worst < 1e-9 and ok_status


def failing(n, m, N=12, batch=8):
    """R indefinite at a few knot points: the failing knot point must be the oracle's, K_k = Qux, d_k = -Qu there."""
    pr = problems.random_ltv(batch, N, n, m)
    rng = np.random.default_rng(5)
    for b in range(0, batch, 2):
        k = int(rng.integers(1, N - 1))
        R = pr["R"][b, k].reshape(m, m)
        R -= 50.0 * np.eye(m)
    bt = altro_amd.Batch(N, n, m, batch, plan=altro_amd.PLAN_MFMA32)
    bt.set_dynamics(pr["A"], pr["B"], pr["f"]); bt.set_cost(pr["Q"], pr["R"], pr["H"], pr["q"], pr["r"])
    bt.backward(0.0)
    ref = oracle.backward_batch(pr["A"], pr["B"], pr["f"], pr["Q"], pr["R"], pr["H"], pr["q"], pr["r"], 0.0, False)
    st = bt.get("status")
    same = bool((st == ref["status"]).all())
    K, d = bt.get("K"), bt.get("d")
    worst = 0.0
    for b in range(batch):
        k0 = ref["status"][b] if ref["status"][b] >= 0 else 0
        worst = max(worst, relerr(K[b:b + 1, k0:], ref["K"][b:b + 1, k0:]), relerr(d[b:b + 1, k0:], ref["d"][b:b + 1, k0:]))
    print("(%2d,%2d) failing: status %s vs %s same=%s  K,d from the failing knot point on: %.2e" % (n, m, st.tolist(), ref["status"].tolist(), same, worst), flush=True)
    return same and worst < 1e-9


def timing(n, m, N=128, batch=4096):
    pr = problems.random_ltv(4, N, n, m)
    bt = altro_amd.Batch(N, n, m, batch, plan=altro_amd.PLAN_MFMA32)
    tile = lambda a: np.ascontiguousarray(np.broadcast_to(a[:1], (batch,) + a.shape[1:]))
    bt.set_dynamics(tile(pr["A"]), tile(pr["B"]), tile(pr["f"]))
    bt.set_cost(tile(pr["Q"]), tile(pr["R"]), tile(pr["H"]), tile(pr["q"]), tile(pr["r"]))
    bt.set_initial_state(tile(pr["x0"]))
    for _ in range(3):
        bt.sweep()
    bt.synchronize()
    out = []
    for fn in (bt.backward, bt.forward_ltv):
        t0 = time.perf_counter()
        for _ in range(10):
            fn(0.0) if fn == bt.backward else fn()
        bt.synchronize()
        out.append((time.perf_counter() - t0) / 10 * 1e3)
    el_b = 3 * n * n + 3 * n * m + m * m + 3 * n + 2 * m
    el_f = 2 * n * n + 2 * n * m + 4 * n + 2 * m
    fr = lambda el, ms: el * 8.0 * N * batch / (ms * 1e-3) / 8e12
    print("(%2d,%2d) N=%d batch=%d: backward %.3f ms (%.2f of 8 TB/s)  forward %.3f ms (%.2f)  sweep %.3f ms" %
          (n, m, N, batch, out[0], fr(el_b, out[0]), out[1], fr(el_f, out[1]), out[0] + out[1]), flush=True)


if __name__ == "__main__":
    shapes = [tuple(int(v) for v in s.split(",")) for s in sys.argv[1:] if "," in s] or SHAPES
    if "--time-only" in sys.argv:
        for n, m in shapes:
            timing(n, m)
        sys.exit(0)
    if "--all" in sys.argv:
        shapes = [(n, m) for n in range(5, 32) for m in range(1, 9) if n + m <= 32 and not (n <= 12 and m <= 4)]
    good = True
    for n, m in shapes:
        good &= one(n, m, N=10 if "--all" in sys.argv else 24, batch=3 if "--all" in sys.argv else 6, quiet="--all" in sys.argv)
    print("%d shapes checked" % len(shapes), flush=True)
    good &= one(13, 4, with_f=False)
    good &= one(13, 4, N=3, batch=1)
    good &= one(20, 6, N=1, batch=3)
    for n, m in [(13, 4), (24, 8), (16, 4)]:
        good &= failing(n, m)
    print("ALL OK" if good else "SOME FAILED", flush=True)
    if "--time" in sys.argv:
        for n, m in [(13, 4), (16, 4), (14, 7), (24, 8), (28, 4), (20, 4), (8, 8), (31, 1)]:
            timing(n, m)
