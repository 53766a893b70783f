"""The merit function of the sweep in which a straggler of bench.py's constrained C1 workload leaves the oracle's path:
phi(alpha), phi'(alpha) on the device and in the oracle at the same steps, from the same state (both after `it` sweeps).

    PYTHONPATH=. python tests/soak/straggler_merit.py 118 7
"""
import sys

import numpy as np

import altro_amd
from oracle import oracle
from tests import problems
from tests.test_gpu_ilqr_mfma16 import _c1_bounded_oracle

b, it = int(sys.argv[1]), int(sys.argv[2])
Nf, batch = 256, 4096
x0 = (2.0 * problems.uniform01((batch, 12), 21) - 1.0)[b:b + 1]
one = problems.c1_double_integrator(1, N=Nf)
bt = altro_amd.Batch(Nf, 12, 4, 1)
bt.set_dynamics(one["A"][0, :1], one["B"][0, :1], None, k_stride_zero=True, batch_stride_zero=True)
Qd = np.stack([np.ones(12), 100.0 * np.ones(12)])
bt.set_tracking_cost(Qd, np.full((1, 4), 1e-2), np.zeros((2, 12)), np.zeros((1, 4)), k_stride_zero=True, batch_stride_zero=True)
bt.set_initial_state(x0)
bt.set_input_guess(np.zeros((1, 1, 4)), k_stride_zero=True, batch_stride_zero=True)
Gb = np.zeros((8, 16)); Gb[:4, 12:] = np.eye(4); Gb[4:, 12:] = -np.eye(4)
bt.add_linear_constraint(0, Nf - 1, altro_amd.CONE_INEQUALITY, Gb, np.full(8, 2.0))
res = bt.ilqr_solve(iterations_max=it)
s, st, iters, log = _c1_bounded_oracle(x0[0], it)
print("after %d sweeps: device status %d iterations %d, oracle %d %d; |x_dev - x_or| = %.3e" %
      (it, res["status"][0], res["iterations"][0], st, iters, np.abs(bt.get_nominal()[0][0] - s.get("x")).max()))
# the next sweep's expansion and backward pass, then the merit function along the step
bt.expand(); bt.backward()
s.L.oracle_ilqr_calc_expansions(s.h)
assert s.L.oracle_ilqr_backward_pass(s.h) == -1
print("|K_dev - K_or| = %.3e  |d_dev - d_or| = %.3e" % (np.abs(bt.get("K")[0] - s.get("K")).max(), np.abs(bt.get("d")[0] - s.get("d")).max()))
full = _c1_bounded_oracle(x0[0], 40)[3]
print("oracle's row of that sweep: alpha %.8g phi0 %.15g phi %.15g dphi0 %.6e ls_iters %d" % (full[it][0], full[it][1], full[it][2], full[it][3], full[it][5]))
for a in [0.0, 1.0, 0.5, 0.25, 0.1, 0.03, 0.01, 0.005, full[it][0], 0.00128107, 0.001, 0.0005, 0.0001]:
    pd, dd = bt.merit(float(a))
    po, do = s.merit(float(a))
    print("alpha %-12.8g phi dev %.15g or %.15g diff %.2e (rel %.1e) | dphi dev %.9e or %.9e diff %.2e | phi-phi0 %.3e  c1*a*dphi0 %.3e" %
          (a, pd[0], po, pd[0] - po, abs(pd[0] - po) / abs(po), dd[0], do, dd[0] - do, po - full[it][1], 1e-4 * a * full[it][3]))
