#!/bin/bash
# tools/scotty_seam.sh -- the reference's 200-step MPC run through the C++ ALTROSolver (tests/cpp/bicycle_mpc_test.cpp), with the seam's
# own clock (ALTRO_TVLQR_DROPIN_STATS=1): what a tvlqr_BackwardPass call costs inside a solver loop, and the rate with the sweeps on the
# host CPU (tests/cpp/tvlqr_cpu_seam.cpp) beside it.
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_scotty.py -q -s -k cpp_altro 2>&1 | grep "scotty through\|passed\|failed"
P=$(ls /tmp/scotty_path_*.txt 2>/dev/null | head -1)
python - <<'PY'
import numpy as np, os
from tests import problems
x_ref, u_ref, exp = problems.scotty()
with open("/tmp/scotty_path.txt", "w") as f:
    f.write("%d\n" % len(x_ref))
    for x, u in zip(x_ref, u_ref):
        f.write(" ".join("%.17g" % v for v in list(x) + list(u)) + "\n")
PY
for i in 1 2 3; do ALTRO_TVLQR_DROPIN_STATS=1 tests/cpp/bicycle_mpc_test.bin /tmp/scotty_path.txt 200 2>&1 | grep -o "Average rate.*\|tvlqr_BackwardPass seam:.*"; done
tests/cpp/bicycle_mpc_test_cpu_seam.bin /tmp/scotty_path.txt 200 2>&1 | grep "Average rate"
