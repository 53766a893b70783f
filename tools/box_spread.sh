#!/bin/bash
# tools/box_spread.sh -- why does the C1 backward sweep take 0.79-0.86 ms depending on the box?  Clocks and power of THIS box while the
# sweep loops, next to its time: run it on a few boxes (one gpurun each) and compare.  Output: gpurun_out/box_spread_<pci>.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/box_spread_$$.txt
python - > $OUT.sweep 2>&1 <<'PY' &
import sys, time, numpy as np
sys.path.insert(0, ".")
import altro_amd
from tests import problems
N, n, m, B = 256, 12, 4, 4096
pr = problems.random_ltv(64, N, n, m)
bt = altro_amd.Batch(N, n, m, B, plan=altro_amd.PLAN_MFMA16)
bt.set_host_batch(64)
bt.set_dynamics(pr["A"], pr["B"], pr["f"]); bt.set_cost(pr["Q"], pr["R"], pr["H"], pr["q"], pr["r"])
bt.set_host_batch(0)
bt.set_initial_state(np.tile(pr["x0"][:64], (B // 64, 1)))
print("device", altro_amd.device_info(0), flush=True)
t_end = time.time() + 12.0
while time.time() < t_end:
    ts = []
    for _ in range(200):
        t0 = time.perf_counter(); bt.backward(); bt.synchronize(); ts.append(time.perf_counter() - t0)
    print("backward median %.4f ms  min %.4f" % (1e3 * np.median(ts), 1e3 * min(ts)), flush=True)
PY
sleep 6
/opt/rocm/bin/rocm-smi --showclocks --showpower --showtemp --showperflevel > $OUT.smi 2>&1
/opt/rocm/bin/rocm-smi --showmemvendor --showvoltage >> $OUT.smi 2>&1
wait
cat $OUT.sweep $OUT.smi | grep -v "^$\|====\|amdgpu.ids" | cut -c1-150 > $OUT
rm -f $OUT.sweep $OUT.smi
cat $OUT
