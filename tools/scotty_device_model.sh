#!/bin/bash
# tools/scotty_device_model.sh <tag> -- the reference's saved MPC run through ALTROSolver::SetDeviceModel under rocprofv3 (HIP API + kernel stats)
TAG=${1:-r06zz}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python - <<'PY'
import os
from tests import problems, cpp_build
x_ref, u_ref, exp = problems.scotty()
with open("/tmp/scotty.txt", "w") as f:
    f.write("%d\n" % len(x_ref))
    for x, u in zip(x_ref, u_ref):
        f.write(" ".join(repr(float(v)) for v in list(x) + list(u)) + "\n")
print(cpp_build.build("bicycle_mpc_test", defines=["DEVICE_MODEL"], out_name="bicycle_mpc_test_device_model"))
PY
OUT=gpurun_out/prof_${TAG}_scotty_dm; rm -rf $OUT; mkdir -p $OUT
tests/cpp/bicycle_mpc_test_device_model.bin /tmp/scotty.txt 200 | tail -3
timeout 300 rocprofv3 --hip-trace --kernel-trace --stats -d $OUT -o t -- tests/cpp/bicycle_mpc_test_device_model.bin /tmp/scotty.txt 200 > $OUT/log.txt 2>&1
tail -3 $OUT/log.txt
python tools/rocpd_summary.py $(find $OUT -name "*.db") 2>/dev/null | cut -c1-150 | head -12
find $OUT -name "*hip_api_stats*" | head -2
F=$(find $OUT -name "*hip_api_stats.csv" | head -1); [ -n "$F" ] && head -12 $F
find $OUT -name "*.db" -delete
