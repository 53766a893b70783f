#!/bin/bash
# merit2: new tests, MFMA16 suite, C1 solve timing A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_merit2.py tests/test_gpu_ilqr_mfma16.py tests/test_gpu_speculation.py tests/test_gpu_golden.py -x -q -s > $O/merit2_tests.log 2>&1; echo "pytest rc=$?"
tail -15 $O/merit2_tests.log
for v in 1 0; do
  ALTRO_HIP_MERIT2=$v timeout 300 python bench.py --no-cpu-baseline --repeat-seconds 0 > $O/bench_merit2_$v.json 2> $O/bench_merit2_$v.err; echo "bench MERIT2=$v rc=$?"
  python - <<PY
import json
d=json.loads(open("$O/bench_merit2_$v.json").read().strip().splitlines()[-1])
print("MERIT2=$v", d["config"]["ilqr_full_solve"], d["config"]["ilqr_sweep"]["ms"])
PY
done
