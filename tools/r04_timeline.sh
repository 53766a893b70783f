#!/bin/bash
# tools/r04_timeline.sh <tag> -- GPU busy / idle share of whole solves on plan MFMA16 (C1 LQ, C1 + input bounds, the (12, 4) MPC example)
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_${TAG}_tl; rm -rf $OUT; mkdir -p $OUT
SUM=gpurun_out/${TAG}_solve_timeline.txt; : > $SUM
run() {  # name, command...
  local name=$1; shift
  echo "## $name: $*" >> $SUM
  timeout 300 rocprofv3 --kernel-trace -d $OUT/$name -o t -- "$@" > $OUT/$name.log 2>&1
  tail -2 $OUT/$name.log | cut -c1-300 >> $SUM
  python tools/solve_timeline.py $(find $OUT/$name -name "*.db" | head -1) >> $SUM
  find $OUT/$name -name "*.db" -delete
}
run c1_lq python tools/c1_solve.py 5
run c1_al python tools/c1_solve.py 3 4096 256 --al
run mpc12 python examples/batched_linear_mpc_12x4.py
cat $SUM
