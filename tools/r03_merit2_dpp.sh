#!/bin/bash
# tools/r03_merit2_dpp.sh <tag>: the C1 iLQR solve with the DPP form of the two-trial merit evaluation and with the LDS form
TAG=${1:-r03q}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SUM=gpurun_out/${TAG}_c1_solve_dpp.txt
: > $SUM
timeout 600 python -m pytest tests/test_gpu_merit2.py -x -q > gpurun_out/${TAG}_merit2_tests.log 2>&1
tail -5 gpurun_out/${TAG}_merit2_tests.log >> $SUM
for v in 1 0; do
  export ALTRO_HIP_MERIT2_DPP=$v
  echo "# ---- ALTRO_HIP_MERIT2_DPP=$v: python tools/c1_solve.py 9 (host clock, no profiler)" >> $SUM
  python tools/c1_solve.py 9 >> $SUM 2>&1
  OUT=gpurun_out/prof_${TAG}_d$v; rm -rf $OUT; mkdir -p $OUT
  echo "# rocprofv3 --kernel-trace --stats -- python tools/c1_solve.py 9   (10 solves incl. the untimed one)" >> $SUM
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o t -- python tools/c1_solve.py 9 > $OUT/log.txt 2>&1
  python tools/rocpd_summary.py $(find $OUT -name "*.db" | sort) | grep -v "rocclr\|expand_copy\|gather_copy\|_pack_kernel\|unpack_kernel\|^===" | cut -c1-160 >> $SUM
  find $OUT -name "*.db" -delete
done
unset ALTRO_HIP_MERIT2_DPP
cat $SUM
