#!/bin/bash
# tools/profile_c1_slots.sh <tag> -- rocprofv3 kernel-trace stats of the constrained C1 solve with an input box (two slots of plan
# MFMA16's knot-point table) and with an input box + a loose state box (four slots): which kernels the extra slots cost.
TAG=${1:-r06f}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export C1_XBOX=4.0
for v in al boxes; do
  OUT=gpurun_out/prof_${TAG}_$v; rm -rf $OUT; mkdir -p $OUT
  SUM=gpurun_out/${TAG}_c1_solve_$v.txt
  CMD="python tools/c1_solve.py 3 4096 256 --$v"
  echo "# $CMD  (C1_XBOX=$C1_XBOX; rocprofv3 --kernel-trace --stats, MI355X)" > $SUM
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
  grep "^C1 solve" $OUT/trace.log >> $SUM
  python tools/rocpd_summary.py $(find $OUT -name "*.db" | sort) | grep -v "rocclr\|_pack_kernel\|unpack_kernel" | cut -c1-200 >> $SUM
  find $OUT -name "*.db" -delete
done
head -20 gpurun_out/${TAG}_c1_solve_*.txt
