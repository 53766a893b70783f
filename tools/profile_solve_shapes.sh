#!/bin/bash
# tools/profile_solve_shapes.sh <tag> "<n m [batch] [N]>" ... -- rocprofv3 kernel-trace stats of tools/solve_shapes.py per shape
TAG=${1:-r06h}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for shape in "$@"; do
  name=$(echo $shape | tr ' -' '__')
  OUT=gpurun_out/prof_${TAG}_$name; rm -rf $OUT; mkdir -p $OUT
  SUM=gpurun_out/${TAG}_solve_$name.txt
  echo "# python tools/solve_shapes.py $shape  (rocprofv3 --kernel-trace --stats, MI355X)" > $SUM
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python tools/solve_shapes.py $shape > $OUT/trace.log 2>&1
  grep "problems, N" $OUT/trace.log >> $SUM
  python tools/rocpd_summary.py $(find $OUT -name "*.db" | sort) | grep -v "rocclr\|_pack_kernel\|unpack_kernel" | cut -c1-170 | head -16 >> $SUM
  find $OUT -name "*.db" -delete
  cat $SUM
done
