#!/bin/bash
# tools/profile_mpc.sh <tag> -- rocprofv3 kernel-trace stats of an example under examples/ (default: batched_bicycle_mpc.py)
TAG=${1:-r02}; EX=${2:-batched_bicycle_mpc}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_${TAG}_mpc; rm -rf $OUT; mkdir -p $OUT
SUM=gpurun_out/${TAG}_${EX}.txt
echo "# python examples/${EX}.py  (rocprofv3 --kernel-trace --stats, MI355X)" > $SUM
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python examples/${EX}.py > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log >> $SUM
python tools/rocpd_summary.py $(find $OUT -name "*.db" | sort) | grep -v "rocclr\|expand_copy\|gather_copy" | cut -c1-200 >> $SUM
find $OUT -name "*.db" -delete
cat $SUM
