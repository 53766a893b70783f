#!/bin/bash
# tools/profile_default.sh <tag> -- the DEFAULT bench command under rocprofv3: kernel-trace stats, then FETCH_SIZE and
# WRITE_SIZE in their own passes (kernel-trace only, each pass bounded); summary in gpurun_out/<tag>_default_rocprofv3.txt
TAG=${1:-r02z}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python bench.py --no-cpu-baseline --no-live-traffic"
OUT=gpurun_out/prof_${TAG}_default; rm -rf $OUT; mkdir -p $OUT
SUM=gpurun_out/${TAG}_default_rocprofv3.txt
echo "# $CMD  (= the default bench command without its CPU leg; rocprofv3, MI355X; one pass per counter set, --kernel-trace only)" > $SUM
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
grep '^{' $OUT/trace.log >> $SUM
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $ctrs -d $OUT/pmc$i -o t -- $CMD > $OUT/pmc$i.log 2>&1
done
python tools/rocpd_summary.py $(find $OUT -name "*.db" | sort) | grep -v "rocclr\|expand_copy\|gather_copy\|_pack_kernel\|unpack_kernel" | cut -c1-220 >> $SUM
find $OUT -name "*.db" -delete
