#!/bin/bash
# the C1 shape with input bounds (AL path of plan MFMA16) under rocprofv3 --kernel-trace
TAG=${1:-r03l}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SUM=gpurun_out/${TAG}_c1_al_solve.txt
python tools/c1_solve.py 5 4096 256 --al > $SUM 2>&1
OUT=gpurun_out/prof_${TAG}_al; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o t -- python tools/c1_solve.py 5 4096 256 --al > $OUT/log.txt 2>&1
python tools/rocpd_summary.py $(find $OUT -name "*.db" | sort) | grep -v "rocclr\|expand_copy\|gather_copy\|_pack_kernel\|unpack_kernel\|^===" | cut -c1-160 >> $SUM
find $OUT -name "*.db" -delete
cat $SUM
